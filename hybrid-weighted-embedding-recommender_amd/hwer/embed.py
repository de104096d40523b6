"""Minimal embedder interface (reference: hwer/embed/BaseEmbed.py).  Feature embedders are OUT OF SCOPE of this
build: the hot path takes embedding tables as given.  Only the abstract interface and the
pass-through embedder are provided so that ContentRecommendation.fit has something to call."""
import abc

import numpy as np


class BaseEmbed(metaclass=abc.ABCMeta):
    def __init__(self, n_dims, make_unit_length=True, **kwargs):
        self.n_dims, self.make_unit_length, self.kwargs = n_dims, make_unit_length, kwargs
        self.is_fit = False

    @abc.abstractmethod
    def fit(self, feature, **kwargs):
        assert not self.is_fit
        self.is_fit = True

    @abc.abstractmethod
    def transform(self, feature, **kwargs) -> np.ndarray:
        assert self.is_fit

    def fit_transform(self, feature, **kwargs) -> np.ndarray:
        self.fit(feature, **kwargs)
        return self.check_output_dims(self.transform(feature, **kwargs), feature)

    def check_output_dims(self, output, feature):
        if output.shape[1] != self.n_dims or output.shape[0] != len(feature):
            raise ValueError("Unmatched Dims. Output Dims = %s, Required Dims = (%s,%s)" % (output.shape, len(feature), self.n_dims))
        return output


class IdentityEmbedding(BaseEmbed):
    def fit(self, feature, **kwargs):
        super().fit(feature, **kwargs)

    def transform(self, feature, **kwargs):
        return self.check_output_dims(np.array(feature), feature)
