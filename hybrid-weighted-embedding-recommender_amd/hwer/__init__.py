"""hwer — MI355X-native drop-in for the retrieval hot path of faizanahemad/Hybrid-Weighted-Embedding-Recommender.

Exports mirror the reference's hwer/__init__.py:1-4 for the classes on the hot path.  GcnNCF training and the
feature embedders are out of scope of this build; `GcnScoredIndex` gives GcnNCF's retrieval
flavour (float32 vectors, score = (2 - dist)/2) over externally trained vectors.
"""
from .recommendation_base import RecommendationBase, Node, Edge, MultiKNN  # noqa: F401
from .content_recommender import ContentRecommendation, ContentRecommender  # noqa: F401
from .embed import BaseEmbed, IdentityEmbedding  # noqa: F401
from .utils import NodeNotFoundException  # noqa: F401


class VectorIndex(RecommendationBase):
    """RecommendationBase over vectors that were trained elsewhere: fit(nodes, vectors)."""

    def fit(self, nodes, vectors=None, node_data=None, **kwargs):
        assert not self.fit_done
        self.add_nodes(nodes)
        self.__build_knn__(vectors)
        self.fit_done = True
        return self.vectors


def prepare_for_knn(collaborative_vectors, n_dims, device=None):
    """GcnNCF.prepare_for_knn (reference: hwer/gcn_ncf.py:447-456): PCA to `n_dims` only if the vectors are wider,
    ValueError if they are narrower, then unit-length rows - all on the GPU (PCA: hwer/fit_pca.py device_pca_unit, float64 Gram
    + projection kernels; otherwise the blend kernel with one table and alpha = 1).  Returns float32 rows, the dtype the
    reference's GcnNCF hands to __build_knn__."""
    import numpy as np
    from .content_recommender import blend_tables
    v = np.asarray(collaborative_vectors)
    if v.shape[1] > n_dims:
        from .fit_pca import device_pca_unit
        return device_pca_unit(v, n_dims, device=device)
    elif v.shape[1] < n_dims:
        raise ValueError()
    return blend_tables([v.astype(np.float32)], np.array([1.0], dtype=np.float32), device=device)


class GcnScoredIndex(VectorIndex):
    """Retrieval flavour of GcnNCF without NCF (reference: hwer/gcn_ncf.py:363-387, 447-456)."""
    scoring = "gcn"

    def fit(self, nodes, vectors=None, node_data=None, **kwargs):
        return super().fit(nodes, prepare_for_knn(vectors, self.n_dims, self.device))
