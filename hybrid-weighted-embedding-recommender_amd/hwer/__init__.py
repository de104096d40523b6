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


class GcnScoredIndex(VectorIndex):
    """Retrieval flavour of GcnNCF without NCF (reference: hwer/gcn_ncf.py:363-387, 447-456)."""
    scoring = "gcn"
