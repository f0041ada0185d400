"""Layer components (reference ``deepfm/models/layers/__init__.py``)."""
from deepfm_amd.models.layers.attention import MultiHeadSelfAttention
from deepfm_amd.models.layers.cin import CIN
from deepfm_amd.models.layers.dnn import DNN
from deepfm_amd.models.layers.embedding import FeatureEmbedding
from deepfm_amd.models.layers.fm import FMInteraction

__all__ = ["CIN", "DNN", "FeatureEmbedding", "FMInteraction", "MultiHeadSelfAttention"]
