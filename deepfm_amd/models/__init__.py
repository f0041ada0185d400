"""Model registry (reference ``deepfm/models/__init__.py:12-36``)."""

from __future__ import annotations

from typing import Dict, Type

from deepfm_amd.config import ExperimentConfig
from deepfm_amd.data.schema import DatasetSchema
from deepfm_amd.models.base import BaseCTRModel
from deepfm_amd.models.deepfm import DeepFM

MODEL_REGISTRY: Dict[str, Type[BaseCTRModel]] = {"deepfm": DeepFM}

try:  # the CIN / attention models register themselves once their kernels exist
    from deepfm_amd.models.xdeepfm import xDeepFM
    MODEL_REGISTRY["xdeepfm"] = xDeepFM
except ImportError:  # pragma: no cover
    pass
try:
    from deepfm_amd.models.attention_deepfm import AttentionDeepFM
    MODEL_REGISTRY["attention_deepfm"] = AttentionDeepFM
except ImportError:  # pragma: no cover
    pass


def create_model(name: str, schema: DatasetSchema, config: ExperimentConfig) -> BaseCTRModel:
    if name not in MODEL_REGISTRY:
        raise ValueError(f"Unknown model: {name}. Choose from {list(MODEL_REGISTRY)}")
    return MODEL_REGISTRY[name](schema, config)


__all__ = ["BaseCTRModel", "DeepFM", "MODEL_REGISTRY", "create_model"]
