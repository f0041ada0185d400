"""BaseCTRModel: embedding -> model-specific components -> raw logits.

Mirror of the reference's ``deepfm/models/base.py:15-83`` (constructor, the two
abstract hooks, ``forward``/``predict``/``get_l2_reg_loss`` and the ``embedding``
attribute), built on the HIP-backed ``FeatureEmbedding``.
"""

from __future__ import annotations

import abc
from typing import Dict

import torch
import torch.nn as nn

from deepfm_amd.config import ExperimentConfig
from deepfm_amd.data.schema import DatasetSchema
from deepfm_amd.models.layers.embedding import FeatureEmbedding


class BaseCTRModel(nn.Module, abc.ABC):
    def __init__(self, schema: DatasetSchema, config: ExperimentConfig) -> None:
        super().__init__()
        self.schema = schema
        self.config = config
        self.embedding = FeatureEmbedding(schema, fm_embed_dim=config.feature.fm_embed_dim)
        self._build_components()

    @abc.abstractmethod
    def _build_components(self) -> None:
        """Create FM / CIN / attention / DNN / heads."""

    @abc.abstractmethod
    def _forward_components(self, first_order: torch.Tensor, field_embeddings: torch.Tensor,
                            flat_embeddings: torch.Tensor) -> torch.Tensor:
        """(B,1), (B,F,fm_dim), (B,sum d) -> raw logits (B,1) (no sigmoid: BCEWithLogits)."""

    def forward(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self._forward_components(*self.embedding(batch))

    def predict(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        return torch.sigmoid(self.forward(batch))

    def get_l2_reg_loss(self) -> torch.Tensor:
        """lambda * sum ||p||^2 over the embedding parameters (base.py:78-83).

        In ``rowsparse`` gradient mode the (V, d) tables are not part of autograd: their
        L2 term is applied lazily to the touched rows by ``RowSparseAdam(l2=...)``, and
        this method covers the remaining embedding parameters (DENSE-field Linears)."""
        emb = self.embedding
        params = emb.non_table_parameters() if emb.grad_mode == "rowsparse" else list(emb.parameters())
        total = torch.zeros((), device=next(self.parameters()).device)
        for p in params:
            total = total + p.pow(2).sum()
        return self.config.feature.embedding_l2_reg * total
