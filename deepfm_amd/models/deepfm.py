"""DeepFM = first-order + FM second-order + DNN (reference ``deepfm/models/deepfm.py:13-42``)."""

from __future__ import annotations

import torch
import torch.nn as nn

from deepfm_amd.models.base import BaseCTRModel
from deepfm_amd.models.layers.dnn import DNN
from deepfm_amd.models.layers.linear import MfmaLinear
from deepfm_amd.models.layers.fm import FMInteraction


class DeepFM(BaseCTRModel):
    def _build_components(self) -> None:
        c = self.config.dnn
        self.fm = FMInteraction()
        self.dnn = DNN(self.schema.total_embedding_dim, c.hidden_units, c.activation, c.dropout,
                       c.use_batch_norm)
        self.output_linear = MfmaLinear(self.dnn.output_dim, 1)

    def _forward_components(self, first_order, field_embeddings, flat_embeddings) -> torch.Tensor:
        deep = self.output_linear(self.dnn(flat_embeddings))
        return first_order + self.fm(field_embeddings) + deep
