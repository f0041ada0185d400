"""xDeepFM = first-order + CIN + DNN (reference ``deepfm/models/xdeepfm.py:13-48``; no FM term)."""

from __future__ import annotations

import torch
import torch.nn as nn

from deepfm_amd.models.base import BaseCTRModel
from deepfm_amd.models.layers.cin import CIN
from deepfm_amd.models.layers.dnn import DNN
from deepfm_amd.models.layers.linear import MfmaLinear


class xDeepFM(BaseCTRModel):
    def _build_components(self) -> None:
        c = self.config
        self.cin = CIN(self.schema.num_fields, c.feature.fm_embed_dim, c.cin.layer_sizes, c.cin.split_half)
        self.dnn = DNN(self.schema.total_embedding_dim, c.dnn.hidden_units, c.dnn.activation, c.dnn.dropout,
                       c.dnn.use_batch_norm)
        self.cin_linear = MfmaLinear(self.cin.output_dim, 1)
        self.dnn_linear = MfmaLinear(self.dnn.output_dim, 1)

    def _forward_components(self, first_order, field_embeddings, flat_embeddings) -> torch.Tensor:
        explicit = self.cin_linear(self.cin(field_embeddings))
        implicit = self.dnn_linear(self.dnn(flat_embeddings))
        return first_order + explicit + implicit
