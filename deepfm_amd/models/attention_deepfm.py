"""AttentionDeepFM = first-order + FM + DNN(cat[attention(fe).flatten(), flat])
(reference ``deepfm/models/attention_deepfm.py:14-66``; FM uses the un-attended embeddings)."""

from __future__ import annotations

import torch
import torch.nn as nn

from deepfm_amd.models.base import BaseCTRModel
from deepfm_amd.models.layers.attention import MultiHeadSelfAttention
from deepfm_amd.models.layers.dnn import DNN
from deepfm_amd.models.layers.linear import MfmaLinear
from deepfm_amd.models.layers.fm import FMInteraction


class AttentionDeepFM(BaseCTRModel):
    def _build_components(self) -> None:
        c = self.config
        fm_dim = c.feature.fm_embed_dim
        self.fm = FMInteraction()
        self.attention = MultiHeadSelfAttention(fm_dim, c.attention.num_heads, c.attention.attention_dim,
                                                c.attention.num_layers, c.attention.use_residual)
        dnn_in = self.schema.num_fields * fm_dim + self.schema.total_embedding_dim
        self.dnn = DNN(dnn_in, c.dnn.hidden_units, c.dnn.activation, c.dnn.dropout, c.dnn.use_batch_norm)
        self.output_linear = MfmaLinear(self.dnn.output_dim, 1)

    def _forward_components(self, first_order, field_embeddings, flat_embeddings) -> torch.Tensor:
        refined = self.attention(field_embeddings)
        dnn_in = torch.cat([refined.reshape(refined.size(0), -1), flat_embeddings], dim=1)
        return first_order + self.fm(field_embeddings) + self.output_linear(self.dnn(dnn_in))
