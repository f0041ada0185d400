"""DNN tower (reference ``deepfm/models/layers/dnn.py:9-59``).

Not one of the four hot-path layers (SURVEY.md §2 keeps it on ``torch.nn`` →
rocBLAS/hipBLASLt); it is mirrored so the three models assemble with the reference's
``dnn.mlp.<i>`` state_dict layout: ``[Linear, (BatchNorm1d), activation, Dropout] * n``.
"""

from __future__ import annotations

from typing import List

import torch
import torch.nn as nn


class DNN(nn.Module):
    ACTIVATIONS = {"relu": nn.ReLU, "leaky_relu": nn.LeakyReLU, "gelu": nn.GELU, "tanh": nn.Tanh}

    def __init__(self, input_dim: int, hidden_units: List[int], activation: str = "relu",
                 dropout: float = 0.1, use_batch_norm: bool = True) -> None:
        super().__init__()
        if not hidden_units:
            raise ValueError("hidden_units must be non-empty")
        try:
            act = self.ACTIVATIONS[activation.lower()]
        except KeyError:
            raise ValueError(f"Unknown activation: {activation}. Choose from {list(self.ACTIVATIONS)}") from None
        stack: List[nn.Module] = []
        width = input_dim
        for units in hidden_units:
            stack.append(nn.Linear(width, units))
            if use_batch_norm:
                stack.append(nn.BatchNorm1d(units))
            stack += [act(), nn.Dropout(p=dropout)]
            width = units
        self.mlp = nn.Sequential(*stack)
        self.output_dim = width

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.mlp(x)
