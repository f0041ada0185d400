"""Checkpoint save/load in the reference's format (``deepfm/utils/io.py:17-26``; written by
``Trainer.fit``, trainer.py:140-148): one ``torch.save`` of

    {"epoch", "model_state_dict", "optimizer_state_dict", "best_metric"}

with the reference's ``state_dict`` keys and shapes.  ``model_state_dict`` moves between the reference
and this package in both directions.  ``optimizer_state_dict``: ``RowSparseAdam.load_state_dict`` reads
both its own name-keyed format and the ``torch.optim.Adam.state_dict()`` the reference writes
(trainer.py:144); what this package writes is its own format (torch's Adam cannot load it).  Two
differences, both on the safe side:

* tensors are made contiguous first: with packed row records (``FeatureEmbedding.pack_tables_``) a
  table is a strided view of a 4x larger buffer, and ``torch.save`` of a view writes the whole
  underlying storage;
* loading uses ``weights_only=True`` (the reference passes ``weights_only=False``): a checkpoint
  holds tensors, numbers, strings, lists and dicts only, and nothing from the file is executed.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, Union

import torch


def _detach_tree(obj: Any) -> Any:
    if isinstance(obj, torch.Tensor):
        return obj.detach().to("cpu").contiguous().clone()
    if isinstance(obj, dict):
        return {k: _detach_tree(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_detach_tree(v) for v in obj)
    return obj


def save_checkpoint(state: dict, path: Union[str, Path]) -> None:
    """Save a checkpoint dict to disk (CPU, contiguous tensors)."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(_detach_tree(state), path)


def load_checkpoint(path: Union[str, Path], device: str = "cpu") -> dict:
    """Load a checkpoint dict from disk without executing anything from the file."""
    return torch.load(path, map_location=device, weights_only=True)
