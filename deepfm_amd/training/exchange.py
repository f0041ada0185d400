"""Data-parallel gradient exchange of the row-sparse step (one process per GPU, tables replicated).

Two collectives per step, nothing else crosses ranks:
  * ``allreduce_flat``     — one all-reduce(SUM) of the flat dense-gradient buffer;
  * ``allgather_row_lists`` — all-gather of every rank's row lists (distinct ids, their count,
    one gradient row + one first-order scalar per id).  The result is rank-major, so list
    ``l = rank * chunks + chunk``; every rank then runs the same deterministic merge
    (csrc/rowadam.hip: the first list holding a row owns it and adds the other lists'
    rows in list order), which keeps the replicas bit-identical without atomics.
On the GPU the backend is RCCL ("nccl") over xGMI; the same code runs on gloo/CPU in the tests.
"""

from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world_size(group: Optional[dist.ProcessGroup] = None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_flat(flat_grad: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> None:
    """In-place SUM over ranks (the caller scales by 1/world: the loss is a per-rank mean)."""
    if world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)


def alloc_gathered(local: Sequence[torch.Tensor], world: int) -> Tuple[torch.Tensor, ...]:
    return tuple(torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                 for t in local)


def allgather_row_lists(local: Sequence[torch.Tensor], out: Sequence[torch.Tensor],
                        group: Optional[dist.ProcessGroup] = None) -> None:
    """local[i] has shape (chunks, ...); out[i] (world*chunks, ...) receives rank-major copies."""
    for dst, src in zip(out, local):
        dist.all_gather_into_tensor(dst, src.contiguous(), group=group)
