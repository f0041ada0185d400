"""Data-parallel gradient exchange of the row-sparse step (one process per GPU, tables replicated).

ONE (coalesced) all-gather per step, nothing else crosses ranks:
  * ``allgather_step`` — the flat dense-gradient buffer and every rank's row lists (distinct ids,
    their count, one gradient row + one first-order scalar per id) in a single grouped launch.  The
    results are rank-major, so list ``l = rank * chunks + chunk``; every rank then runs the same
    deterministic merge (csrc/tail_bodies.h: the first list holding a row owns it and adds the other
    lists' rows in list order) and the same rank-ordered sum of the dense gradients
    (csrc/step_tail.hip), which keeps the replicas bit-identical without atomics.
  * ``allreduce_flat`` / ``allgather_row_lists`` — the same exchange as separate collectives (kept
    for backends without coalescing and for the CPU tests).
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


_GROUPED: dict = {}     # process group -> bool: the choice made (and logged) on first use


def _grouped_supported(group: Optional[dist.ProcessGroup]) -> bool:
    """One grouped launch needs the backend's coalescing support: RCCL ("nccl") has it (ncclGroupStart /
    ncclGroupEnd around the all-gathers), gloo does not.  Decided once per group from the backend NAME —
    never by catching an error of a collective — and logged, so a slower exchange is never silent."""
    key = group if group is not None else "default"
    if key not in _GROUPED:
        backend = dist.get_backend(group)
        _GROUPED[key] = backend == "nccl"
        import logging
        logging.getLogger("deepfm_amd.exchange").info(
            "gradient exchange over %s: %s", backend,
            "ONE grouped all-gather per step" if _GROUPED[key] else "one all-gather per tensor (no coalescing support)")
    return _GROUPED[key]


def allgather_step(local: Sequence[torch.Tensor], out: Sequence[torch.Tensor],
                   group: Optional[dist.ProcessGroup] = None) -> None:
    """All tensors of ``local`` all-gathered into ``out`` (``out[i]`` is ``(world * local[i].shape[0], ...)``)
    as ONE grouped collective launch on RCCL, one collective per tensor on backends without coalescing
    (gloo: the CPU tests).  Every tensor must be contiguous.  Errors of the collectives propagate."""
    if _grouped_supported(group):
        from torch.distributed import distributed_c10d as c10d
        with c10d._coalescing_manager(group=group):      # torch's only handle on ncclGroupStart/End
            for dst, src in zip(out, local):
                dist.all_gather_into_tensor(dst, src, group=group)
    else:
        for dst, src in zip(out, local):
            dist.all_gather_into_tensor(dst, src, group=group)


_LOGGED: set = set()


def _log_once(key, message: str, *args) -> None:
    if key not in _LOGGED:
        _LOGGED.add(key)
        import logging
        logging.getLogger("deepfm_amd.exchange").info(message, *args)


def all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits: Sequence[int], in_splits: Sequence[int],
               group: Optional[dist.ProcessGroup] = None) -> None:
    """``inp`` = one contiguous segment per peer (``in_splits[q]`` elements for rank q), ``out`` = one
    per source rank (``out_splits[p]`` from rank p): the three exchanges of the field-sharded step
    (training/sharded.py).  RCCL: one grouped send/recv launch (``all_to_all_single``), capturable in a
    HIP graph.  One rank without a process group: a device copy.  gloo has no all-to-all on device
    buffers: the CPU/one-GPU rehearsals stage through host memory (logged; never the RCCL path)."""
    if not (dist.is_available() and dist.is_initialized()):
        if list(out_splits) != list(in_splits) or len(in_splits) != 1:
            raise RuntimeError("all_to_all without a process group: one rank only")
        out.copy_(inp)
        return
    if dist.get_backend(group) != "nccl" and inp.is_cuda:
        _log_once(("a2a-host", group), "all-to-all over %s: staged through host memory (rehearsal only)",
                  dist.get_backend(group))
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host_out, inp.cpu(), list(out_splits), list(in_splits), group=group)
        out.copy_(host_out)
        return
    dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=group)


def all_gather_flat(out: torch.Tensor, mine: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> None:
    """``out[r * n : (r + 1) * n] = mine`` of rank r (``mine``: n contiguous elements).  One rank without a
    process group: a copy."""
    if not (dist.is_available() and dist.is_initialized()):
        out.copy_(mine)
        return
    dist.all_gather_into_tensor(out, mine, group=group)
