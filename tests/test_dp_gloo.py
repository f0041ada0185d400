"""CPU, world_size 2, gloo: the data-parallel exchange of the row-sparse step.

Two processes each reduce their half of a global batch to row lists (with the oracle), run
the package's exchange functions (the same code that runs on RCCL), merge with a numpy
restatement of the ownership rule of csrc/rowadam.hip, and must end up (a) bit-identical to
each other and (b) equal to a single-process reduction of the whole batch.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ctr_oracle as O

CH = 64          # list capacity used by this CPU test (the HIP path uses 4096)
S, D, V, B = 3, 8, 40, 48


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_lists(ids, g2, g1):
    """(chunks=1, S, CH[, D]) buffers like RowSparseBuffers, from the oracle's ordered reduction."""
    uniq = np.zeros((1, S, CH), np.int32)
    num = np.zeros((1, S), np.int32)
    r2 = np.zeros((1, S, CH, D), np.float32)
    r1 = np.zeros((1, S, CH), np.float32)
    for s in range(S):
        u, a2, a1 = O.rowsparse_from_batch(ids[s], g2[:, s, :], g1)
        n = len(u)
        uniq[0, s, :n], num[0, s], r2[0, s, :n], r1[0, s, :n] = u, n, a2, a1
    return uniq, num, r2, r1


def _merge(uniq, num, r2, r1, scale):
    """Ownership merge: the first list holding a row owns it; others are added in list order."""
    L = uniq.shape[0]
    merged = [dict() for _ in range(S)]
    for s in range(S):
        for l in range(L):
            for u in range(num[l, s]):
                row = int(uniq[l, s, u])
                if row in merged[s]:
                    continue
                a2, a1 = r2[l, s, u].copy(), np.float32(r1[l, s, u])
                for l2 in range(l + 1, L):
                    pos = np.searchsorted(uniq[l2, s, :num[l2, s]], row)
                    if pos < num[l2, s] and uniq[l2, s, pos] == row:
                        a2 = (a2 + r2[l2, s, pos]).astype(np.float32)
                        a1 = np.float32(a1 + r1[l2, s, pos])
                merged[s][row] = ((np.float32(scale) * a2).astype(np.float32), np.float32(scale) * a1)
    return merged


def _worker(rank, world, port, ids, g2, g1, flat, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepfm_amd.training import exchange
    half = B // world
    sl = slice(rank * half, (rank + 1) * half)
    local = tuple(torch.from_numpy(a) for a in _local_lists(ids[:, sl], g2[sl], g1[sl]))
    gathered = exchange.alloc_gathered(local, world)
    exchange.allgather_row_lists(local, gathered)
    fg = torch.from_numpy(flat[rank].copy())
    exchange.allreduce_flat(fg)
    assert exchange.world_size() == world
    # the step's single grouped all-gather (dense buffer + row lists) must deliver the same bytes
    full = (torch.from_numpy(flat[rank].copy()).view(1, -1),) + local
    gathered2 = exchange.alloc_gathered(full, world)
    exchange.allgather_step(full, gathered2)
    for a, b in zip(gathered, gathered2[1:]):
        assert torch.equal(a, b)
    assert torch.equal(gathered2[0], torch.from_numpy(flat[:world]))
    merged = _merge(*(t.numpy() for t in gathered), 1.0 / world)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=fg.numpy() / world,
             **{f"u{s}": np.array(sorted(merged[s])) for s in range(S)},
             **{f"g{s}": np.stack([merged[s][r][0] for r in sorted(merged[s])]) for s in range(S)},
             **{f"f{s}": np.array([merged[s][r][1] for r in sorted(merged[s])]) for s in range(S)})
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_exchange_matches_global_batch(tmp_path):
    rng = np.random.default_rng(0)
    ids = rng.integers(0, V, size=(S, B)).astype(np.int64)
    g2 = rng.standard_normal((B, S, D)).astype(np.float32)
    g1 = rng.standard_normal(B).astype(np.float32)
    flat = rng.standard_normal((2, 100)).astype(np.float32)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ids, g2, g1, flat, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), f"replicas diverged on {k}"      # bit-identical replicas
    np.testing.assert_allclose(r0["flat"], flat.mean(axis=0), rtol=1e-6)
    for s in range(S):                                                         # == whole-batch reduction / world
        u, a2, a1 = O.rowsparse_reduce_fast(ids[s], g2[:, s, :], g1)
        assert np.array_equal(r0[f"u{s}"], u)
        np.testing.assert_allclose(r0[f"g{s}"], a2 / 2, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r0[f"f{s}"], a1 / 2, rtol=1e-5, atol=1e-6)
