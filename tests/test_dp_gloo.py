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


# --------------------------------------------------------------------------------------------------
# Field-sharded tables (training/sharded.py): the three all-to-alls with the segment layouts of
# csrc/shard.hip, restated in numpy, world_size 2 over gloo.  Every rank must see exactly the rows of
# its own batch, and every owner must end up with the whole-batch reduction of its fields.
# --------------------------------------------------------------------------------------------------
SH_S, SH_D, SH_V, SH_B, SH_N = 3, 4, 30, 16, 10      # fields, dim, vocabulary, batch per rank, dense gradient size


def _sharded_inputs(world):
    rng = np.random.default_rng(5)
    tables = rng.standard_normal((SH_S, SH_V, SH_D)).astype(np.float32)
    first = rng.standard_normal((SH_S, SH_V)).astype(np.float32)
    ids = rng.integers(0, SH_V, size=(world, SH_S, SH_B)).astype(np.int64)
    g_fe = rng.standard_normal((world, SH_B, SH_S, SH_D)).astype(np.float32)
    g_first = rng.standard_normal((world, SH_B)).astype(np.float32)
    dense = rng.standard_normal((world, SH_N)).astype(np.float32)
    return tables, first, ids, g_fe, g_first, dense


def _sharded_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepfm_amd.training import exchange
    from deepfm_amd.training.sharded import FieldShards
    tables, first, ids, g_fe, g_first, dense = _sharded_inputs(world)
    sh = FieldShards(SH_S, world)
    B, D, nf, lo = SH_B, SH_D, sh.count[rank], sh.first[rank]
    # ids: the (S, B) block of the batch record is the send buffer, split by owner
    ids_recv = torch.empty(world * nf * B, dtype=torch.int64)
    exchange.all_to_all(ids_recv, torch.from_numpy(ids[rank].reshape(-1).copy()), [nf * B] * world,
                        [c * B for c in sh.count])
    gid = ids_recv.numpy().reshape(world, nf, B)
    # owner: rows of the owned tables in the send layout [p][e (B, nf, D) | w (B, nf)]
    send = np.zeros((world, B * nf * (D + 1)), np.float32)
    for p in range(world):
        e = np.stack([tables[lo + j][gid[p, j]] for j in range(nf)], axis=1)          # (B, nf, D)
        w = np.stack([first[lo + j][gid[p, j]] for j in range(nf)], axis=1)           # (B, nf)
        send[p] = np.concatenate([e.reshape(-1), w.reshape(-1)])
    rows_recv = torch.empty(sum(B * c * (D + 1) for c in sh.count))
    exchange.all_to_all(rows_recv, torch.from_numpy(send.reshape(-1)), [B * c * (D + 1) for c in sh.count],
                        [B * nf * (D + 1)] * world)
    fe = np.zeros((B, SH_S, D), np.float32)
    fo = np.zeros((B, SH_S), np.float32)
    off, rr = 0, rows_recv.numpy()
    for p in range(world):
        c = sh.count[p]
        fe[:, sh.first[p]:sh.first[p] + c] = rr[off:off + B * c * D].reshape(B, c, D)
        fo[:, sh.first[p]:sh.first[p] + c] = rr[off + B * c * D:off + B * c * (D + 1)].reshape(B, c)
        off += B * c * (D + 1)
    for s in range(SH_S):                                    # exactly the rows a local lookup would give
        assert np.array_equal(fe[:, s], tables[s][ids[rank, s]]) and np.array_equal(fo[:, s], first[s][ids[rank, s]])
    # gradients: segment for owner q = [d e of q's fields (B, nf_q, D) | d first (B) | dense (n)]
    seg = lambda c: B * c * D + B + SH_N                     # noqa: E731
    gs = np.concatenate([np.concatenate([g_fe[rank][:, sh.first[q]:sh.first[q] + sh.count[q]].reshape(-1),
                                         g_first[rank], dense[rank]]) for q in range(world)])
    grad_recv = torch.empty(world * seg(nf))
    exchange.all_to_all(grad_recv, torch.from_numpy(gs), [seg(nf)] * world, [seg(c) for c in sh.count])
    gr = grad_recv.numpy().reshape(world, seg(nf))
    g2 = gr[:, :B * nf * D].reshape(world * B, nf, D)        # sample p * B + b of the global batch
    g1 = gr[:, B * nf * D:B * nf * D + B].reshape(world * B)
    dsum = np.zeros(SH_N, np.float32)
    for p in range(world):                                   # rank-ordered mean, as the prepare launch forms it
        dsum += gr[p, B * nf * D + B:]
    out = {"dense": dsum / np.float32(world)}
    for j in range(nf):
        gids = gid[:, j].reshape(-1)
        u, a2, a1 = O.rowsparse_reduce_fast(gids, g2[:, j], g1)
        out[f"u{lo + j}"], out[f"g{lo + j}"], out[f"f{lo + j}"] = u, a2, a1
    part = torch.tensor([float(rank + 1), 2.0 * (rank + 1)])
    gathered = torch.empty(2 * world)
    exchange.all_gather_flat(gathered, part)
    assert gathered.tolist() == [v for r in range(world) for v in (float(r + 1), 2.0 * (r + 1))]
    np.savez(os.path.join(out_dir, f"shard{rank}.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_field_sharded_exchange_matches_global_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    tables, first, ids, g_fe, g_first, dense = _sharded_inputs(world)
    from deepfm_amd.training.sharded import FieldShards
    sh = FieldShards(SH_S, world)
    r = [np.load(tmp_path / f"shard{k}.npz") for k in range(world)]
    assert np.array_equal(r[0]["dense"], r[1]["dense"])                       # bit-identical dense replicas
    np.testing.assert_allclose(r[0]["dense"], dense.mean(axis=0), rtol=1e-6)
    all_ids = np.concatenate([ids[k] for k in range(world)], axis=1)          # (S, world * B)
    all_g = np.concatenate([g_fe[k] for k in range(world)], axis=0)
    all_g1 = np.concatenate([g_first[k] for k in range(world)])
    for s in range(SH_S):
        owner, _ = sh.owner(s)
        assert f"u{s}" in r[owner].files and f"u{s}" not in r[1 - owner].files   # every field has ONE owner
        u, a2, a1 = O.rowsparse_reduce_fast(all_ids[s], all_g[:, s], all_g1)
        assert np.array_equal(r[owner][f"u{s}"], u)
        np.testing.assert_allclose(r[owner][f"g{s}"], a2, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(r[owner][f"f{s}"], a1, rtol=1e-6, atol=1e-6)
