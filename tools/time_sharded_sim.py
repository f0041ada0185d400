#!/usr/bin/env python3
"""Device work of ONE rank of an N-rank field-sharded job, on one GPU (no second GPU exists here).

Rank `rank` of `world` is built for real — its table shard, the global-batch row plan, the gradient
segments of all N sources — and the three all-to-alls are replaced by local kernels that fill every
peer's receive segment with data of the right shape: the ids of the other ranks are this rank's ids
shifted (so the global batch has N times the distinct rows, as with independent minibatches), rows and
gradients are copies of the own segment.  What this measures: every kernel the rank runs per step at
N ranks (shard gather over N*B samples, row plan / row gradients / merge / row-wise Adam over the
global batch of its own fields) — NOT the wire time of the collectives, which DESIGN.md §6 adds from
link arithmetic.  Run under rocprofv3 --kernel-trace and sum the kernels that are not the stand-ins
(tools/kstats.py); the script itself prints the eager wall time per step (launch-bound, upper bound).

usage: python tools/time_sharded_sim.py [world=8] [rank=0] [steps=30]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd.config import ExperimentConfig  # noqa: E402
from deepfm_amd.models import create_model  # noqa: E402
from deepfm_amd.training import exchange  # noqa: E402
from deepfm_amd.training.sharded import ShardedRowAdam, TableShard, sharded_step_class  # noqa: E402
from deepfm_amd.data.synthetic import schema_from_fields  # noqa: E402
from tools_shared import criteo_fields  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    B, V, D = 4096, 1_000_000, 16
    cfg = ExperimentConfig()
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = create_model("deepfm", schema_from_fields(criteo_fields(V, D)), cfg)
    model.train()
    model.embedding.pack_tables_()
    model.embedding.set_grad_mode("rowsparse")
    shard = TableShard(model, rank, world)
    opt = ShardedRowAdam(model, shard, lr=cfg.training.lr, l2=cfg.feature.embedding_l2_reg,
                         max_grad_norm=cfg.training.gradient_clip_norm)
    step = sharded_step_class(model)(model, opt, B, use_graph=False)
    sh = shard.shards
    nf = sh.count[rank]
    shift = torch.arange(world, device="cuda", dtype=torch.int64).view(world, 1) * 7919
    # rows: source p's segment holds count[p] fields per sample; take the first count[p] of the own nf
    idx = []
    own = B * nf * (D + 1)
    for p in range(world):
        c = min(sh.count[p], nf)
        e = (torch.arange(B).view(B, 1, 1) * nf + torch.arange(sh.count[p]).view(1, -1, 1) % c) * D + torch.arange(D).view(1, 1, D)
        w = B * nf * D + torch.arange(B).view(B, 1) * nf + torch.arange(sh.count[p]).view(1, -1) % c
        idx.append(torch.cat([e.reshape(-1), w.reshape(-1)]) + rank * own)
    row_index = torch.cat(idx).cuda()

    def fake_all_to_all(out, inp, out_splits, in_splits, group=None):
        lo = sum(in_splits[:rank])
        mine = inp[lo:lo + in_splits[rank]]
        if inp.dtype == torch.int64:                                   # ids
            o = out.view(world, -1)
            torch.add(mine.view(1, -1), shift, out=o)
            o.remainder_(V)
        elif len(set(out_splits)) == 1:                                # gradients: equal segments
            out.view(world, -1).copy_(mine.view(1, -1).expand(world, -1))
        else:                                                          # rows
            torch.index_select(inp, 0, row_index, out=out)

    def fake_all_gather(out, mine, group=None):
        out.view(world, -1).copy_(mine.view(1, -1).expand(world, -1))

    exchange.all_to_all = fake_all_to_all
    exchange.all_gather_flat = fake_all_gather
    g = torch.Generator(device="cuda").manual_seed(1)
    n = steps + 5
    ids = torch.randint(1, V, (n, 26, B), generator=g, device="cuda", dtype=torch.int64)
    dense = torch.rand((n, 13, B), generator=g, device="cuda")
    labels = (torch.rand((n, B), generator=g, device="cuda") < 0.3).float()
    recs = step.pack_batches(ids, dense, labels)
    for i in range(5):
        step.run_from(recs[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, n):
        step.run_from(recs[i])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rs = shard.emb.rowsparse
    print(f"rank {rank} of {world}: owns {nf} fields, global batch {world * B}; eager {el / steps * 1e3:.3f} ms/step "
          f"(launch-bound upper bound), {int(rs.num_uniq.sum())} distinct rows updated per step, loss {float(step.loss):.4f}",
          flush=True)


if __name__ == "__main__":
    main()
