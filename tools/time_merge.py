#!/usr/bin/env python3
"""Cost of the optimizer tail (merge + clip + row-wise Adam + dense Adam) when every replica applies the
row lists of N data-parallel ranks — the part of the N-GPU step that grows with N (tables are
replicated).  One GPU: the N lists are built locally from N random batches.
usage: python tools/time_merge.py [N ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd.config import ExperimentConfig  # noqa: E402
from deepfm_amd.models import create_model  # noqa: E402
from deepfm_amd.training.rowsparse import RowSparseAdam  # noqa: E402
from deepfm_amd.data.synthetic import schema_from_fields  # noqa: E402
from tools_shared import criteo_fields  # noqa: E402


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    B, V, D = 4096, 1_000_000, 16
    fields = criteo_fields(V, D)
    cfg = ExperimentConfig()
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = create_model("deepfm", schema_from_fields(fields), cfg)
    model.train()
    emb = model.embedding
    emb.pack_tables_()
    emb.set_grad_mode("rowsparse")
    opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
    g = torch.Generator(device="cuda").manual_seed(1)
    for world in worlds:
        parts = []
        for r in range(world):
            ids = torch.randint(1, V, (26, B), generator=g, device="cuda", dtype=torch.int64)
            dense = torch.rand((13, B), generator=g, device="cuda")
            inputs = [ids[i] for i in range(26)] + [dense[i] for i in range(13)]
            emb._ensure_plan(inputs[0].device)
            rs = emb.build_rowplan(inputs, B)
            g_fe = torch.randn(B, 39, D, device="cuda", generator=g) * 1e-3
            g_fo = torch.randn(B, 1, device="cuda", generator=g) * 1e-3
            emb.backward_rowsparse(inputs, g_fo, g_fe, {})
            parts.append([t.clone() for t in (rs.uniq_rows, rs.num_uniq, rs.row_g2, rs.row_g1)])
        gathered = [torch.cat([p[i] for p in parts], dim=0) for i in range(4)]
        opt.world = world
        pristine = [t.clone() for t in gathered]

        def once():
            for dst, src in zip(gathered, pristine):      # merge rewrites the gradient rows in place
                dst.copy_(src)
            opt._cur = tuple(gathered) + (world,)
            opt.apply()

        for _ in range(3):
            once()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            for dst, src in zip(gathered, pristine):
                dst.copy_(src)
            opt._cur = tuple(gathered) + (world,)
            a.record()
            opt.apply()
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"lists of {world} rank(s): optimizer tail {ms[len(ms) // 2] * 1e3:.1f} us (median of 20, "
              f"three launches), {world * 26 * B} row entries", flush=True)
    opt.world = 1


if __name__ == "__main__":
    main()
