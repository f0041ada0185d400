#!/usr/bin/env python3
"""Whole-model training steps at the BASELINE.json configurations 2-4 (B = 4096, V = 1e6):
DeepFM (fused tower step), xDeepFM (CIN [128,128,128]) and AttentionDeepFM (embed_dim 32, 4 heads) on
the autograd step, all with row-sparse Adam, packed tables and HIP graphs.
usage: python tools/time_models.py [steps] [model ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd.config import ExperimentConfig  # noqa: E402
from deepfm_amd.models import create_model  # noqa: E402
from deepfm_amd.training.fused_step import fused_step_class  # noqa: E402
from deepfm_amd.training.rowsparse import RowSparseAdam  # noqa: E402
from deepfm_amd.training.step import RowSparseTrainStep  # noqa: E402
from deepfm_amd.data.synthetic import schema_from_fields  # noqa: E402
from tools_shared import criteo_fields  # noqa: E402


def run(name, dim, steps, B=4096, V=1_000_000):
    cfg = ExperimentConfig()
    cfg.feature.fm_embed_dim = dim
    if name == "xdeepfm":
        cfg.cin.layer_sizes = [128, 128, 128]
    fields = criteo_fields(V, dim)
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = create_model(name, schema_from_fields(fields), cfg)
    model.train()
    model.embedding.pack_tables_()
    model.embedding.set_grad_mode("rowsparse")
    opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
    cls = fused_step_class(model) or RowSparseTrainStep
    step = cls(model, opt, B, use_graph=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    n = 16
    ids = torch.randint(1, V, (n, 26, B), generator=g, device="cuda", dtype=torch.int64)
    dense = torch.rand((n, 13, B), generator=g, device="cuda")
    labels = (torch.rand((n, B), generator=g, device="cuda") < 0.25).float()
    recs = step.pack_batches(ids, dense, labels)
    step.load_packed(recs[0])
    step.capture()
    for i in range(10):
        step.run_from(recs[i % n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step.run_from(recs[i % n])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{name:18s} embed_dim {dim:2d}: {ms:7.3f} ms/step  {B / ms * 1e3 / 1e6:6.2f} M samples/s  "
          f"({cls.__name__}, loss {float(step.loss):.4f})", flush=True)
    del step, opt, model
    torch.cuda.empty_cache()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    which = sys.argv[2:] or ["deepfm", "xdeepfm", "attention_deepfm"]
    for name in which:
        run(name, 32 if name == "attention_deepfm" else 16, steps)
