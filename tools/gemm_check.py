#!/usr/bin/env python3
"""Check + time dfm_gemm_f32 on the DNN-tower shapes (all three Linear GEMMs per layer).
usage: python tools/gemm_check.py   (DFM_LIB_PATH picks another build)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd.models.layers.dnn import _gemm  # noqa: E402


def run(M, N, K, a_kc, b_kc, acc=False, iters=50):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    b = torch.randn(N, K, device="cuda", generator=g)
    want = (a.double() @ b.double().t()).float()
    A = a if a_kc else a.t().contiguous()
    Bm = b if b_kc else b.t().contiguous()
    c = torch.zeros(M, N, device="cuda")
    f = lambda: _gemm(A, K if a_kc else M, a_kc, Bm, K if b_kc else N, b_kc, c, M, N, K)
    f()
    err = (c - want).abs()
    bad = (err > 1e-4 * want.abs() + 1e-4).sum().item()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / iters * 1e6
    fl = 2.0 * M * N * K
    where = ""
    if bad:
        idx = (err > 1e-4 * want.abs() + 1e-4).nonzero()
        where = f" rows {idx[:,0].min().item()}..{idx[:,0].max().item()} cols {idx[:,1].min().item()}..{idx[:,1].max().item()} first {idx[:5].tolist()}"
    print(f"M{M} N{N} K{K} akc={int(a_kc)} bkc={int(b_kc)}: maxerr {err.max().item():.2e} bad {bad} {us:7.1f} us {fl / us / 1e6:6.1f} TF{where}", flush=True)


B = 4096
for (i, o) in ((624, 256), (256, 128), (128, 64), (64, 1), (2496, 256)):
    run(B, o, i, True, True)       # forward
    run(B, i, o, True, False)      # d input
    run(o, i, B, False, False)     # d weight
