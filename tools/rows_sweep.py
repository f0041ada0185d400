#!/usr/bin/env python3
"""Time of the many-rows GEMM shapes of the attention projections (HIP events around 20 back-to-back
calls; below ~15 us the Python call rate, not the kernel, is what is measured — use rocprofv3 then).
usage: python tools/rows_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd.models.layers.dnn import _gemm
R = 4096 * 39
g = torch.Generator(device="cuda").manual_seed(0)
for (N, K, kc, acc) in ((192, 32, True, False), (32, 64, True, False), (32, 192, False, True), (64, 32, False, False)):
    a = torch.randn(R, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    W = w if kc else w.t().contiguous()
    c = torch.zeros(R, N, device="cuda")
    f = lambda: _gemm(a, K, True, W, K if kc else N, kc, c, R, N, K, accumulate=acc)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    mb = (R * K + R * N * (2 if acc else 1)) * 4 / 1e6
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"N{N} K{K}: {us:6.1f} us  {mb / us:5.2f} TB/s  ({mb:.0f} MB, floor {mb / 8.0:.1f} us at 8 TB/s)")
