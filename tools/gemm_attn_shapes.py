#!/usr/bin/env python3
"""dfm_gemm_f32 on the attention-projection shapes of Cfg4 (B*F = 159 744 rows, D = 32, A = 64):
time, TFLOP/s and the HBM-traffic floor of each.  usage: python tools/gemm_attn_shapes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_check import run  # noqa: E402  (prints the DNN shapes first)

R = 4096 * 39
print("attention projections:")
run(R, 192, 32, True, True)      # qkv = x Wqkv^T
run(R, 32, 64, True, True)       # out = o Wo^T
run(R, 32, 192, True, False)     # d x = d qkv Wqkv
run(192, 32, R, False, False)    # d Wqkv = d qkv^T x
run(R, 64, 32, True, False)      # d o = d out Wo
run(32, 64, R, False, False)     # d Wo = d out^T o

# accumulate + bias variants against torch (fp64 reference)
import torch
from deepfm_amd.models.layers.dnn import _gemm
g = torch.Generator(device="cuda").manual_seed(5)
for (N, K, kc) in ((192, 32, True), (32, 64, True), (32, 192, False), (64, 32, False)):
    a = torch.randn(R - 7, K, device="cuda", generator=g)            # ragged row count
    w = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    c0 = torch.randn(R - 7, N, device="cuda", generator=g)
    c = c0.clone()
    W = w if kc else w.t().contiguous()
    _gemm(a, K, True, W, K if kc else N, kc, c, R - 7, N, K, bias=bias, accumulate=True)
    want = (c0.double() + a.double() @ w.double().t() + bias.double()).float()
    print(f"rows N{N} K{K} kc={int(kc)} +bias +acc: maxerr {(c - want).abs().max().item():.2e}")
