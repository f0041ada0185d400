#!/usr/bin/env python3
"""Time the CIN / attention layers at the BASELINE.json config-3/4 shapes (B=4096) on the GPU.
usage: python tools/time_layers.py [cin|attn] [iters] [split|bf16|fp32]   (third argument: CIN arithmetic, dfm_cin_set_mode)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bench(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "cin"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    B = int(os.environ.get("B", 4096))
    mode = sys.argv[3] if len(sys.argv) > 3 else "split"
    from deepfm_amd import _lib
    _lib.check(_lib.load().dfm_cin_set_mode({"split": 0, "bf16": 1, "fp32": 2}[mode]))
    torch.manual_seed(0)
    if what == "cin":
        from deepfm_amd.models.layers.cin import CIN
        layer = CIN(39, 16, [128, 128, 128], True).cuda()
        x = (torch.randn(B, 39, 16, device="cuda") * 0.5).requires_grad_()
        flops_f, flops_b = 26.68e6 * B, 53.35e6 * B
    else:
        from deepfm_amd.models.layers.attention import MultiHeadSelfAttention
        layer = MultiHeadSelfAttention(32, 4, 64, 1, True).cuda()
        x = torch.randn(B, 39, 32, device="cuda").requires_grad_()
        flops_f, flops_b = 1.03e6 * B, 2.5e6 * B
    with torch.no_grad():
        f = bench(lambda: layer(x), iters)

    def fb():
        out = layer(x)
        out.sum().backward()
    t = bench(fb, max(iters // 2, 2))
    print(f"{what} mode={mode} B={B}: fwd {f:.3f} ms ({flops_f / f / 1e9:.1f} TFLOP/s alg), "
          f"fwd+bwd {t:.3f} ms ({(flops_f + flops_b) / t / 1e9:.1f} TFLOP/s alg)")


if __name__ == "__main__":
    main()
