#!/usr/bin/env python3
"""Event-timed exact-fp32 tower GEMM launches at first-layer shapes: forward, d weight, d input, both.
usage: python tools/time_tower_f32.py [iters=100]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd import _lib  # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    lib = _lib.load()
    st = _lib.stream_handle()
    B = 4096
    out = []
    for n, k in ((256, 624), (256, 2496), (128, 256)):
        dz = torch.randn(B, n, device="cuda")
        x = torch.randn(B, k, device="cuda")
        w = torch.randn(n, k, device="cuda")
        z = torch.empty(B, n, device="cuda")
        gx = torch.empty(B, k, device="cuda")
        wsf = torch.zeros(lib.dfm_linear_bn_workspace_bytes(B, n) // 4, device="cuda")
        ws = torch.zeros(max(lib.dfm_linear_backward_workspace_bytes(B, n, k) // 4, 1), device="cuda")
        t = [timed(lambda: lib.dfm_linear_bn_forward(x.data_ptr(), k, w.data_ptr(), None, B, n, k, z.data_ptr(), wsf.data_ptr(), st), iters)]
        for parts in (1, 2, 3):
            t.append(timed(lambda: lib.dfm_linear_backward(dz.data_ptr(), B, n, x.data_ptr(), k, w.data_ptr(), gx.data_ptr(), None,
                                                           None, parts, ws.data_ptr(), st), iters))
        out.append(f"{k}->{n}: fwd {t[0]:.1f} dW {t[1]:.1f} dX {t[2]:.1f} both {t[3]:.1f}")
    print(" | ".join(out), flush=True)


if __name__ == "__main__":
    main()
