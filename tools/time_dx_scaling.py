#!/usr/bin/env python3
"""d-input product of dfm_linear_backward (parts = 2): time against the contraction length (out_features) and against the
number of workgroups (batch) — slope = cost of a slice, intercept = fixed cost of a workgroup round.
usage: python tools/time_dx_scaling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd import _lib  # noqa: E402


def timed(fn, iters=60):
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
    lib = _lib.load()
    st = _lib.stream_handle()
    k = 624
    for B in (1024, 1600, 2048, 3264, 4096, 8192):
        row = []
        for n in (64, 128, 256, 512, 1024):
            dz = torch.randn(B, n, device="cuda")
            x = torch.randn(B, k, device="cuda")
            w = torch.randn(n, k, device="cuda")
            gx = torch.empty(B, k, device="cuda")
            ws = torch.zeros(max(lib.dfm_linear_backward_workspace_bytes(B, n, k) // 4, 1), device="cuda")
            t = timed(lambda: lib.dfm_linear_backward(dz.data_ptr(), B, n, x.data_ptr(), k, w.data_ptr(), gx.data_ptr(), None, None, 2,
                                                      ws.data_ptr(), st))
            row.append(f"N={n}: {t:6.1f}")
        print(f"B={B} ({(B + 63) // 64 * 10} workgroups)  " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
