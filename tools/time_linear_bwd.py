#!/usr/bin/env python3
"""dfm_linear_backward at the tower's first-layer shapes: d weight (parts=1), d input (parts=2), both (3).
usage: python tools/time_linear_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    B = 4096
    for n, k in ((256, 624), (256, 2496), (128, 256), (64, 128)):
        dz = torch.randn(B, n, device="cuda")
        x = torch.randn(B, k, device="cuda")
        w = torch.randn(n, k, device="cuda")
        gx = torch.empty(B, k, device="cuda")
        ws = torch.zeros(max(lib.dfm_linear_backward_workspace_bytes(B, n, k) // 4, 1), device="cuda")
        flops = 2.0 * B * n * k
        for parts in (1, 2, 3):
            def run():
                _lib.check(lib.dfm_linear_backward(dz.data_ptr(), B, n, x.data_ptr(), k, w.data_ptr(), gx.data_ptr(), None,
                                                   None, parts, ws.data_ptr(), _lib.stream_handle()))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / 20 * 1e3
            f = flops * (2 if parts == 3 else 1)
            print(f"out {n:4d} in {k:5d} parts {parts}: {us:8.1f} us  {f / us / 1e6:6.1f} TFLOP/s  "
                  f"(splits {lib.dfm_linear_backward_splits(B, n, k)})")


if __name__ == "__main__":
    main()
