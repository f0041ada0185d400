#!/usr/bin/env python3
"""dfm_rowgrad_build (one gradient row per distinct id) under id distributions from uniform to one hot id:
the kernel's time must not blow up with the length of the runs (csrc/tail_bodies.h::rowgrad_body).
usage: python tools/time_rowgrad.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    S, F, D, B, V = 26, 39, 16, 4096, 1_000_000
    rng = np.random.default_rng(0)
    cases = {
        "uniform": rng.integers(1, V, size=(S, B)),
        "zipf(1.05) clipped": np.clip(rng.zipf(1.05, size=(S, B)), 1, V - 1),
        "half of the batch on one id": np.where(rng.random((S, B)) < 0.5, 7, rng.integers(1, V, size=(S, B))),
        "one id": np.full((S, B), 7),
        "64 ids": rng.integers(1, 65, size=(S, B)),
    }
    g_fe = torch.randn(B, F, D, device="cuda")
    g_fo = torch.randn(B, device="cuda")
    CH = _lib.ROWPLAN_CHUNK
    i32 = dict(dtype=torch.int32, device="cuda")
    fmap = (C.c_int32 * S)(*range(S))
    for name, ids in cases.items():
        t_ids = torch.from_numpy(ids.astype(np.int64)).cuda()
        sorted_pos, uniq = torch.empty(1, S, CH, **i32), torch.empty(1, S, CH, **i32)
        seg, num = torch.empty(1, S, CH + 1, **i32), torch.zeros(1, S, **i32)
        err = torch.zeros(1, **i32)
        idp = (C.c_void_p * S)(*[t_ids[s].data_ptr() for s in range(S)])
        vocab = (C.c_int32 * S)(*([V] * S))
        g2, g1 = torch.zeros(1, S, CH, D, device="cuda"), torch.zeros(1, S, CH, device="cuda")

        def plan():
            _lib.check(lib.dfm_rowplan_build(idp, vocab, S, B, sorted_pos.data_ptr(), uniq.data_ptr(), seg.data_ptr(),
                                             num.data_ptr(), err.data_ptr(), None, 0, _lib.stream_handle()))

        def grad():
            _lib.check(lib.dfm_rowgrad_build(fmap, S, F, D, B, g_fo.data_ptr(), g_fe.data_ptr(), sorted_pos.data_ptr(),
                                             seg.data_ptr(), num.data_ptr(), g2.data_ptr(), g1.data_ptr(),
                                             _lib.stream_handle()))
        out = []
        for fn in (plan, grad):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            out.append(sorted(a.elapsed_time(b) for a, b in ev)[10] * 1e3)
        longest = int(max(np.bincount(ids[s]).max() for s in range(S)))
        print(f"{name:30s} longest run {longest:5d}: row plan {out[0]:7.1f} us, row gradients {out[1]:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
