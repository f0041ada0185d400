#!/usr/bin/env python3
"""Event-timed launches of the DNN tower's GEMM kernels at the headline shapes (B = 4096, 624 -> 256 -> 128 -> 64),
exact-fp32 entry points beside the bf16 x 6 ones.  usage: python tools/time_tower_kernels.py [iters=200] [first layer in_features=624; configuration 4: 2496]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepfm_amd import _lib  # noqa: E402


def timed(fn, iters):
    for _ in range(10):
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
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first_in = int(sys.argv[2]) if len(sys.argv) > 2 else 624
    lib = _lib.load()
    st = _lib.stream_handle()
    B = 4096
    dims = [first_in, 256, 128, 64]
    g = torch.Generator(device="cuda").manual_seed(0)
    out = []
    for i in range(3 if first_in == 624 else 1):
        K, N = dims[i], dims[i + 1]
        x = torch.randn(B, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
        dz = torch.randn(B, N, device="cuda", generator=g) * 1e-3
        z = torch.empty(B, N, device="cuda")
        gx = torch.empty(B, K, device="cuda")
        ws = torch.zeros(lib.dfm_linear_bn_workspace_bytes(B, N) // 4, device="cuda")
        wl0 = torch.zeros(max(lib.dfm_linear_backward_workspace_bytes(B, N, K) // 4, 1), device="cuda")
        wl6 = torch.zeros(max(lib.dfm_linear_backward_x6_workspace_bytes(B, N, K) // 4, 1), device="cuda")

        def planes(r, c):
            return torch.zeros(lib.dfm_planes_bytes(r, c) // 2, dtype=torch.bfloat16, device="cuda")
        xf, xs, wf, wsp, dzf, dzs = planes(B, K), planes(K, B), planes(N, K), planes(K, N), planes(B, N), planes(N, B)
        jobs = (_lib.SplitJob * 3)()
        for j, (src, r, c, pf, ps) in enumerate([(x, B, K, xf, xs), (w, N, K, wf, wsp), (dz, B, N, dzf, dzs)]):
            jobs[j].src, jobs[j].rows, jobs[j].cols = src.data_ptr(), r, c
            jobs[j].planes_f, jobs[j].planes_s = pf.data_ptr(), ps.data_ptr()
        _lib.check(lib.dfm_split_planes(jobs, 3, st))
        first = i == 0
        t_f0 = timed(lambda: lib.dfm_linear_bn_forward(x.data_ptr(), K, w.data_ptr(), None, B, N, K, z.data_ptr(), ws.data_ptr(), st), iters)
        t_f6 = timed(lambda: lib.dfm_linear_bn_forward_x6(x.data_ptr() if first else None, K, None if first else xf.data_ptr(),
                                                          wf.data_ptr(), None, B, N, K, z.data_ptr(), ws.data_ptr(), st), iters)
        t_f6p = timed(lambda: lib.dfm_linear_bn_forward_x6(None, K, xf.data_ptr(), wf.data_ptr(), None, B, N, K, z.data_ptr(),
                                                           ws.data_ptr(), st), iters)
        t_b0 = timed(lambda: lib.dfm_linear_backward(dz.data_ptr(), B, N, x.data_ptr(), K, w.data_ptr(), gx.data_ptr(), None, None, 3,
                                                     wl0.data_ptr(), st), iters)
        t_b6 = timed(lambda: lib.dfm_linear_backward_x6(dzf.data_ptr(), dzs.data_ptr(), B, N, x.data_ptr() if first else None,
                                                        None if first else xs.data_ptr(), K, wsp.data_ptr(), gx.data_ptr(), None, None,
                                                        wl6.data_ptr(), st), iters)
        t_b6p = timed(lambda: lib.dfm_linear_backward_x6(dzf.data_ptr(), dzs.data_ptr(), B, N, None, xs.data_ptr(), K, wsp.data_ptr(),
                                                         gx.data_ptr(), None, None, wl6.data_ptr(), st), iters)
        out.append(f"L{i + 1} {K}->{N}: fwd f32 {t_f0:.1f} x6 {t_f6:.1f} x6-planes {t_f6p:.1f} | bwd f32 {t_b0:.1f} x6 {t_b6:.1f} x6-planes {t_b6p:.1f}")
    print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
