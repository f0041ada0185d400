#!/usr/bin/env python3
"""Why the DNN tower stays on the exact-fp32 matrix pipe: the bf16 x 3 split that the CIN
uses (A = Ah + Al, B = Bh + Bl, A B ~ Ah Bh + Ah Bl + Al Bh, fp32 accumulate; 2^-16 per product) emulated in numpy
on the DeepFM tower of BASELINE.json's config 2 (624 -> 256 -> 128 -> 64 -> 1, train-mode BatchNorm, B = 4096),
against the same tower in float64.  Every GEMM of the forward and the backward is replaced; the BatchNorm / ReLU /
loss arithmetic stays fp32.  Prints the share of elements outside the parity bar of tests/helpers.assert_close
(1e-4 relative + 1e-5 of the tensor's scale) for the logits and every gradient, next to plain fp32.
CPU only:  python tools/emulate_tower_bf16x3.py"""
import numpy as np


def bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000          # round to nearest even
    return r.view(np.float32)


def split(x):
    hi = bf16(x)
    return hi, bf16(x - hi)


def mm(a, b, mode):
    if mode == "f64":
        return a.astype(np.float64) @ b.astype(np.float64)
    if mode == "f32":
        return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)
    ah, al = split(a.astype(np.float32))
    bh, bl = split(b.astype(np.float32))
    f = np.float64                                             # products exact, accumulation ~fp32 or better: a LOWER bound
    return (ah.astype(f) @ bh.astype(f) + ah.astype(f) @ bl.astype(f) + al.astype(f) @ bh.astype(f)).astype(np.float32)


def tower(x, Ws, gs, bs, w_head, y, mode):
    bwd_mode = mode
    if mode == "x3bwd":            # exact-fp32 forward (the ReLU masks are fp32's), bf16 x 3 in the backward GEMMs only
        mode, bwd_mode = "f32", "x3"
    dt = np.float64 if mode == "f64" else np.float32
    h, cache = x.astype(dt), []
    for W, g, b in zip(Ws, gs, bs):
        z = mm(h, W.T, mode).astype(dt)
        mu, var = z.mean(0), z.var(0)
        rstd = 1.0 / np.sqrt(var + 1e-5)
        zh = (z - mu) * rstd
        a = np.maximum(zh * g + b, 0)
        cache.append((h, zh, rstd, a))
        h = a.astype(dt)
    logit = mm(h, w_head.T, mode).astype(dt)[:, 0]
    d = ((1 / (1 + np.exp(-logit)) - y) / len(y)).astype(dt)[:, None]
    grads = {"head": mm(d.T, h, bwd_mode)}
    gh = mm(d, w_head, bwd_mode).astype(dt)
    for i in reversed(range(len(Ws))):
        hin, zh, rstd, a = cache[i]
        ga = gh * (a > 0)
        grads[f"gamma{i}"], grads[f"beta{i}"] = (ga * zh).sum(0), ga.sum(0)
        gz = ga * gs[i]
        gz = rstd * (gz - gz.mean(0) - zh * (gz * zh).mean(0))
        grads[f"W{i}"] = mm(gz.T, hin, bwd_mode)
        gh = mm(gz, Ws[i], bwd_mode).astype(dt)
    grads["x"] = gh
    return logit, grads


def outside(got, ref):
    ref = np.asarray(ref, np.float64)
    err = np.abs(np.asarray(got, np.float64) - ref)
    return float((err > 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max()).mean()), float(err.max() / np.abs(ref).max())


def main(kink_free=False):
    rng = np.random.default_rng(0)
    B, dims = 4096, [624, 256, 128, 64]
    x = ((rng.random((B, dims[0])) - 0.5) * 0.5).astype(np.float32)           # trained-scale embeddings (+-0.25)
    Ws = [((rng.random((dims[i + 1], dims[i])) - 0.5) * 2 / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    gs = [np.ones(d, np.float32) for d in dims[1:]]
    # kink_free: BatchNorm biases +6 / -6 (every third unit dead): no pre-activation within rounding of the ReLU kink,
    # so what remains is the arithmetic error alone (tests/test_gpu_fullsize.py's "-kinkfree" construction)
    bs = [np.where(np.arange(d) % 3 == 2, -6.0, 6.0).astype(np.float32) if kink_free else np.zeros(d, np.float32)
          for d in dims[1:]]
    w_head = ((rng.random((1, 64)) - 0.5) * 0.25).astype(np.float32)
    y = (rng.random(B) < 0.25).astype(np.float32)
    ref_l, ref_g = tower(x, Ws, gs, bs, w_head, y, "f64")
    print(f"--- {'kink-free tower (BatchNorm biases +-6)' if kink_free else 'natural tower (BatchNorm biases 0)'}")
    print(f"{'tensor':10s} {'fp32: share outside / max err':>32s} {'bf16x3: share outside / max err':>34s} {'bf16x3 backward only':>26s}")
    res = {m: tower(x, Ws, gs, bs, w_head, y, m) for m in ("f32", "x3", "x3bwd")}
    rows = [("logits", lambda r: r[0])] + [(k, (lambda k: lambda r: r[1][k])(k)) for k in ref_g]
    for name, get in rows:
        ref = ref_l if name == "logits" else ref_g[name]
        a, b, c = outside(get(res["f32"]), ref), outside(get(res["x3"]), ref), outside(get(res["x3bwd"]), ref)
        print(f"{name:10s} {a[0]:>20.2e} / {a[1]:.1e} {b[0]:>22.2e} / {b[1]:.1e} {c[0]:>16.2e} / {c[1]:.1e}")


if __name__ == "__main__":
    main(False)
    main(True)
