#!/usr/bin/env python3
"""Times the REFERENCE's training step on this container's CPU cores ("reference-here").

Build container only (needs /root/reference; never runs on the GPU box).  The reference's layer
classes are imported by file path as in tools/make_golden.py and wired per deepfm.py:20-42 /
xdeepfm.py:20-48 / attention_deepfm.py:25-66 (`_RefComposite`); the step is the body of
Trainer._train_epoch (trainer.py:212-240): forward, BCEWithLogitsLoss (:59), get_l2_reg_loss
(base.py:78-83, lambda 1e-5), backward, clip_grad_norm_(1.0) (:232-235), Adam(lr 1e-3) (:67-70, 237).
Synthetic Criteo shape (26 SPARSE x V, 13 DENSE, batch 4096, uniform ids in [1, V)), pre-collated
batches, 1 warm-up + N timed steps, split timers as in the table of BASELINE.md §2.

usage: python tools/time_reference.py [deepfm|xdeepfm|attention_deepfm] [vocab=1000000] [steps=3]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as G  # noqa: E402  (reference layer classes + schema helpers)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "deepfm"
    vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dim = 32 if kind == "attention_deepfm" else 16
    B = 4096
    torch.set_num_threads(os.cpu_count() or 8)
    fields = G.criteo_fields(vocab, dim)
    schema = G.to_schema(fields)
    torch.manual_seed(0)
    t0 = time.perf_counter()
    model = G._RefComposite(kind, schema, dim, [256, 128, 64], cin_sizes=[128, 128, 128])
    # the reference default dropout (config.py) is 0.1; _RefComposite builds the parity variant (0.0)
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.1
    model.train()
    n_params = sum(p.numel() for p in model.parameters())
    print(f"{kind}: {n_params / 1e6:.1f} M parameters, built in {time.perf_counter() - t0:.1f} s, "
          f"{torch.get_num_threads()} threads", flush=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = nn.BCEWithLogitsLoss()
    rng = np.random.default_rng(1)

    def batch():
        b = {f"C{j + 1}": torch.from_numpy(rng.integers(1, vocab, B, dtype=np.int64)) for j in range(26)}
        b.update({f"I{j + 1}": torch.from_numpy(rng.random(B, dtype=np.float32)) for j in range(13)})
        return b, torch.from_numpy((rng.random((B, 1)) < 0.25).astype(np.float32))

    names = ["emb fwd", "components fwd", "L2 fwd", "backward", "clip", "Adam"]
    tot = np.zeros(len(names))
    for it in range(steps + 1):
        x, y = batch()
        ts = [time.perf_counter()]
        opt.zero_grad()
        fo, fe, fl = model.embedding(x)
        ts.append(time.perf_counter())
        if kind == "deepfm":
            logits = fo + model.fm(fe) + model.output_linear(model.dnn(fl))
        elif kind == "xdeepfm":
            logits = fo + model.cin_linear(model.cin(fe)) + model.dnn_linear(model.dnn(fl))
        else:
            a = model.attention(fe)
            logits = fo + model.fm(fe) + model.output_linear(model.dnn(torch.cat([a.reshape(B, -1), fl], dim=1)))
        loss = crit(logits, y)
        ts.append(time.perf_counter())
        loss = loss + 1e-5 * sum(p.pow(2).sum() for p in model.embedding.parameters())
        ts.append(time.perf_counter())
        loss.backward()
        ts.append(time.perf_counter())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        ts.append(time.perf_counter())
        opt.step()
        ts.append(time.perf_counter())
        if it > 0:
            tot += np.diff(ts)
    tot /= steps
    print(" | ".join(f"{n} {t:.3f}" for n, t in zip(names, tot)))
    print(f"total {tot.sum():.3f} s/step -> {B / tot.sum():.0f} samples/s (V = {vocab}, batch {B})")


if __name__ == "__main__":
    main()
