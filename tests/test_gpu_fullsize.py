"""GPU: the HEADLINE configuration at its true size — 26 tables x 1 000 000 rows, 13 dense fields, D = 16,
batch 4096, packed 256-B row records, ``FusedDeepFMStep`` replayed as a HIP graph from packed batch
records (exactly what ``bench.py`` times) — against the oracle (SURVEY.md §8d "on-box large-shape
self-check").  Replaces, at full size: embedding.py:76-126, fm.py:18-23, dnn.py:45-59, deepfm.py:30-42,
trainer.py:219-237.

The oracle cannot hold 1.77 GB tables per step in seconds, and does not need to: a row-sparse step only
reads and writes the rows the batch touches.  The touched rows of both steps are copied out of the GPU
tables into compact host tables (ids renumbered 1..n per field, 0 stays the padding id), the oracle runs
the same two steps on those, and every touched row, every dense parameter, the logits and the losses
are compared; all UNtouched rows (26 x ~992 000 records incl. their Adam moments) must be bit-unchanged.
"""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, npy, schema_from_fields
from tools_shared import criteo_fields

pytestmark = pytest.mark.gpu

V, B, D, S, ND = 1_000_000, 4096, 16, 26, 13


def _pool(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    ids = torch.randint(1, V, (n, S, B), generator=g, device="cuda", dtype=torch.int64)
    ids.masked_fill_(torch.rand((n, S, B), generator=g, device="cuda") < 0.01, 0)      # 1 % padding ids
    ids[:, :, 0] = V - 1                                                               # the largest id
    ids[:, :, 5:9] = ids[:, :, 4:5]                                                    # duplicates inside a batch
    ids[1, :, 100:200] = ids[0, :, 100:200]                                            # rows hit by both steps
    dense = torch.rand((n, ND, B), generator=g, device="cuda")
    labels = (torch.rand((n, B), generator=g, device="cuda") < 0.25).float()
    return ids, dense, labels


def test_headline_config_two_graph_steps_vs_oracle():
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    from deepfm_amd.training.fused_step import FusedDeepFMStep
    from deepfm_amd.training.rowsparse import RowSparseAdam
    fields = criteo_fields(V, D)
    cfg = ExperimentConfig()                       # reference defaults: tower [256,128,64], lr 1e-3, l2 1e-5, clip 1
    cfg.dnn.dropout = 0.0                          # parity runs: the dropout RNG streams differ by design
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = create_model("deepfm", schema_from_fields(fields), cfg)
    model.train()
    model.embedding.pack_tables_()
    model.embedding.set_grad_mode("rowsparse")
    hp = dict(lr=cfg.training.lr, l2=cfg.feature.embedding_l2_reg, max_grad_norm=cfg.training.gradient_clip_norm)
    opt = RowSparseAdam(model, lr=hp["lr"], l2=hp["l2"], max_grad_norm=hp["max_grad_norm"])
    assert FusedDeepFMStep.eligible(model)
    step = FusedDeepFMStep(model, opt, B, use_graph=True)
    ids, dense, labels = _pool(2, 11)
    records = step.pack_batches(ids, dense, labels)
    names = [f["name"] for f in fields[:S]]
    before = {n: model.embedding.packed[n]["buffer"].clone() for n in names}          # 26 x 256 MB
    dense_before = {k: npy(v).copy() for k, v in model.state_dict().items()
                    if "embeddings.C" not in k and not k.endswith("num_batches_tracked")}
    step.load_packed(records[0])
    step.capture()
    for n in names:                                # capture() is side-effect free, also at this size
        assert torch.equal(model.embedding.packed[n]["buffer"], before[n]), n

    # ---- compact host problem: only the rows either step touches
    ids_h = npy(ids)
    params = dict(dense_before)
    state, uniq, small_ids = {}, {}, {}
    for j, n in enumerate(names):
        u = np.unique(ids_h[:, j, :])
        u = u[u != 0]
        uniq[n] = u
        rows = npy(before[n][torch.from_numpy(u).cuda()])                             # (n_u, 64) records
        assert not rows[:, D + 1:D + 3].any() and not rows[:, D + 4:3 * D + 4].any()   # Adam moments start at 0
        k2, k1 = f"embedding.second_order_embeddings.{n}.weight", f"embedding.first_order_embeddings.{n}.weight"
        params[k2] = np.concatenate([np.zeros((1, D), np.float32), rows[:, :D]])
        params[k1] = np.concatenate([np.zeros((1, 1), np.float32), rows[:, D:D + 1]])
        assert not npy(before[n][0]).any()                                             # padding row
        small_ids[n] = np.searchsorted(u, ids_h[:, j, :]) + 1
        small_ids[n][ids_h[:, j, :] == 0] = 0
    for k, v in params.items():
        if "running_" not in k:
            state["m/" + k], state["v/" + k] = np.zeros_like(v), np.zeros_like(v)
    small_fields = [dict(f, vocab=len(uniq[f["name"]]) + 1) if f["type"] == "sparse" else f for f in fields]
    ocfg = dict(fm_dim=D, hidden_units=cfg.dnn.hidden_units)
    dense_h, labels_h = npy(dense), npy(labels)

    ill = {}                                       # |clipped gradient| within 100x of Adam's eps: ill-conditioned
    for t in range(2):
        step.run_from(records[t])
        torch.cuda.synchronize()
        batch = {n: small_ids[n][t] for n in names}
        batch.update({f["name"]: dense_h[t, i] for i, f in enumerate(fields[S:])})
        info = {}
        oloss = O.deepfm_train_step_rowsparse(small_fields, params, state, batch, labels_h[t], ocfg, hp, t + 1, info=info)
        assert_close(npy(step.logits), info["logits"].reshape(-1), what=f"logits step {t}")
        assert abs(float(step.loss) - float(oloss)) < 1e-4 * float(oloss)
        assert abs(float(opt.sq_norm) - info["sq_norm"]) < 1e-4 * info["sq_norm"]
        assert abs(float(opt.clip_coef) - float(info["coef"])) < 1e-5
        for n, (u, r2, r1) in info["rows"].items():
            m = ill.setdefault(n, np.zeros((len(uniq[n]) + 1, D + 1), bool))
            m[u, :D] |= np.abs(r2) * float(info["coef"]) < 1e-6
            m[u, D] |= np.abs(r1) * float(info["coef"]) < 1e-6
        assert int(model.embedding._err.item()) == 0

    # ---- touched rows: weights (1e-4 + 2 % of one Adam step on well-conditioned elements) and moments
    lr = hp["lr"]
    n_checked = 0
    for j, n in enumerate(names):
        buf = model.embedding.packed[n]["buffer"]
        u = uniq[n]
        got = npy(buf[torch.from_numpy(u).cuda()])
        k2, k1 = f"embedding.second_order_embeddings.{n}.weight", f"embedding.first_order_embeddings.{n}.weight"
        want_w = np.concatenate([params[k2][1:], params[k1][1:]], axis=1)               # (n_u, 17)
        ok = ~ill[n][1:]
        assert ok.mean() > 0.9, (n, ok.mean())
        assert_close(np.where(ok, got[:, :D + 1], 0), np.where(ok, want_w, 0), rtol=1e-4, atol_scale=0.0,
                     floor=0.02 * lr, what=f"{n} rows")
        moved = np.abs(got[:, :D] - npy(before[n][torch.from_numpy(u).cuda()])[:, :D]).max(axis=1)
        assert (moved > 0.5 * lr).all(), "a touched row did not take its Adam step"
        assert_close(got[:, D + 4:2 * D + 4], state["m/" + k2][1:], rtol=1e-3, atol_scale=1e-4, what=f"{n} exp_avg")
        assert_close(got[:, 2 * D + 4:3 * D + 4], state["v/" + k2][1:], rtol=2e-3, atol_scale=1e-4, what=f"{n} exp_avg_sq")
        # ---- untouched rows: bit-unchanged records (weights AND moments)
        changed = (buf != before[n]).any(dim=1).nonzero().view(-1)
        assert np.array_equal(npy(changed), u), f"{n}: rows changed that the batches did not touch (or vice versa)"
        n_checked += len(u)
    assert n_checked > 2 * S * B * 0.95
    # ---- dense parameters
    got = {k: npy(v) for k, v in model.state_dict().items()}
    for k, want in params.items():
        if "embeddings.C" in k or "running_" in k:
            continue
        if k.startswith("dnn.mlp.") and k.endswith(".bias") and int(k.split(".")[2]) % 4 == 0:
            continue                                # identically-zero gradient in front of BatchNorm
        g_small = np.zeros(want.shape, bool)
        if k in info["grads"]:
            g_small = np.abs(info["grads"][k]) * float(info["coef"]) < 1e-6
        assert_close(np.where(g_small, 0, got[k]), np.where(g_small, 0, want), rtol=1e-4, atol_scale=0.0,
                     floor=0.05 * lr, what=k)
