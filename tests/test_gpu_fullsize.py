"""GPU: BASELINE.json's three single-GPU configurations at their TRUE size — 26 tables x 1 000 000 rows, 13
dense fields, batch 4096, packed row records, the fused step replayed as a HIP graph from packed batch
records (exactly what ``bench.py`` times: the headline and both ``extra_configs``) — against the oracle
(SURVEY.md §8d "on-box large-shape self-check"):

  * config 2: DeepFM, D = 16, ``FusedDeepFMStep``              (deepfm.py:30-42)
  * config 3: xDeepFM, D = 16, CIN [128,128,128] split-half, ``FusedXDeepFMStep``   (xdeepfm.py:36-48, cin.py:66-105)
  * config 4: AttentionDeepFM, D = 32, 4 heads, attention_dim 64, residual LayerNorm,
              ``FusedAttentionDeepFMStep``                      (attention_deepfm.py:48-66, attention.py:91-120)

each with embedding.py:76-126, dnn.py:45-59 and the trainer body trainer.py:219-237 around it.

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

V, B, S, ND = 1_000_000, 4096, 26, 13

# kind -> (embedding dim, table scale, share of the batch allowed to sit on a ReLU kink, oracle cfg).
# Fresh xavier tables at V = 10^6 are ~2e-3: products of two or three of them (CIN) and scores of a softmax
# (attention) vanish in fp32 noise, and the interaction layers would be checked on zeros.  Configurations 3
# and 4 therefore scale the tables to a trained-like +-0.25 (what the goldens use) before the first step.
# "-kinkfree" variants: every BatchNorm bias of the tower (and every CIN bias) is set to +6 / -6 (every third unit
# dead), so that no pre-activation sits within rounding of the ReLU kink: the gradient is continuous there, NO
# sample may disagree with the oracle and every moment of every parameter is held to the bar with no outliers.
KINDS = {
    "deepfm": dict(D=16, table_scale=1.0, kink_share=0.003, ocfg={}),
    "deepfm-kinkfree": dict(D=16, table_scale=1.0, kink_share=0.0, ocfg={}),
    "xdeepfm-kinkfree": dict(D=16, table_scale=100.0, kink_share=0.0,
                             ocfg=dict(cin_layer_sizes=[128, 128, 128], cin_split_half=True)),
    "attention_deepfm-kinkfree": dict(D=32, table_scale=100.0, kink_share=0.0,
                                      ocfg=dict(num_heads=4, num_layers=1, use_residual=True)),
    # CIN: 3 x 128 x 16 ReLUs per sample on split-bf16 products (relative error ~1e-5 against the fp32
    # oracle) next to the tower's 448: a few dozen of the batch's 25 M pre-activations land on the other side
    "xdeepfm": dict(D=16, table_scale=100.0, kink_share=0.03,
                    ocfg=dict(cin_layer_sizes=[128, 128, 128], cin_split_half=True)),
    "attention_deepfm": dict(D=32, table_scale=100.0, kink_share=0.003,
                             ocfg=dict(num_heads=4, num_layers=1, use_residual=True)),
}


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


@pytest.mark.parametrize("case", list(KINDS))
def test_config_two_graph_steps_vs_oracle(case):
    kind, kink_free = case.split("-")[0], case.endswith("-kinkfree")
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    from deepfm_amd.training.fused_step import fused_step_class
    from deepfm_amd.training.rowsparse import RowSparseAdam
    D, table_scale, kink_share = KINDS[case]["D"], KINDS[case]["table_scale"], KINDS[case]["kink_share"]
    RS = (3 * D + 4 + 31) // 32 * 32               # floats per packed record (embedding.py::pack_tables_)
    fields = criteo_fields(V, D)
    cfg = ExperimentConfig()                       # reference defaults: tower [256,128,64], lr 1e-3, l2 1e-5, clip 1
    cfg.dnn.dropout = 0.0                          # parity runs: the dropout RNG streams differ by design
    cfg.feature.fm_embed_dim = D
    if kind == "xdeepfm":
        cfg.cin.layer_sizes, cfg.cin.split_half = [128, 128, 128], True       # BASELINE.json config 3
    if kind == "attention_deepfm":                                            # BASELINE.json config 4
        cfg.attention.num_heads, cfg.attention.attention_dim = 4, 64
        cfg.attention.num_layers, cfg.attention.use_residual = 1, True
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = create_model(kind, schema_from_fields(fields), cfg)
    model.train()
    if kink_free:
        with torch.no_grad():
            for m in model.dnn.mlp:
                if isinstance(m, torch.nn.BatchNorm1d):
                    c = torch.arange(m.bias.numel(), device="cuda")
                    m.bias.copy_(torch.where(c % 3 == 2, -6.0, 6.0))
            if kind == "xdeepfm":
                for conv in model.cin.conv_layers:
                    c = torch.arange(conv.bias.numel(), device="cuda")
                    conv.bias.copy_(torch.where(c % 3 == 2, -6.0, 6.0))
                    conv.weight.mul_(0.25)
                model.cin_linear.weight.mul_(0.02)     # pooled activations are ~16 x 6 per live channel: keep the logit O(1)
    model.embedding.pack_tables_()
    if table_scale != 1.0:
        with torch.no_grad():
            for n in model.embedding.packed:
                model.embedding.packed[n]["buffer"][:, :D + 1].mul_(table_scale)
    model.embedding.set_grad_mode("rowsparse")
    hp = dict(lr=cfg.training.lr, l2=cfg.feature.embedding_l2_reg, max_grad_norm=cfg.training.gradient_clip_norm)
    opt = RowSparseAdam(model, lr=hp["lr"], l2=hp["l2"], max_grad_norm=hp["max_grad_norm"])
    cls = fused_step_class(model)                  # what bench.py picks for this model
    assert cls is not None and cls.__name__.lower() == "fused" + kind.replace("_", "") + "step"
    step = cls(model, opt, B, use_graph=True)
    ids, dense, labels = _pool(2, 11)
    records = step.pack_batches(ids, dense, labels)
    names = [f["name"] for f in fields[:S]]
    before = {n: model.embedding.packed[n]["buffer"].clone() for n in names}          # 26 x 256 MB (512 MB at D = 32)
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
        rows = npy(before[n][torch.from_numpy(u).cuda()])                             # (n_u, RS) records
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
    ocfg = dict(fm_dim=D, hidden_units=cfg.dnn.hidden_units, **KINDS[case]["ocfg"])
    dense_h, labels_h = npy(dense), npy(labels)

    lr, (b1, b2), eps = hp["lr"], (0.9, 0.999), 1e-8
    k2s = {n: f"embedding.second_order_embeddings.{n}.weight" for n in names}
    k1s = {n: f"embedding.first_order_embeddings.{n}.weight" for n in names}
    sel = {n: torch.from_numpy(uniq[n]).cuda() for n in names}
    n_checked = 0
    for t in range(2):
        prev = {n: npy(model.embedding.packed[n]["buffer"][sel[n]]) for n in names}
        step.run_from(records[t])
        torch.cuda.synchronize()
        batch = {n: small_ids[n][t] for n in names}
        batch.update({f["name"]: dense_h[t, i] for i, f in enumerate(fields[S:])})
        info = {}
        oloss = O.train_step_rowsparse(kind, small_fields, params, state, batch, labels_h[t], ocfg, hp, t + 1, info=info)
        assert_close(npy(step.logits), info["logits"].reshape(-1), what=f"logits step {t}")
        assert abs(float(step.loss) - float(oloss)) < 1e-4 * float(oloss)
        assert abs(float(opt.sq_norm) - info["sq_norm"]) < 1e-4 * info["sq_norm"]
        coef = float(info["coef"])
        assert abs(float(opt.clip_coef) - coef) < 1e-4 * coef + 1e-6
        assert int(model.embedding._err.item()) == 0
        bc1, bc2 = 1 - b1 ** (t + 1), 1 - b2 ** (t + 1)
        kinked = set()
        for j, n in enumerate(names):
            got = npy(model.embedding.packed[n]["buffer"][sel[n]])                       # (n_u, RS) records
            u_t = info["rows"][n][0]                                                    # compact ids this step touched
            hit = np.zeros(len(uniq[n]), bool)
            hit[u_t - 1] = True
            # (1) Adam moments are LINEAR in the gradient: well conditioned, held to the oracle directly.
            #     They pin gather -> forward -> backward -> row reduction -> lazy L2 -> clip at full size.
            m2, v2 = got[:, D + 4:2 * D + 4], got[:, 2 * D + 4:3 * D + 4]
            m1, v1 = got[:, D + 1], got[:, D + 2]
            want_m2 = state["m/" + k2s[n]][1:]
            # ReLU kinks: a hidden pre-activation within rounding of 0 may take the other side than in the
            # oracle's summation order; that changes ONE sample's d(embedding) by a few percent.  Rows fed
            # by such a sample are collected (kinked, at most 0.3 % of the batch over all fields) and left
            # out; every other row is held to the oracle.
            bad_rows = np.flatnonzero((np.abs(m2 - want_m2) > 1e-4 * np.abs(want_m2) + 2e-5 * np.abs(want_m2).max()).any(axis=1))
            for r in bad_rows:
                pos = np.flatnonzero(ids_h[t, j] == uniq[n][r])
                assert len(pos) >= 1, f"{n} step {t}: exp_avg differs for a row outside the batch"
                assert np.abs(m2[r] - want_m2[r]).max() < 0.05 * np.abs(want_m2).max(), (n, t, int(uniq[n][r]))
                kinked.update(pos.tolist())
            keep = np.ones(len(uniq[n]), bool)
            keep[bad_rows] = False
            assert_close(m2[keep], want_m2[keep], rtol=1e-4, atol_scale=2e-5, what=f"{n} exp_avg step {t}")
            assert_close(v2[keep], state["v/" + k2s[n]][1:][keep], rtol=2e-4, atol_scale=2e-5, what=f"{n} exp_avg_sq step {t}")
            assert_close(m1, state["m/" + k1s[n]][1:, 0], rtol=1e-4, atol_scale=2e-5, what=f"{n} exp_avg(1st) step {t}")
            assert_close(v1, state["v/" + k1s[n]][1:, 0], rtol=2e-4, atol_scale=2e-5, what=f"{n} exp_avg_sq(1st) step {t}")
            # (2) the weight update IS ill conditioned wherever |g| ~ eps (at B = 4096 most row-gradient
            #     elements are 1e-9..1e-6): it is checked as Adam's formula (trainer.py:67-70, 237) applied to
            #     the kernel's own moments, and against the oracle's weights on the well-conditioned elements.
            w_prev = np.concatenate([prev[n][:, :D], prev[n][:, D:D + 1]], axis=1).astype(np.float64)
            m = np.concatenate([m2, m1[:, None]], axis=1).astype(np.float64)
            v = np.concatenate([v2, v1[:, None]], axis=1).astype(np.float64)
            w_formula = w_prev - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)
            w_got = np.concatenate([got[:, :D], got[:, D:D + 1]], axis=1)
            assert np.abs(w_got[hit] - w_formula[hit]).max() < 1e-3 * lr, f"{n}: weights are not Adam(m, v) at step {t}"
            assert np.array_equal(got[~hit], prev[n][~hit]), f"{n}: a record outside step {t}'s batch changed"
            want_w = np.concatenate([params[k2s[n]][1:], params[k1s[n]][1:]], axis=1)
            r2, r1 = info["rows"][n][1], info["rows"][n][2]
            well = np.zeros(want_w.shape, bool)
            well[u_t - 1] = np.abs(np.concatenate([r2, r1[:, None]], axis=1)) * coef > 1e-6
            well[~keep] = False
            assert_close(np.where(well, w_got, 0), np.where(well, want_w, 0), rtol=1e-4, atol_scale=0.0, floor=0.02 * lr,
                         what=f"{n} rows step {t}")
            n_checked += int(hit.sum())
            # teacher forcing: the next step of the oracle starts from the kernel's state, so that step 1 is
            # checked on its own and not through the ill-conditioned part of step 0
            params[k2s[n]][1:], params[k1s[n]][1:, 0] = got[:, :D], got[:, D]
            state["m/" + k2s[n]][1:], state["v/" + k2s[n]][1:] = m2, v2
            state["m/" + k1s[n]][1:, 0], state["v/" + k1s[n]][1:, 0] = m1, v1
        assert len(kinked) <= kink_share * B, \
            f"step {t}: {len(kinked)} samples disagree with the oracle (ReLU-kink allowance: {int(kink_share * B)})"
        print(f"[fullsize {case}] step {t}: {len(kinked)} kinked samples, loss {float(step.loss):.6f} "
              f"(oracle {float(oloss):.6f}), |g| {float(opt.sq_norm) ** 0.5:.5f}, clip {coef:.5f}")
        # ---- dense parameters: moments against the oracle, then teacher forcing.  A kinked sample moves the
        # gradient row of the flipped unit by ~1/B of its magnitude: bounded outliers (<= 1 % of a tensor,
        # each within 1 % of the tensor's gradient scale), everything else at the normal bar.
        osd = opt.state_dict()["state"]
        got_p = {k: npy(v) for k, v in model.state_dict().items()}
        for k in list(params):
            if "embeddings.C" in k or "running_" in k:
                continue
            # identically-zero gradients (noise on both sides): a Linear bias in front of BatchNorm, W_k.bias
            # (softmax shift invariance), the attention's last LayerNorm bias (a constant into Linear -> BN)
            pre_bn_bias = (k.startswith("dnn.mlp.") and k.endswith(".bias") and int(k.split(".")[2]) % 4 == 0) \
                or k.endswith("W_k.bias") or k == "attention.layers.0.layer_norm.bias"
            # kink-free tower: a live unit's ReLU is the identity on the whole batch, so the gradient of its BatchNorm
            # bias is W^T sum_b(d z of the next BatchNorm) = 0 — BatchNorm's backward sums to zero over the batch
            if kink_free and k.startswith("dnn.mlp.") and k.endswith(".bias") and int(k.split(".")[2]) % 4 == 1 \
                    and int(k.split(".")[2]) // 4 < len(cfg.dnn.hidden_units) - 1:
                pre_bn_bias = True
            gm, gv = npy(osd[k]["exp_avg"]), npy(osd[k]["exp_avg_sq"])
            if not pre_bn_bias:
                gscale = float(np.abs(info["grads"][k]).max()) * coef
                # a kinked sample flips ONE unit: its row of the Linear weight, its BatchNorm weight and bias
                allow = (2.0 * len(kinked) / state["m/" + k].shape[0] + 0.01) if kinked else 0.0
                em = np.abs(gm - state["m/" + k])
                out_m = em > 1e-4 * np.abs(state["m/" + k]) + 1e-3 * gscale       # batch sums of 4096 cancelling terms
                if k.startswith("attention.") and kinked:
                    # every element of these small tensors is a sum over ALL samples with cancellation (~sqrt(B) of one
                    # sample's term): each kinked sample shifts every element by up to a few % of 1 / sqrt(B) of the scale
                    assert em.max() <= (1e-3 + 1e-3 * len(kinked)) * gscale, (k, t, float(em.max() / gscale), len(kinked))
                else:
                    assert out_m.mean() <= allow and em.max() <= 1e-2 * gscale, \
                        (k, t, float(out_m.mean()), float(em.max() / gscale))
                ev = np.abs(gv - state["v/" + k])
                out_v = ev > 2e-4 * np.abs(state["v/" + k]) + 2e-6 * gscale ** 2
                att_kinked = k.startswith("attention.") and bool(kinked)
                assert out_v.mean() <= allow or \
                    (att_kinked and ev.max() <= (2e-3 + 2e-3 * len(kinked)) * gscale ** 2), (k, t, float(out_v.mean()))
                well = (np.abs(info["grads"][k]) * coef > 1e-6) & ~out_m
                # 2 % of one Adam step; a kinked sample may push a few small-gradient elements further
                # (the step is lr * g / |g|-like): at most 0.1 % of a tensor, none beyond 20 % of a step
                err = np.abs(np.where(well, got_p[k].astype(np.float64) - params[k], 0))
                bound = 1e-4 * np.abs(params[k]) + 0.02 * lr
                # (AttentionDeepFM's 2496-wide first layer: a kinked sample can flip the SIGN of a gradient element
                #  just above the 1e-6 floor — one step each way, 2 lr apart; the kink-free variant allows none of this)
                assert ((err > bound).mean() <= allow / 10 or att_kinked) and \
                    err.max() <= (2.1 if (kinked and kind == "attention_deepfm") else 0.2) * lr, \
                    (k, t, float((err > bound).mean()), float(err.max()))
            params[k][...], state["m/" + k][...], state["v/" + k][...] = got_p[k], gm, gv
        for k in params:
            if "running_" in k:
                params[k][...] = got_p[k]
    assert n_checked > 2 * S * B * 0.95

    # ---- untouched rows: bit-unchanged records (weights AND moments), all 26 x 10^6 of them
    for n in names:
        buf = model.embedding.packed[n]["buffer"]
        changed = (buf != before[n]).any(dim=1).nonzero().view(-1)
        assert np.array_equal(npy(changed), uniq[n]), f"{n}: rows changed that the batches did not touch (or vice versa)"
