"""GPU parity: whole-model logits/gradients (DeepFM here; xDeepFM / AttentionDeepFM in their
own files) and the row-sparse training step, against golden vectors and the oracle."""
import copy

import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import (assert_close, cfg_of, fields_of, group, load, load_params, npy,
                           random_fields_batch, schema_from_fields, to_device_batch)
from tools_shared import criteo_fields

pytestmark = pytest.mark.gpu


def _config(c):
    from deepfm_amd.config import ExperimentConfig
    cfg = ExperimentConfig()
    cfg.feature.fm_embed_dim = c["fm_dim"]
    cfg.dnn.hidden_units = list(c["hidden_units"])
    cfg.dnn.dropout = 0.0
    if c["kind"] == "xdeepfm":
        cfg.cin.layer_sizes, cfg.cin.split_half = list(c["cin_sizes"]), c["cin_split"]
    if c["kind"] == "attention_deepfm":
        cfg.attention.num_heads, cfg.attention.attention_dim = c["heads"], c["A"]
        cfg.attention.num_layers, cfg.attention.use_residual = c["layers"], c["residual"]
    return cfg


def check_model_case(case):
    from deepfm_amd.models import create_model
    g = load(case)
    c = cfg_of(g)
    model = create_model(c["kind"], schema_from_fields(fields_of(g)), _config(c))
    want_p = group(g, "param/")
    assert sorted(model.state_dict().keys()) == sorted(want_p.keys())      # drop-in state_dict
    load_params(model, want_p)
    model.embedding.strict_indices = True
    batch = to_device_batch(group(g, "batch/"))
    model.eval()
    with torch.no_grad():
        logits = model(batch)
        assert logits.shape == g["logits_eval"].shape
        assert_close(npy(logits), g["logits_eval"], what="eval logits")
        p = npy(model.predict(batch))
        assert (p >= 0).all() and (p <= 1).all()                           # tests/test_models.py:36-41
    model.train()
    logits = model(batch)
    assert_close(npy(logits), g["logits_train"], what="train logits")
    loss = torch.nn.functional.binary_cross_entropy_with_logits(
        logits.squeeze(-1), torch.from_numpy(g["labels"]).cuda())
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    want_g = group(g, "grad/")
    for k, prm in model.named_parameters():
        assert prm.grad is not None, k
        pre_bn_bias = k.startswith("dnn.mlp.") and k.endswith(".bias") and int(k.split(".")[2]) % 4 == 0
        # identically-zero gradients (reference value is rounding noise): a Linear bias in front of
        # train-mode BatchNorm, the softmax-invariant W_k.bias, and the last LayerNorm bias of the
        # attention stack (a per-feature constant into Linear -> BatchNorm)
        floor = 1e-6 if (pre_bn_bias or k.endswith("layer_norm.bias")) else (2e-6 if k.endswith("W_k.bias") else 0.0)
        assert_close(npy(prm.grad), want_g[k], what=k, floor=floor)
    assert float(model.get_l2_reg_loss()) > 0                              # tests/test_models.py:43-46


@pytest.mark.parametrize("case", ["model_deepfm", "model_deepfm_movielens"])
def test_deepfm_vs_golden(case):
    check_model_case(case)


def test_registry():
    """tests/test_models.py:98-112"""
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import MODEL_REGISTRY, create_model
    assert "deepfm" in MODEL_REGISTRY
    with pytest.raises(ValueError):
        create_model("nope", schema_from_fields(criteo_fields(10, 16)), ExperimentConfig())


# ------------------------------------------------------------------ row-sparse mode

def test_rowgrad_equals_dense_reference_gradients():
    from deepfm_amd.models.layers.embedding import FeatureEmbedding
    g = load("emb_criteo_d16")
    fields = fields_of(g)
    emb = load_params(FeatureEmbedding(schema_from_fields(fields), 16), group(g, "param/"))
    emb.set_grad_mode("rowsparse")
    fo, fe, fl = emb(to_device_batch(group(g, "batch/")))
    assert_close(npy(fe), g["out/field_embeddings"], what="fe")
    up = {k: torch.from_numpy(g["upstream/" + k]).cuda() for k in ("first_order", "field_embeddings", "flat_embeddings")}
    ((fo * up["first_order"]).sum() + (fe * up["field_embeddings"]).sum() + (fl * up["flat_embeddings"]).sum()).backward()
    rs = emb.rowsparse
    assert rs.has_grad and rs.chunks == 1
    want = group(g, "grad/")
    num = npy(rs.num_uniq)[0]
    batch = group(g, "batch/")
    g_fe_total = g["upstream/field_embeddings"] + g["upstream/flat_embeddings"].reshape(g["upstream/field_embeddings"].shape)
    for s, f in enumerate(fields[:26]):
        n = int(num[s])
        rows = npy(rs.uniq_rows)[0, s, :n]
        assert (np.diff(rows) > 0).all() and (rows > 0).all()
        dense2 = np.zeros_like(want[f"second_order_embeddings.{f['name']}.weight"])
        dense1 = np.zeros_like(want[f"first_order_embeddings.{f['name']}.weight"])
        dense2[rows] = npy(rs.row_g2)[0, s, :n]
        dense1[rows, 0] = npy(rs.row_g1)[0, s, :n]
        assert_close(dense2, want[f"second_order_embeddings.{f['name']}.weight"], what=f["name"])
        assert_close(dense1, want[f"first_order_embeddings.{f['name']}.weight"], what=f["name"])
        # bit-exact against the oracle's ordered reduction (same fp32 addition order)
        u, r2, r1 = O.rowsparse_from_batch(batch[f["name"]], g_fe_total[:, s, :], g["upstream/first_order"][:, 0])
        assert np.array_equal(u, rows) and np.array_equal(r2, npy(rs.row_g2)[0, s, :n])
        assert np.array_equal(r1, npy(rs.row_g1)[0, s, :n])
    for k, p in emb.named_parameters():       # DENSE-field Linears still get autograd gradients
        if ".I" in k:
            assert_close(npy(p.grad), want[k], what=k)


def _small_deepfm(V=300, seed=0):
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    fields = criteo_fields(V, 16)
    cfg = ExperimentConfig()
    cfg.dnn.hidden_units, cfg.dnn.dropout = [64, 32], 0.0
    torch.manual_seed(seed)
    model = create_model("deepfm", schema_from_fields(fields), cfg).cuda().train()
    model.embedding.set_grad_mode("rowsparse")
    return fields, cfg, model


def _pool(fields, n, B, rng):
    ids = np.stack([np.stack([random_fields_batch([f], B, rng, 0.05)[f["name"]] for f in fields[:26]]) for _ in range(n)])
    dense = rng.random((n, 13, B)).astype(np.float32)
    labels = (rng.random((n, B)) < 0.25).astype(np.float32)
    return ids, dense, labels


def _oracle_state(model):
    params = {k: npy(v).copy() for k, v in model.state_dict().items() if not k.endswith("num_batches_tracked")}
    state = {}
    for k, v in params.items():
        if "running_" not in k:
            state["m/" + k], state["v/" + k] = np.zeros_like(v), np.zeros_like(v)
    return params, state


@pytest.mark.parametrize("B", [512, 6000])      # 6000 -> two sorted chunks merged by ownership
def test_rowsparse_train_steps_vs_oracle(B):
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    fields, cfg, model = _small_deepfm()
    hp = dict(lr=1e-3, l2=1e-5, max_grad_norm=1.0)
    params, state = _oracle_state(model)
    opt = RowSparseAdam(model, lr=hp["lr"], l2=hp["l2"], max_grad_norm=hp["max_grad_norm"])
    step = RowSparseTrainStep(model, opt, B, use_graph=False)
    rng = np.random.default_rng(3)
    ids, dense, labels = _pool(fields, 3, B, rng)
    ocfg = dict(fm_dim=16, hidden_units=cfg.dnn.hidden_units)
    for i in range(3):
        step.load_batch(torch.from_numpy(ids[i]).cuda(), torch.from_numpy(dense[i]).cuda(), torch.from_numpy(labels[i]).cuda())
        step.run()
        batch = {f["name"]: ids[i, j] for j, f in enumerate(fields[:26])}
        batch.update({f["name"]: dense[i, j] for j, f in enumerate(fields[26:])})
        oloss = O.deepfm_train_step_rowsparse(fields, params, state, batch, labels[i], ocfg, hp, i + 1, exact_order=(B <= 512))
        assert abs(float(step.loss) - float(oloss)) < 2e-5 + 1e-4 * abs(float(oloss)), (i, float(step.loss), float(oloss))
    got = {k: npy(v) for k, v in model.state_dict().items()}
    for k, want in params.items():
        if "running_" in k:
            continue        # BatchNorm running statistics are not modelled by the oracle step
        if k.startswith("dnn.mlp.") and k.endswith(".bias") and int(k.split(".")[2]) % 4 == 0:
            continue        # zero-gradient parameter: Adam amplifies rounding noise to +-lr
        # Adam divides by sqrt(v): elements with |g| ~ eps are ill-conditioned -> absolute floor of lr/10
        assert_close(got[k], want, rtol=1e-4, atol_scale=0.0, floor=1e-4, what=k)


def test_graph_replay_is_bitwise_equal_to_eager_and_deterministic():
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    B = 1024
    results = []
    rng = np.random.default_rng(9)
    fields = criteo_fields(300, 16)
    ids, dense, labels = _pool(fields, 4, B, rng)
    for use_graph in (False, True, True):
        _, _, model = _small_deepfm(seed=4)
        opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
        step = RowSparseTrainStep(model, opt, B, use_graph=use_graph)
        start = copy.deepcopy(model.state_dict())
        step.load_batch(torch.from_numpy(ids[0]).cuda(), torch.from_numpy(dense[0]).cuda(), torch.from_numpy(labels[0]).cuda())
        step.capture()
        # capture() is side-effect free: model, optimizer state and the loaded batch are untouched
        for k, v in model.state_dict().items():
            assert torch.equal(v, start[k]), f"capture() changed {k}"
        assert int(opt.step_count) == 0 and not opt.flat_m.any() and not opt.flat_v.any() and not opt.flat_grad.any()
        assert all(not t.any() for t in opt.exp_avg + opt.exp_avg_sq)
        assert torch.equal(step.in_ids, torch.from_numpy(ids[0]).cuda())      # the loaded batch is still in the inbox
        for i in range(4):
            step.load_batch(torch.from_numpy(ids[i]).cuda(), torch.from_numpy(dense[i]).cuda(), torch.from_numpy(labels[i]).cuda())
            step.run()
        torch.cuda.synchronize()
        results.append({k: npy(v).copy() for k, v in model.state_dict().items()})
    for k in results[0]:
        assert np.array_equal(results[1][k], results[2][k]), f"graph replay not deterministic: {k}"
        assert np.array_equal(results[0][k], results[1][k]), f"graph != eager: {k}"


def test_merge_of_two_simulated_ranks_vs_oracle():
    """The data-parallel merge kernel with L = 2 lists: two half batches reduced separately (as
    two ranks would), stacked rank-major like all_gather_into_tensor, merged + clipped + Adam'd
    on the GPU, against the oracle's reduction of the whole batch (grad mean over 2 ranks)."""
    from deepfm_amd.training.rowsparse import RowSparseAdam
    fields, cfg, model = _small_deepfm(V=200, seed=2)
    emb = model.embedding
    opt = RowSparseAdam(model, lr=1e-2, l2=0.0, max_grad_norm=0.5)
    Bh, F, D = 700, 39, 16
    rng = np.random.default_rng(21)
    ids = np.stack([random_fields_batch([f], 2 * Bh, rng, 0.05)[f["name"]] for f in fields[:26]])   # (26, 2Bh)
    dense = rng.random((13, 2 * Bh)).astype(np.float32)
    g_fe = rng.standard_normal((2 * Bh, F, D)).astype(np.float32)
    g_fo = rng.standard_normal((2 * Bh, 1)).astype(np.float32)
    w_before = {k: npy(v).copy() for k, v in emb.state_dict().items()}
    parts = []
    for r in range(2):
        sl = slice(r * Bh, (r + 1) * Bh)
        inputs = [torch.from_numpy(np.ascontiguousarray(ids[s, sl])).cuda() for s in range(26)] + \
                 [torch.from_numpy(np.ascontiguousarray(dense[j, sl])).cuda() for j in range(13)]
        emb._ensure_plan(inputs[0].device)
        rs = emb.build_rowplan(inputs, Bh)
        emb.backward_rowsparse(inputs, torch.from_numpy(g_fo[sl]).cuda(), torch.from_numpy(g_fe[sl]).cuda(), {})
        parts.append([t.clone() for t in (rs.uniq_rows, rs.num_uniq, rs.row_g2, rs.row_g1)])
    gathered = [torch.cat([parts[0][i], parts[1][i]], dim=0) for i in range(4)]
    opt.world = 2
    opt._cur = tuple(gathered) + (2,)
    opt.flat_grad.zero_()
    opt.apply()
    torch.cuda.synchronize()
    # oracle: whole batch, mean over the two ranks
    rows, sq = {}, 0.0
    for s, f in enumerate(fields[:26]):
        u, a2, a1 = O.rowsparse_reduce_fast(ids[s], g_fe[:, s, :], g_fo[:, 0])
        a2, a1 = (0.5 * a2).astype(np.float32), (0.5 * a1).astype(np.float32)
        rows[f["name"]] = (u, a2, a1)
        sq += float((a2.astype(np.float64) ** 2).sum() + (a1.astype(np.float64) ** 2).sum())
    assert abs(float(opt.sq_norm) - sq) < 1e-4 * sq
    coef = O.clip_coef(sq, 0.5)
    assert abs(float(opt.clip_coef) - float(coef)) < 1e-5
    got = {k: npy(v) for k, v in emb.state_dict().items()}
    for name, (u, a2, a1) in rows.items():
        for key, g in ((f"second_order_embeddings.{name}.weight", a2), (f"first_order_embeddings.{name}.weight", a1[:, None])):
            w = w_before[key].copy()
            wu, m, v = w[u], np.zeros_like(w[u]), np.zeros_like(w[u])
            O.adam_update(wu, m, v, g * coef, 1, 1e-2)
            w[u] = wu
            assert_close(got[key], w, rtol=1e-4, atol_scale=0.0, floor=2e-5, what=key)
            untouched = np.setdiff1d(np.arange(w.shape[0]), u)
            assert np.array_equal(got[key][untouched], w_before[key][untouched])   # only touched rows move


def test_packed_row_records_are_bitwise_equivalent():
    """FeatureEmbedding.pack_tables_(): same values, same state_dict, bit-identical forward and
    bit-identical training steps as the separate-tensor layout (only addresses change)."""
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    B = 768
    rng = np.random.default_rng(17)
    fields = criteo_fields(300, 16)
    ids, dense, labels = _pool(fields, 3, B, rng)
    finals = []
    for packed in (False, True):
        _, _, model = _small_deepfm(seed=6)
        before = {k: npy(v).copy() for k, v in model.state_dict().items()}
        if packed:
            model.embedding.pack_tables_()
            after = {k: npy(v) for k, v in model.state_dict().items()}
            assert sorted(before) == sorted(after)
            for k in before:
                assert np.array_equal(before[k], after[k]), k
            w = model.embedding.second_order_embeddings["C1"].weight
            assert w.shape == (300, 16) and w.stride(0) == 64
        model.embedding.set_grad_mode("rowsparse")
        opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
        step = RowSparseTrainStep(model, opt, B, use_graph=False)
        for i in range(3):
            step.load_batch(torch.from_numpy(ids[i]).cuda(), torch.from_numpy(dense[i]).cuda(), torch.from_numpy(labels[i]).cuda())
            step.run()
        torch.cuda.synchronize()
        finals.append({k: npy(v).copy() for k, v in model.state_dict().items()})
        with torch.no_grad():   # dense-gradient mode also works on packed tables
            pass
    for k in finals[0]:
        assert np.array_equal(finals[0][k], finals[1][k]), f"packed layout changed {k}"


def test_packed_tables_dense_grad_mode_vs_golden():
    from deepfm_amd.models.layers.embedding import FeatureEmbedding
    g = load("emb_criteo_d16")
    emb = load_params(FeatureEmbedding(schema_from_fields(fields_of(g)), 16), group(g, "param/"))
    emb.pack_tables_()
    emb.strict_indices = True
    fo, fe, fl = emb(to_device_batch(group(g, "batch/")))
    assert np.array_equal(npy(fe)[:, :26], g["out/field_embeddings"][:, :26])
    assert_close(npy(fo), g["out/first_order"], what="fo")
    up = {k: torch.from_numpy(g["upstream/" + k]).cuda() for k in ("first_order", "field_embeddings", "flat_embeddings")}
    ((fo * up["first_order"]).sum() + (fe * up["field_embeddings"]).sum() + (fl * up["flat_embeddings"]).sum()).backward()
    want = group(g, "grad/")
    for k, p in emb.named_parameters():
        assert_close(npy(p.grad), want[k], what=k)


def test_match_kernel_is_bitwise_equivalent_for_many_lists():
    """With >= 3 row lists (data-parallel ranks x chunks) the merge resolves list memberships through the
    LDS match kernel instead of global binary searches: same owners, same sums, bit-identical tables.
    Small vocabulary: nearly every row is shared by several of the 5 lists."""
    from deepfm_amd.training.rowsparse import RowSparseAdam
    L, Bh, F, D = 5, 600, 39, 16
    rng = np.random.default_rng(33)
    fields = criteo_fields(200, 16)
    ids = np.stack([random_fields_batch([f], L * Bh, rng, 0.05)[f["name"]] for f in fields[:26]])
    dense = rng.random((13, L * Bh)).astype(np.float32)
    g_fe = rng.standard_normal((L * Bh, F, D)).astype(np.float32)
    g_fo = rng.standard_normal((L * Bh, 1)).astype(np.float32)
    results = []
    for use_match in (True, False):
        _, _, model = _small_deepfm(V=200, seed=2)
        emb = model.embedding
        opt = RowSparseAdam(model, lr=1e-2, l2=1e-4, max_grad_norm=0.5)
        parts = []
        for r in range(L):
            sl = slice(r * Bh, (r + 1) * Bh)
            inputs = [torch.from_numpy(np.ascontiguousarray(ids[s, sl])).cuda() for s in range(26)] + \
                     [torch.from_numpy(np.ascontiguousarray(dense[j, sl])).cuda() for j in range(13)]
            emb._ensure_plan(inputs[0].device)
            rs = emb.build_rowplan(inputs, Bh)
            emb.backward_rowsparse(inputs, torch.from_numpy(g_fo[sl]).cuda(), torch.from_numpy(g_fe[sl]).cuda(), {})
            parts.append([t.clone() for t in (rs.uniq_rows, rs.num_uniq, rs.row_g2, rs.row_g1)])
        gathered = [torch.cat([p[i] for p in parts], dim=0) for i in range(4)]
        opt.world = L
        opt._cur = tuple(gathered) + (L,)
        opt.flat_grad.zero_()
        if not use_match:
            # size the scratch buffers, then drop the match workspace: the merge falls back to searches
            import deepfm_amd._lib as _l
            real = _l.load().dfm_step_match_bytes
            _l.load().dfm_step_match_bytes = lambda *a: 0
            try:
                opt.apply()
            finally:
                _l.load().dfm_step_match_bytes = real
            assert opt._match is None
        else:
            opt.apply()
            assert opt._match is not None
        torch.cuda.synchronize()
        results.append({k: npy(v).copy() for k, v in emb.state_dict().items()})
        results[-1]["__sq"] = npy(opt.sq_norm).copy()
    for k in results[0]:
        assert np.array_equal(results[0][k], results[1][k]), k


@pytest.mark.parametrize("fused", [False, True])
def test_run_from_record_equals_load_then_run(fused):
    """run_from(record): the gather reads the batch from its record and refreshes the static inputs
    itself — bit-identical to load_packed(record) + run(), including odd tails (B not a multiple of 16)."""
    from deepfm_amd.training.fused_step import FusedDeepFMStep
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    B = 1000
    rng = np.random.default_rng(12)
    fields = criteo_fields(300, 16)
    ids, dense, labels = _pool(fields, 3, B, rng)
    results = []
    for staged in (False, True):
        _, _, model = _small_deepfm(seed=6)
        opt = RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
        step = (FusedDeepFMStep if fused else RowSparseTrainStep)(model, opt, B, use_graph=False)
        recs = step.pack_batches(torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda(), torch.from_numpy(labels).cuda())
        losses = []
        for i in range(3):
            if staged:
                step.run_from(recs[i])
            else:
                step.load_packed(recs[i])
                step.run()
            losses.append(float(step.loss))
        if staged:      # the static buffers hold the last batch
            assert torch.equal(step.ids, torch.from_numpy(ids[2]).cuda())
            assert torch.equal(step.dense, torch.from_numpy(dense[2]).cuda())
            assert torch.equal(step.labels, torch.from_numpy(labels[2]).cuda())
        results.append(({k: npy(v).copy() for k, v in model.state_dict().items()}, losses))
    assert results[0][1] == results[1][1]
    for k in results[0][0]:
        assert np.array_equal(results[0][0][k], results[1][0][k]), k


@pytest.mark.parametrize("hot", [False, True])
@pytest.mark.parametrize("vocab", [3, 64, 65, 128, 129, 200, 1000, 20000, (1 << 20) - 2, (1 << 20) - 1, 3_000_000])
def test_rowplan_key_widths_vs_oracle(vocab, hot):
    """dfm_rowplan_build sorts 32-bit keys (id << 12 | pos) when every vocabulary is below 2^20 - 1 and
    64-bit keys otherwise: both against the oracle's ordered reduction, with duplicates, padding ids, the
    largest id and an odd tail (two chunks).  Vocabularies of <= 128 ids take one pass of the ballot-counting
    radix sort, larger ones the bucket ranking, and with `hot` (40 % of the batch on three ids, the largest id among
    them) the buckets overflow and the multi-pass radix sort runs (2 ... 4 passes, both key widths).  Bit-exact for every row with at most 64 contributions (summed
    in sample order); longer runs (the 50-id field: ~80 per row) go through the workgroup-wide fixed tree of
    rowgrad_body and are held to rounding."""
    from deepfm_amd import _lib
    import ctypes as C
    lib = _lib.load()
    B, D, S, F = 5000, 16, 2, 3
    rng = np.random.default_rng(vocab % 977)
    ids = [rng.integers(1, vocab, size=B).astype(np.int64), rng.integers(1, min(vocab, 50), size=B).astype(np.int64)]
    if hot:
        pick = np.array([vocab - 1, max(1, vocab // 2), min(5, vocab - 1)], dtype=np.int64)
        ids[0] = np.where(rng.random(B) < 0.4, pick[rng.integers(0, 3, size=B)], ids[0])
    ids[0][:7] = [vocab - 1, vocab - 1, 0, 1, 0, vocab - 1, 1]
    d_ids = [torch.from_numpy(a).cuda() for a in ids]
    ch = _lib.ROWPLAN_CHUNK
    chunks = (B + ch - 1) // ch
    i32 = dict(dtype=torch.int32, device="cuda")
    sorted_pos, uniq = torch.empty(chunks, S, ch, **i32), torch.empty(chunks, S, ch, **i32)
    seg, num = torch.empty(chunks, S, ch + 1, **i32), torch.zeros(chunks, S, **i32)
    err = torch.zeros(1, **i32)
    ptrs = (C.c_void_p * S)(*[t.data_ptr() for t in d_ids])
    voc = (C.c_int32 * S)(vocab, min(vocab, 50))
    _lib.check(lib.dfm_rowplan_build(ptrs, voc, S, B, sorted_pos.data_ptr(), uniq.data_ptr(), seg.data_ptr(),
                                     num.data_ptr(), err.data_ptr(), None, 0, _lib.stream_handle()))
    g_fe = rng.standard_normal((B, F, D)).astype(np.float32)
    g_fo = rng.standard_normal((B, 1)).astype(np.float32)
    row_g2 = torch.zeros(chunks, S, ch, D, device="cuda")
    row_g1 = torch.zeros(chunks, S, ch, device="cuda")
    fmap = (C.c_int32 * S)(0, 2)
    _lib.check(lib.dfm_rowgrad_build(fmap, S, F, D, B, torch.from_numpy(g_fo).cuda().data_ptr(),
                                     torch.from_numpy(g_fe).cuda().data_ptr(), sorted_pos.data_ptr(), seg.data_ptr(),
                                     num.data_ptr(), row_g2.data_ptr(), row_g1.data_ptr(), _lib.stream_handle()))
    assert int(err) == 0
    for s, f in enumerate((0, 2)):
        for c in range(chunks):
            sl = slice(c * ch, min((c + 1) * ch, B))
            u, r2, r1 = O.rowsparse_from_batch(ids[s][sl], g_fe[sl, f, :], g_fo[sl, 0])
            n = int(num[c, s])
            assert n == len(u)
            assert np.array_equal(npy(uniq)[c, s, :n], u)
            runs = np.bincount(ids[s][sl], minlength=int(u.max()) + 1)[u]
            short = runs <= 64
            got2, got1 = npy(row_g2)[c, s, :n], npy(row_g1)[c, s, :n]
            assert np.array_equal(got2[short], r2[short]) and np.array_equal(got1[short], r1[short])
            # (a run of n standard normals sums to ~sqrt(n); two fp32 summation orders differ by ~n * 6e-8 of that)
            np.testing.assert_allclose(got2[~short], r2[~short], rtol=1e-4, atol=5e-4)
            np.testing.assert_allclose(got1[~short], r1[~short], rtol=1e-4, atol=5e-4)
            if s == 1 and c == 0 and vocab >= 50:
                assert (~short).any()
