#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE implementation.

Run in the build container only (needs /root/reference; it never travels to
the GPU box):   python tools/make_golden.py

How the reference is imported: ``deepfm.data.schema`` imports normally from
/root/reference; the five layer files under deepfm/models/layers/ are loaded
by file path (importlib) because ``deepfm/models/__init__.py`` imports
``deepfm.config`` -> the third-party ``dacite`` package, which is not installed
here.  No stand-in for dacite is created.  Consequently the three model
classes (deepfm.py / xdeepfm.py / attention_deepfm.py, which need
deepfm.config at import) are NOT imported: the ``model_*`` cases compose the
reference's own layer classes with torch.nn heads using the reference's
attribute names and formulas (deepfm.py:20-42, xdeepfm.py:20-48,
attention_deepfm.py:25-66), so logits and state_dict keys are the reference's.

Every case stores inputs, parameters (keyed like the reference state_dict),
outputs and autograd gradients of the reference.  Fixtures are data only.
"""

from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

sys.path.insert(0, REF)
from deepfm.data.schema import DatasetSchema, FeatureType, FieldSchema  # noqa: E402


def _load_layer(stem: str):
    spec = importlib.util.spec_from_file_location(
        f"_ref_layer_{stem}", f"{REF}/deepfm/models/layers/{stem}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


RefEmbedding = _load_layer("embedding").FeatureEmbedding
RefFM = _load_layer("fm").FMInteraction
RefCIN = _load_layer("cin").CIN
RefAttention = _load_layer("attention").MultiHeadSelfAttention
RefDNN = _load_layer("dnn").DNN

TYPE = {"sparse": FeatureType.SPARSE, "dense": FeatureType.DENSE, "sequence": FeatureType.SEQUENCE}


# ----------------------------------------------------------------------------
# schemas (plain dict descriptions shared with the tests through the npz)
# ----------------------------------------------------------------------------

def criteo_fields(vocab: int, dim: int, n_sparse: int = 26, n_dense: int = 13):
    fs = [dict(name=f"C{i+1}", type="sparse", vocab=vocab, dim=dim, max_len=1, combiner="mean")
          for i in range(n_sparse)]
    fs += [dict(name=f"I{i+1}", type="dense", vocab=0, dim=dim, max_len=1, combiner="mean")
           for i in range(n_dense)]
    return fs


def movielens_fields(combiner: str = "mean"):
    """MovieLens-shaped schema (movielens.py:346-418) with synthetic vocabularies."""
    sp = [("user_id", 944, 16), ("movie_id", 1683, 16), ("gender", 3, 4), ("age", 8, 4),
          ("occupation", 22, 8), ("zip_prefix", 400, 8)]
    fs = [dict(name=n, type="sparse", vocab=v, dim=d, max_len=1, combiner="mean") for n, v, d in sp]
    fs.append(dict(name="genres", type="sequence", vocab=20, dim=8, max_len=6, combiner=combiner))
    for n, v in (("release_year_bucket", 16), ("movie_age_at_rating", 8), ("num_genres", 8)):
        fs.append(dict(name=n, type="sparse", vocab=v, dim=4, max_len=1, combiner="mean"))
    for n in ("dow_sin", "dow_cos", "hour_sin", "hour_cos"):
        fs.append(dict(name=n, type="dense", vocab=0, dim=4, max_len=1, combiner="mean"))
    for n in ("user_rating_count", "item_rating_count"):
        fs.append(dict(name=n, type="dense", vocab=0, dim=8, max_len=1, combiner="mean"))
    return fs


def to_schema(fields) -> DatasetSchema:
    d = {}
    for f in fields:
        d[f["name"]] = FieldSchema(name=f["name"], feature_type=TYPE[f["type"]],
                                   vocabulary_size=f["vocab"], embedding_dim=f["dim"],
                                   max_length=f["max_len"], combiner=f["combiner"])
    return DatasetSchema(fields=d, label_field="label")


def make_batch(fields, B: int, rng: np.random.Generator):
    """Edge cases on purpose: id 0 (padding), duplicates, the maximum id, all-pad bags."""
    batch = {}
    for f in fields:
        if f["type"] == "sparse":
            x = rng.integers(1, f["vocab"], size=B, dtype=np.int64)
            x[rng.random(B) < 0.1] = 0
            x[0] = f["vocab"] - 1
            if B > 4:
                x[3] = x[1]
                x[4] = x[1]                # >= 3 duplicates of one row
        elif f["type"] == "sequence":
            L = f["max_len"]
            x = rng.integers(1, f["vocab"], size=(B, L), dtype=np.int64)
            lens = rng.integers(0, L + 1, size=B)
            for b in range(B):
                x[b, lens[b]:] = 0
            x[0, :] = 0                    # all-pad bag
            if B > 2:
                x[2, :] = x[2, 0]          # repeated id inside one bag
        else:
            x = rng.random(B).astype(np.float32) * 2 - 1
        batch[f["name"]] = x
    return batch


def randomize_(module: nn.Module, rng: np.random.Generator, scale: float = 0.5, keep_pad=True):
    """Overwrite every parameter with seeded numpy values (trained-like scale);
    embedding row 0 stays the zero padding row."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            v = (rng.random(p.shape, dtype=np.float32) * 2 - 1) * scale
            p.copy_(torch.from_numpy(v))
        if keep_pad:
            for m in module.modules():
                if isinstance(m, (nn.Embedding, nn.EmbeddingBag)):
                    m.weight[0].zero_()
        for m in module.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.running_mean.copy_(torch.from_numpy((rng.random(m.num_features, dtype=np.float32) - 0.5) * 0.2))
                m.running_var.copy_(torch.from_numpy(rng.random(m.num_features, dtype=np.float32) * 0.5 + 0.75))


def sd_np(module: nn.Module, prefix="param/"):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module: nn.Module, prefix="grad/"):
    out = {}
    for k, p in module.named_parameters():
        out[prefix + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
    return out


def tb(batch):
    return {k: torch.from_numpy(v) for k, v in batch.items()}


def save(name: str, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays")


def fields_meta(fields):
    import json
    return np.array(json.dumps(fields))


# ----------------------------------------------------------------------------
# cases
# ----------------------------------------------------------------------------

def case_embedding(name, fields, fm_dim, B, seed):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    emb = RefEmbedding(to_schema(fields), fm_embed_dim=fm_dim)
    randomize_(emb, rng)
    batch = make_batch(fields, B, rng)
    fo, fe, fl = emb(tb(batch))
    g_fo = rng.standard_normal(fo.shape).astype(np.float32)
    g_fe = rng.standard_normal(fe.shape).astype(np.float32)
    g_fl = rng.standard_normal(fl.shape).astype(np.float32)
    loss = (fo * torch.from_numpy(g_fo)).sum() + (fe * torch.from_numpy(g_fe)).sum() + \
        (fl * torch.from_numpy(g_fl)).sum()
    loss.backward()
    arrays = dict(fields=fields_meta(fields), fm_dim=np.int64(fm_dim))
    arrays.update({"batch/" + k: v for k, v in batch.items()})
    arrays.update(sd_np(emb))
    arrays.update({"out/first_order": fo.detach().numpy(), "out/field_embeddings": fe.detach().numpy(),
                   "out/flat_embeddings": fl.detach().numpy(),
                   "upstream/first_order": g_fo, "upstream/field_embeddings": g_fe,
                   "upstream/flat_embeddings": g_fl})
    arrays.update(grads_np(emb))
    save(name, **arrays)


def case_fm(seed=11):
    rng = np.random.default_rng(seed)
    e = rng.standard_normal((64, 39, 16)).astype(np.float32)
    t = torch.from_numpy(e).requires_grad_()
    out = RefFM()(t)
    g = rng.standard_normal(out.shape).astype(np.float32)
    (out * torch.from_numpy(g)).sum().backward()
    known = np.array([[[1, 2], [3, 4], [5, 6]]], dtype=np.float32)   # notes/deepfm.md:72-90
    known_out = RefFM()(torch.from_numpy(known)).numpy()
    single = rng.standard_normal((4, 1, 8)).astype(np.float32)       # tests/test_layers.py:94-98
    save("fm", x=e, out=out.detach().numpy(), upstream=g, d_x=t.grad.numpy(),
         known_x=known, known_out=known_out,
         single_x=single, single_out=RefFM()(torch.from_numpy(single)).numpy())


def hashed_weights(shape, salt: int, scale: float) -> np.ndarray:
    """Closed-form pseudo-random fp32 values (exact integer arithmetic), so large
    parameter sets need not be stored: the tests recompute them."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503)) % np.uint64(1 << 32)
    h = (h ^ (h >> np.uint64(15))) * np.uint64(2246822519) % np.uint64(1 << 32)
    h = (h ^ (h >> np.uint64(13))) % np.uint64(1 << 24)
    v = (h.astype(np.float64) / float(1 << 24) - 0.5) * 2.0 * scale
    return v.astype(np.float32).reshape(shape)


def case_cin(name, F, D, layer_sizes, split_half, B, seed, store_params=True, w_scale=None):
    rng = np.random.default_rng(seed)
    cin = RefCIN(F, D, list(layer_sizes), split_half)
    if store_params:
        randomize_(cin, rng, scale=0.3, keep_pad=False)
    else:
        with torch.no_grad():
            for li, conv in enumerate(cin.conv_layers):
                k = conv.weight.shape[1]
                conv.weight.copy_(torch.from_numpy(hashed_weights(conv.weight.shape, 2 * li + 1, 2.0 / np.sqrt(k))))
                conv.bias.copy_(torch.from_numpy(hashed_weights(conv.bias.shape, 2 * li + 2, 0.1)))
    x = (rng.standard_normal((B, F, D)) * 0.7).astype(np.float32)
    t = torch.from_numpy(x).requires_grad_()
    out = cin(t)
    g = rng.standard_normal(out.shape).astype(np.float32)
    (out * torch.from_numpy(g)).sum().backward()
    arrays = dict(x=x, out=out.detach().numpy(), upstream=g, d_x=t.grad.numpy(),
                  layer_sizes=np.array(layer_sizes, dtype=np.int64), split_half=np.bool_(split_half),
                  direct_sizes=np.array(cin.direct_sizes), next_sizes=np.array(cin.next_sizes),
                  output_dim=np.int64(cin.output_dim), hashed=np.bool_(not store_params))
    if store_params:
        arrays.update(sd_np(cin))
        arrays.update(grads_np(cin))
    else:
        # large weights: keep every 97th gradient element + all bias gradients
        for k, p in cin.named_parameters():
            gflat = p.grad.detach().numpy().reshape(-1)
            arrays["grad_sample/" + k] = gflat[::97].copy() if k.endswith("weight") else gflat.copy()
    save(name, **arrays)


def case_attention(name, F, D, heads, A, layers, residual, B, seed):
    rng = np.random.default_rng(seed)
    att = RefAttention(D, heads, A, layers, residual)
    randomize_(att, rng, scale=0.4, keep_pad=False)
    x = rng.standard_normal((B, F, D)).astype(np.float32)
    t = torch.from_numpy(x).requires_grad_()
    out = att(t)
    g = rng.standard_normal(out.shape).astype(np.float32)
    (out * torch.from_numpy(g)).sum().backward()
    arrays = dict(x=x, out=out.detach().numpy(), upstream=g, d_x=t.grad.numpy(),
                  num_heads=np.int64(heads), attention_dim=np.int64(A), num_layers=np.int64(layers),
                  use_residual=np.bool_(residual))
    arrays.update(sd_np(att))
    arrays.update(grads_np(att))
    save(name, **arrays)


class _RefComposite(nn.Module):
    """Reference layer classes wired per deepfm.py:20-42 / xdeepfm.py:20-48 /
    attention_deepfm.py:25-66 (the model classes themselves need dacite to import)."""

    def __init__(self, kind, schema, fm_dim, hidden, cin_sizes=None, cin_split=True,
                 heads=4, A=64, layers=1, residual=True):
        super().__init__()
        self.kind = kind
        self.embedding = RefEmbedding(schema, fm_embed_dim=fm_dim)
        total = schema.total_embedding_dim
        if kind == "deepfm":
            self.fm = RefFM()
            self.dnn = RefDNN(total, hidden, "relu", 0.0, True)
            self.output_linear = nn.Linear(self.dnn.output_dim, 1)
        elif kind == "xdeepfm":
            self.cin = RefCIN(schema.num_fields, fm_dim, cin_sizes, cin_split)
            self.dnn = RefDNN(total, hidden, "relu", 0.0, True)
            self.cin_linear = nn.Linear(self.cin.output_dim, 1)
            self.dnn_linear = nn.Linear(self.dnn.output_dim, 1)
        else:
            self.fm = RefFM()
            self.attention = RefAttention(fm_dim, heads, A, layers, residual)
            self.dnn = RefDNN(schema.num_fields * fm_dim + total, hidden, "relu", 0.0, True)
            self.output_linear = nn.Linear(self.dnn.output_dim, 1)

    def forward(self, batch):
        fo, fe, fl = self.embedding(batch)
        if self.kind == "deepfm":
            return fo + self.fm(fe) + self.output_linear(self.dnn(fl))
        if self.kind == "xdeepfm":
            return fo + self.cin_linear(self.cin(fe)) + self.dnn_linear(self.dnn(fl))
        a = self.attention(fe)
        return fo + self.fm(fe) + self.output_linear(self.dnn(torch.cat([a.reshape(a.size(0), -1), fl], dim=1)))


def case_model(name, kind, fields, fm_dim, hidden, B, seed, **kw):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = _RefComposite(kind, to_schema(fields), fm_dim, hidden, **kw)
    randomize_(model, rng, scale=0.25)
    batch = make_batch(fields, B, rng)
    labels = (rng.random(B) < 0.25).astype(np.float32)
    model.eval()
    sd = sd_np(model)                               # before train-mode BN updates running stats
    with torch.no_grad():
        logits_eval = model(tb(batch)).numpy()
    model.train()                                   # BN uses batch statistics; dropout p = 0
    logits_train = model(tb(batch))
    loss = nn.BCEWithLogitsLoss()(logits_train.squeeze(-1), torch.from_numpy(labels))
    loss.backward()
    import json
    cfg = dict(kind=kind, fm_dim=fm_dim, hidden_units=hidden, **kw)
    arrays = dict(fields=fields_meta(fields), cfg=np.array(json.dumps(cfg)), labels=labels,
                  logits_eval=logits_eval, logits_train=logits_train.detach().numpy(),
                  loss=np.float32(loss.item()))
    arrays.update({"batch/" + k: v for k, v in batch.items()})
    arrays.update(sd)
    arrays.update(grads_np(model))
    save(name, **arrays)


def covering_batch(fields, B: int, rng: np.random.Generator):
    """Every row 1..V-1 of every SPARSE table occurs at least once (then the reference's dense
    Adam + full-table L2 and a row-wise lazy step coincide: no untouched row exists, and the
    padding row 0 has gradient 0 and stays 0), plus duplicates and padding ids."""
    batch = {}
    for f in fields:
        if f["type"] == "sparse":
            V = f["vocab"]
            assert B >= V - 1
            x = np.concatenate([np.arange(1, V, dtype=np.int64),
                                rng.integers(0, V, size=B - (V - 1), dtype=np.int64)])
            batch[f["name"]] = rng.permutation(x)
        else:
            batch[f["name"]] = rng.random(B).astype(np.float32) * 2 - 1
    return batch


def case_train_steps(name, fields, hidden, B, steps, seed, lr, l2, clip, scale=0.25, kind="deepfm", fm_dim=16, **kw):
    """The body of the reference's ``Trainer._train_epoch`` (trainer.py:212-240) run for ``steps``
    batches on the reference's own layer classes: BCEWithLogitsLoss (trainer.py:59) + the L2 term of
    ``BaseCTRModel.get_l2_reg_loss`` (base.py:78-83: lambda * sum ||p||_2^2 over embedding.parameters()),
    ``optimizer.zero_grad`` / ``backward`` / ``clip_grad_norm_(model.parameters(), clip)``
    (trainer.py:228-235) / ``torch.optim.Adam(model.parameters(), lr)`` (trainer.py:67-70, 237)."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = _RefComposite(kind, to_schema(fields), fm_dim, hidden, **kw)
    randomize_(model, rng, scale=scale)
    model.train()
    criterion = nn.BCEWithLogitsLoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    import json
    cfg = dict(kind=kind, fm_dim=fm_dim, hidden_units=hidden, **kw)
    arrays = dict(fields=fields_meta(fields), cfg=np.array(json.dumps(cfg)), steps=np.int64(steps),
                  lr=np.float64(lr), l2=np.float64(l2), clip=np.float64(clip))
    arrays.update(sd_np(model, "init/"))
    for t in range(steps):
        batch = covering_batch(fields, B, rng)
        labels = (rng.random(B) < 0.3).astype(np.float32)
        logits = model(tb(batch)).squeeze(1)
        bce = criterion(logits, torch.from_numpy(labels))
        l2_loss = torch.tensor(0.0)
        for p in model.embedding.parameters():
            l2_loss = l2_loss + p.norm(2).pow(2)
        l2_term = l2 * l2_loss
        loss = bce + l2_term
        optimizer.zero_grad()
        loss.backward()
        arrays.update(grads_np(model, f"step{t}/grad/"))        # before clipping: d(bce + l2)/dp
        total_norm = nn.utils.clip_grad_norm_(model.parameters(), clip)
        optimizer.step()
        arrays.update({f"step{t}/batch/" + k: v for k, v in batch.items()})
        arrays[f"step{t}/labels"] = labels
        arrays[f"step{t}/logits"] = logits.detach().numpy().copy()
        arrays[f"step{t}/bce"] = np.float32(bce.item())
        arrays[f"step{t}/l2_term"] = np.float32(l2_term.item())
        arrays[f"step{t}/loss"] = np.float32(loss.item())
        arrays[f"step{t}/grad_norm"] = np.float32(float(total_norm))
        arrays.update(sd_np(model, f"step{t}/param/"))
    names = [k for k, _ in model.named_parameters()]
    for i, k in enumerate(names):                               # torch Adam state is keyed by position
        st = optimizer.state_dict()["state"][i]
        arrays["adam_m/" + k] = st["exp_avg"].numpy().copy()
        arrays["adam_v/" + k] = st["exp_avg_sq"].numpy().copy()
    save(name, **arrays)


def main():
    torch.set_num_threads(4)
    # train-step tail: BCE + L2 + clip + Adam as the reference's trainer runs them
    case_train_steps("train_steps_deepfm", criteo_fields(12, 16), [64, 32], 64, 3, 501,
                     lr=1e-3, l2=1e-5, clip=1.0)                       # reference defaults (config.py:30,64,70)
    case_train_steps("train_steps_deepfm_l2clip", criteo_fields(9, 16), [64, 32], 48, 3, 502,
                     lr=1e-2, l2=1e-2, clip=0.25)                      # L2 and clipping both bite
    # the same trainer body over the other two compositions (xdeepfm.py:36-48: CIN with split-half, three
    # layers; attention_deepfm.py:48-66: embed_dim 32, attention_dim 64, 4 heads, residual LayerNorm) —
    # what bench.py's extra_configs time as FusedXDeepFMStep / FusedAttentionDeepFMStep
    case_train_steps("train_steps_xdeepfm", criteo_fields(12, 16), [64, 32], 64, 3, 503,
                     lr=1e-3, l2=1e-5, clip=1.0, kind="xdeepfm", cin_sizes=[16, 16, 8], cin_split=True)
    case_train_steps("train_steps_xdeepfm_l2clip", criteo_fields(9, 16), [64, 32], 48, 3, 504,
                     lr=1e-2, l2=1e-2, clip=0.25, kind="xdeepfm", cin_sizes=[24, 12], cin_split=True)
    case_train_steps("train_steps_attention_deepfm", criteo_fields(10, 32), [32, 32], 64, 3, 505,
                     lr=1e-3, l2=1e-5, clip=1.0, kind="attention_deepfm", fm_dim=32,
                     heads=4, A=64, layers=1, residual=True)
    case_train_steps("train_steps_attention_deepfm_l2clip", criteo_fields(9, 32), [32, 32], 48, 3, 506,
                     lr=1e-2, l2=1e-2, clip=0.25, kind="attention_deepfm", fm_dim=32,
                     heads=4, A=64, layers=2, residual=True)
    if os.environ.get("GOLDEN_ONLY") == "train":
        return
    # FeatureEmbedding
    case_embedding("emb_movielens_mean", movielens_fields("mean"), 16, 64, 101)
    case_embedding("emb_movielens_sum", movielens_fields("sum"), 16, 32, 102)
    case_embedding("emb_movielens_max", movielens_fields("max"), 16, 32, 103)
    case_embedding("emb_criteo_d16", criteo_fields(200, 16), 16, 64, 104)
    case_embedding("emb_criteo_d32", criteo_fields(100, 32), 32, 32, 105)
    case_embedding("emb_layers_test_schema", [   # tests/test_layers.py:13-19
        dict(name="u", type="sparse", vocab=100, dim=8, max_len=1, combiner="mean"),
        dict(name="i", type="sparse", vocab=200, dim=16, max_len=1, combiner="mean"),
        dict(name="g", type="sparse", vocab=3, dim=4, max_len=1, combiner="mean")], 16, 8, 106)
    case_fm()
    # CIN
    case_cin("cin_small_split", 7, 8, [16, 16, 8], True, 16, 201)
    case_cin("cin_small_nosplit", 5, 4, [12, 10], False, 8, 202)
    case_cin("cin_single_layer", 6, 8, [64], True, 8, 203)
    case_cin("cin_odd_split", 4, 6, [5, 7, 3], True, 8, 204)
    case_cin("cin_criteo_full", 39, 16, [128, 128, 128], True, 8, 205, store_params=False)
    # attention
    case_attention("attn_cfg4", 39, 32, 4, 64, 1, True, 16, 301)
    case_attention("attn_two_layers", 39, 32, 4, 64, 2, True, 8, 302)
    case_attention("attn_no_residual", 16, 16, 2, 32, 1, False, 8, 303)
    case_attention("attn_odd", 7, 12, 3, 24, 2, True, 5, 304)
    # whole models (reference layers composed with the reference formulas)
    case_model("model_deepfm", "deepfm", criteo_fields(50, 16), 16, [32, 16], 32, 401)
    case_model("model_xdeepfm", "xdeepfm", criteo_fields(50, 16), 16, [32, 16], 16, 402,
               cin_sizes=[16, 16], cin_split=True)
    case_model("model_attention_deepfm", "attention_deepfm", criteo_fields(40, 32), 32, [32, 16], 16, 403,
               heads=4, A=64, layers=1, residual=True)
    case_model("model_deepfm_movielens", "deepfm", movielens_fields("mean"), 16, [32, 16], 32, 404)


if __name__ == "__main__":
    main()
