"""CPU oracle for the CTR feature-interaction path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy (fp32) restatement of the reference's algorithm for the
hot path — FeatureEmbedding, FMInteraction, CIN, MultiHeadSelfAttention — plus
the pieces that surround it in one training step (DNN, BCE-with-logits, the
embedding L2 term, global-norm clipping, Adam).  Every function cites the
reference file:line it follows (paths relative to the reference checkout).

Who may use it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — always as the *checker* or the *CPU
baseline*, never as a product code path.  Nothing under ``deepfm_amd/`` imports
this module; the product fails loudly when its HIP library is missing.

Pinning: the arithmetic of the reference lives in PyTorch ATen (torch 2.10.0,
pinned by the reference's uv.lock).  ``tools/make_golden.py`` imported the
reference's own layer classes from ``/root/reference`` in the build container,
ran them on seeded inputs and stored inputs/outputs/gradients under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below
against those vectors (and against the reference's own known answers: the FM
pairwise identity tests/test_layers.py:79-92, the padding-row test :43-51 and
the worked example notes/deepfm.md:72-90 = 67).

Conventions: all floating point is float32, indices int64.  ``params`` are
dicts keyed exactly like the reference ``state_dict`` of the module in question
(e.g. ``second_order_embeddings.C1.weight``).  Fields are described by plain
dicts ``{"name", "type" in {"sparse","dense","sequence"}, "vocab", "dim",
"max_len", "combiner"}`` in schema order.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32
Array = np.ndarray


# --------------------------------------------------------------------------- #
# FeatureEmbedding  (reference deepfm/models/layers/embedding.py)
# --------------------------------------------------------------------------- #

def _bag_pool(table: Array, ids: Array, combiner: str) -> Tuple[Array, Array]:
    """nn.EmbeddingBag(mode=combiner, padding_idx=0) over (B, L) ids
    (embedding.py:41-50, 91-94).  Pad entries (id 0) are excluded from the
    reduction and from the mean's divisor; an all-pad bag pools to 0.
    Returns (pooled (B,d), aux) where aux is count (B,) for sum/mean or the
    per-dimension arg-max position (B,d) for max (-1 where the bag is empty)."""
    rows = table[ids]                                   # (B, L, d)
    valid = ids != 0                                    # (B, L)
    count = valid.sum(axis=1)
    if combiner in ("sum", "mean"):
        pooled = (rows * valid[:, :, None].astype(F32)).sum(axis=1, dtype=F32)
        if combiner == "mean":
            pooled = pooled / np.maximum(count, 1).astype(F32)[:, None]
        return pooled.astype(F32), count
    if combiner == "max":
        masked = np.where(valid[:, :, None], rows, -np.inf)
        arg = masked.argmax(axis=1)                     # first maximum wins
        pooled = np.take_along_axis(masked, arg[:, None, :], axis=1)[:, 0, :]
        empty = count == 0
        pooled = np.where(empty[:, None], 0.0, pooled).astype(F32)
        arg = np.where(empty[:, None], -1, arg)
        return pooled, arg
    raise ValueError(f"unknown combiner {combiner!r}")


def embedding_forward(
    fields: Sequence[dict], params: Dict[str, Array], batch: Dict[str, Array], fm_dim: int
) -> Tuple[Array, Array, Array]:
    """FeatureEmbedding.forward (embedding.py:76-126).

    Returns first_order (B,1), field_embeddings (B,F,fm_dim), flat_embeddings (B,sum d).
    SPARSE: pure row gather, row 0 is the zero padding row (embedding.py:34-40, 95-98).
    DENSE: Linear(1,d) and Linear(1,1) on x[:,None] (embedding.py:51-56, 88-90).
    SEQUENCE: see _bag_pool.  Projection Linear(d, fm_dim, bias=False) where
    d != fm_dim (embedding.py:59-62, 110-115).  Assembly: stack/sum/cat in
    schema order (embedding.py:118-124)."""
    fo_parts: List[Array] = []
    fe_parts: List[Array] = []
    flat_parts: List[Array] = []
    for spec in fields:
        name, kind = spec["name"], spec["type"]
        x = batch[name]
        w2 = params[f"second_order_embeddings.{name}.weight"]
        w1 = params[f"first_order_embeddings.{name}.weight"]
        if kind == "dense":
            xv = x.astype(F32)[:, None]
            raw = xv * w2[:, 0][None, :] + params[f"second_order_embeddings.{name}.bias"][None, :]
            fo = xv * w1[0, 0] + params[f"first_order_embeddings.{name}.bias"][0]
        elif kind == "sequence":
            raw, _ = _bag_pool(w2, x, spec["combiner"])
            fo, _ = _bag_pool(w1, x, spec["combiner"])
        elif kind == "sparse":
            raw = w2[x]
            fo = w1[x]
        else:
            raise ValueError(kind)
        raw = raw.astype(F32)
        fo_parts.append(fo.astype(F32).reshape(-1, 1))
        flat_parts.append(raw)
        pkey = f"projections.{name}.weight"
        fe_parts.append((raw @ params[pkey].T).astype(F32) if pkey in params else raw)
    first_order = np.stack(fo_parts, axis=1).sum(axis=1, dtype=F32)
    field_embeddings = np.stack(fe_parts, axis=1)
    flat_embeddings = np.concatenate(flat_parts, axis=-1)
    return first_order, field_embeddings, flat_embeddings


def embedding_backward(
    fields: Sequence[dict],
    params: Dict[str, Array],
    batch: Dict[str, Array],
    fm_dim: int,
    d_first_order: Array,
    d_field_embeddings: Array,
    d_flat_embeddings: Array,
) -> Dict[str, Array]:
    """Autograd of embedding_forward w.r.t. every parameter, as *dense* arrays
    (the reference uses nn.Embedding(sparse=False): embedding.py:35-40).
    Duplicated ids accumulate; row 0 (padding_idx) receives no gradient."""
    grads: Dict[str, Array] = {}
    off = 0
    for fidx, spec in enumerate(fields):
        name, kind, d = spec["name"], spec["type"], spec["dim"]
        x = batch[name]
        k2 = f"second_order_embeddings.{name}.weight"
        k1 = f"first_order_embeddings.{name}.weight"
        w2, w1 = params[k2], params[k1]
        g_raw = d_flat_embeddings[:, off:off + d].astype(F32).copy()
        off += d
        g_fe = d_field_embeddings[:, fidx, :].astype(F32)
        pkey = f"projections.{name}.weight"
        if pkey in params:
            # forward: fe = raw @ P.T ; need raw for dP
            if kind == "dense":
                raw = x.astype(F32)[:, None] * w2[:, 0][None, :] + params[
                    f"second_order_embeddings.{name}.bias"][None, :]
            elif kind == "sequence":
                raw, _ = _bag_pool(w2, x, spec["combiner"])
            else:
                raw = w2[x]
            grads[pkey] = (g_fe.T @ raw.astype(F32)).astype(F32)
            g_raw += g_fe @ params[pkey]
        else:
            g_raw += g_fe
        g_fo = d_first_order.reshape(-1).astype(F32)
        if kind == "dense":
            xv = x.astype(F32)
            grads[k2] = (xv[:, None] * g_raw).sum(axis=0, dtype=F32)[:, None]
            grads[f"second_order_embeddings.{name}.bias"] = g_raw.sum(axis=0, dtype=F32)
            grads[k1] = np.array([[np.dot(xv, g_fo)]], dtype=F32)
            grads[f"first_order_embeddings.{name}.bias"] = np.array([g_fo.sum(dtype=F32)], dtype=F32)
        elif kind == "sparse":
            g2 = np.zeros_like(w2)
            g1 = np.zeros_like(w1)
            keep = x != 0
            np.add.at(g2, x[keep], g_raw[keep])
            np.add.at(g1[:, 0], x[keep], g_fo[keep])
            grads[k2], grads[k1] = g2, g1
        else:  # sequence
            g2 = np.zeros_like(w2)
            g1 = np.zeros_like(w1)
            comb = spec["combiner"]
            valid = x != 0
            if comb in ("sum", "mean"):
                count = np.maximum(valid.sum(axis=1), 1).astype(F32)
                scale = (1.0 / count if comb == "mean" else np.ones_like(count)).astype(F32)
                bsel, lsel = np.nonzero(valid)
                np.add.at(g2, x[bsel, lsel], g_raw[bsel] * scale[bsel, None])
                np.add.at(g1[:, 0], x[bsel, lsel], g_fo[bsel] * scale[bsel])
            else:  # max: gradient flows to the arg-max entry of every dimension
                _, arg2 = _bag_pool(w2, x, "max")
                _, arg1 = _bag_pool(w1, x, "max")
                for b in range(x.shape[0]):
                    for j in range(d):
                        if arg2[b, j] >= 0:
                            g2[x[b, arg2[b, j]], j] += g_raw[b, j]
                    if arg1[b, 0] >= 0:
                        g1[x[b, arg1[b, 0]], 0] += g_fo[b]
            grads[k2], grads[k1] = g2, g1
    return grads


# --------------------------------------------------------------------------- #
# FMInteraction  (reference deepfm/models/layers/fm.py:18-23)
# --------------------------------------------------------------------------- #

def fm_forward(field_embeddings: Array) -> Array:
    """0.5 * sum_d[(sum_f e)^2 - sum_f e^2]  ->  (B,1)."""
    e = field_embeddings.astype(F32)
    square_of_sum = e.sum(axis=1, dtype=F32) ** 2
    sum_of_squares = (e * e).sum(axis=1, dtype=F32)
    return (F32(0.5) * (square_of_sum - sum_of_squares)).sum(axis=1, keepdims=True, dtype=F32)


def fm_backward(field_embeddings: Array, d_out: Array) -> Array:
    """d e[b,f,d] = g[b] * (S[b,d] - e[b,f,d]),  S = sum_f e."""
    e = field_embeddings.astype(F32)
    s = e.sum(axis=1, keepdims=True, dtype=F32)
    return (d_out.reshape(-1, 1, 1).astype(F32) * (s - e)).astype(F32)


def fm_pairwise(field_embeddings: Array) -> Array:
    """Explicit O(F^2) pairwise inner products (tests/test_layers.py:79-92)."""
    e = field_embeddings.astype(np.float64)
    n = e.shape[1]
    total = np.zeros(e.shape[0])
    for i in range(n):
        for j in range(i + 1, n):
            total += (e[:, i, :] * e[:, j, :]).sum(axis=1)
    return total[:, None].astype(F32)


# --------------------------------------------------------------------------- #
# CIN  (reference deepfm/models/layers/cin.py)
# --------------------------------------------------------------------------- #

def cin_layout(num_fields: int, layer_sizes: Sequence[int], split_half: bool):
    """direct/next bookkeeping of CIN.__init__ (cin.py:41-64).
    Returns (in_channels per layer, direct_sizes, next_sizes, output_dim)."""
    in_ch, direct, nxt = [], [], []
    prev = num_fields
    for i, size in enumerate(layer_sizes):
        in_ch.append(prev * num_fields)
        if split_half and i < len(layer_sizes) - 1:
            d = size // 2
            direct.append(d)
            nxt.append(size - d)
            prev = size - d
        else:
            direct.append(size)
            nxt.append(size)
            prev = size
    return in_ch, direct, nxt, sum(direct)


def cin_forward(
    x0: Array, params: Dict[str, Array], layer_sizes: Sequence[int], split_half: bool,
    return_cache: bool = False,
):
    """CIN.forward (cin.py:66-105).  Per layer: Z[b,h*F+f,d] = hidden[b,h,d]*x0[b,f,d]
    (:84-87), Y = relu(W Z + bias) (:90-91), split [direct | next] along channels,
    direct first (:93-96), sum-pool direct over d (:102), concat (:105)."""
    x0 = x0.astype(F32)
    B, Fn, D = x0.shape
    _, direct, nxt, _ = cin_layout(Fn, layer_sizes, split_half)
    hidden = x0
    outs, cache = [], []
    for i in range(len(layer_sizes)):
        W = params[f"conv_layers.{i}.weight"][:, :, 0]          # (C, H*F)
        bias = params[f"conv_layers.{i}.bias"]
        Z = (hidden[:, :, None, :] * x0[:, None, :, :]).reshape(B, -1, D)
        Y = np.matmul(W[None], Z) + bias[None, :, None]           # (B, C, D)
        Y = np.maximum(Y, 0).astype(F32)
        cache.append((hidden, Y))
        if split_half and i < len(layer_sizes) - 1:
            dpart, hidden = Y[:, :direct[i], :], Y[:, direct[i]:, :]
        else:
            dpart, hidden = Y, Y
        outs.append(dpart.sum(axis=2, dtype=F32))
    out = np.concatenate(outs, axis=1)
    return (out, cache) if return_cache else out


def cin_backward(
    x0: Array, params: Dict[str, Array], layer_sizes: Sequence[int], split_half: bool,
    d_out: Array,
) -> Tuple[Array, Dict[str, Array]]:
    """Autograd of cin_forward: returns (d x0, {param grads})."""
    x0 = x0.astype(F32)
    B, Fn, D = x0.shape
    _, direct, nxt, _ = cin_layout(Fn, layer_sizes, split_half)
    _, cache = cin_forward(x0, params, layer_sizes, split_half, return_cache=True)
    L = len(layer_sizes)
    grads: Dict[str, Array] = {}
    d_x0 = np.zeros_like(x0)
    d_hidden_next: Optional[Array] = None      # gradient w.r.t. layer i's "next" part
    col = int(sum(direct))
    for i in reversed(range(L)):
        hidden, Y = cache[i]
        C = Y.shape[1]
        col -= direct[i]
        g_direct = np.broadcast_to(d_out[:, col:col + direct[i], None], (B, direct[i], D))
        dY = np.zeros_like(Y)
        last_or_nosplit = not (split_half and i < L - 1)
        if last_or_nosplit:
            dY += g_direct
            if d_hidden_next is not None:      # no-split: hidden == whole Y
                dY += d_hidden_next
        else:
            dY[:, :direct[i], :] = g_direct
            if d_hidden_next is not None:
                dY[:, direct[i]:, :] = d_hidden_next
        dY = (dY * (Y > 0)).astype(F32)
        W = params[f"conv_layers.{i}.weight"][:, :, 0]
        H = hidden.shape[1]
        Z = (hidden[:, :, None, :] * x0[:, None, :, :]).reshape(B, -1, D)
        grads[f"conv_layers.{i}.weight"] = np.einsum("bcd,bkd->ck", dY, Z, optimize=True).astype(F32)[:, :, None]
        grads[f"conv_layers.{i}.bias"] = dY.sum(axis=(0, 2), dtype=F32)
        dZ = np.matmul(W.T[None], dY).reshape(B, H, Fn, D)
        d_h = (dZ * x0[:, None, :, :]).sum(axis=2, dtype=F32)
        d_x0 += (dZ * hidden[:, :, None, :]).sum(axis=1, dtype=F32)
        if i == 0:
            d_x0 += d_h                        # hidden_0 is x0 itself
            d_hidden_next = None
        else:
            d_hidden_next = d_h
    return d_x0.astype(F32), grads


# --------------------------------------------------------------------------- #
# MultiHeadSelfAttention  (reference deepfm/models/layers/attention.py)
# --------------------------------------------------------------------------- #

LN_EPS = 1e-5  # nn.LayerNorm default (attention.py:88)


def _attn_block_forward(x: Array, p: Dict[str, Array], prefix: str, heads: int, residual: bool):
    """_AttentionBlock.forward (attention.py:91-120)."""
    B, Fn, D = x.shape
    Wq, bq = p[prefix + "W_q.weight"], p[prefix + "W_q.bias"]
    Wk, bk = p[prefix + "W_k.weight"], p[prefix + "W_k.bias"]
    Wv, bv = p[prefix + "W_v.weight"], p[prefix + "W_v.bias"]
    Wo, bo = p[prefix + "W_out.weight"], p[prefix + "W_out.bias"]
    A = Wq.shape[0]
    hd = A // heads
    scale = F32(math.sqrt(hd))
    Q = (x @ Wq.T + bq).reshape(B, Fn, heads, hd).transpose(0, 2, 1, 3)
    K = (x @ Wk.T + bk).reshape(B, Fn, heads, hd).transpose(0, 2, 1, 3)
    V = (x @ Wv.T + bv).reshape(B, Fn, heads, hd).transpose(0, 2, 1, 3)
    S = np.matmul(Q, K.transpose(0, 1, 3, 2)) / scale              # (B,h,F,F)
    S = S - S.max(axis=-1, keepdims=True)
    P = np.exp(S)
    P = (P / P.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    O = np.matmul(P, V).transpose(0, 2, 1, 3).reshape(B, Fn, A)     # concat heads
    out = (O @ Wo.T + bo).astype(F32)
    cache = dict(x=x, Q=Q, K=K, V=V, P=P, O=O)
    if residual:
        y = out + x
        mu = y.mean(axis=-1, keepdims=True, dtype=F32)
        var = ((y - mu) ** 2).mean(axis=-1, keepdims=True, dtype=F32)
        rstd = (1.0 / np.sqrt(var + F32(LN_EPS))).astype(F32)
        xhat = ((y - mu) * rstd).astype(F32)
        out = (xhat * p[prefix + "layer_norm.weight"] + p[prefix + "layer_norm.bias"]).astype(F32)
        cache.update(xhat=xhat, rstd=rstd)
    return out, cache


def attention_forward(
    x: Array, params: Dict[str, Array], num_heads: int, num_layers: int, use_residual: bool,
    return_cache: bool = False,
):
    """MultiHeadSelfAttention.forward (attention.py:52-64): stacked blocks."""
    x = x.astype(F32)
    caches = []
    for li in range(num_layers):
        x, c = _attn_block_forward(x, params, f"layers.{li}.", num_heads, use_residual)
        caches.append(c)
    return (x, caches) if return_cache else x


def attention_backward(
    x: Array, params: Dict[str, Array], num_heads: int, num_layers: int, use_residual: bool,
    d_out: Array,
) -> Tuple[Array, Dict[str, Array]]:
    """Autograd of attention_forward: returns (d x, {param grads})."""
    _, caches = attention_forward(x, params, num_heads, num_layers, use_residual, return_cache=True)
    grads: Dict[str, Array] = {}
    g = d_out.astype(F32)
    for li in reversed(range(num_layers)):
        pre = f"layers.{li}."
        c = caches[li]
        xin = c["x"]
        B, Fn, D = xin.shape
        Wq, Wk, Wv, Wo = (params[pre + n + ".weight"] for n in ("W_q", "W_k", "W_v", "W_out"))
        A = Wq.shape[0]
        hd = A // num_heads
        scale = F32(math.sqrt(hd))
        d_x = np.zeros_like(xin)
        if use_residual:
            gamma = params[pre + "layer_norm.weight"]
            xhat, rstd = c["xhat"], c["rstd"]
            grads[pre + "layer_norm.weight"] = (g * xhat).sum(axis=(0, 1), dtype=F32)
            grads[pre + "layer_norm.bias"] = g.sum(axis=(0, 1), dtype=F32)
            gx = g * gamma
            d_y = rstd * (gx - gx.mean(axis=-1, keepdims=True, dtype=F32)
                          - xhat * (gx * xhat).mean(axis=-1, keepdims=True, dtype=F32))
            d_x += d_y
            g_lin = d_y
        else:
            g_lin = g
        O = c["O"]
        grads[pre + "W_out.weight"] = np.einsum("bfd,bfa->da", g_lin, O, optimize=True).astype(F32)
        grads[pre + "W_out.bias"] = g_lin.sum(axis=(0, 1), dtype=F32)
        dO = (g_lin @ Wo).reshape(B, Fn, num_heads, hd).transpose(0, 2, 1, 3)   # (B,h,F,hd)
        P, Q, K, V = c["P"], c["Q"], c["K"], c["V"]
        dV = np.matmul(P.transpose(0, 1, 3, 2), dO)
        dP = np.matmul(dO, V.transpose(0, 1, 3, 2))
        dS = P * (dP - (dP * P).sum(axis=-1, keepdims=True, dtype=F32))
        dS = (dS / scale).astype(F32)
        dQ = np.matmul(dS, K)
        dK = np.matmul(dS.transpose(0, 1, 3, 2), Q)
        for nm, dT in (("W_q", dQ), ("W_k", dK), ("W_v", dV)):
            dT2 = dT.transpose(0, 2, 1, 3).reshape(B, Fn, A).astype(F32)
            grads[pre + nm + ".weight"] = np.einsum("bfa,bfd->ad", dT2, xin, optimize=True).astype(F32)
            grads[pre + nm + ".bias"] = dT2.sum(axis=(0, 1), dtype=F32)
            d_x += dT2 @ params[pre + nm + ".weight"]
        g = d_x.astype(F32)
    return g, grads


# --------------------------------------------------------------------------- #
# DNN, heads, loss  (reference dnn.py:45-59, deepfm.py:30-42, trainer.py:59)
# These are the callers around the hot path; restated so that whole-model
# logits and one full training step can be checked and timed on the CPU.
# --------------------------------------------------------------------------- #

BN_EPS = 1e-5  # nn.BatchNorm1d default


def _act(name: str, z: Array) -> Array:
    if name == "relu":
        return np.maximum(z, 0)
    if name == "leaky_relu":
        return np.where(z > 0, z, F32(0.01) * z)
    if name == "tanh":
        return np.tanh(z)
    if name == "gelu":
        from scipy.special import erf
        return z * F32(0.5) * (1 + erf(z / np.sqrt(F32(2.0))))
    raise ValueError(name)


def _act_grad(name: str, z: Array) -> Array:
    if name == "relu":
        return (z > 0).astype(F32)
    if name == "leaky_relu":
        return np.where(z > 0, F32(1.0), F32(0.01)).astype(F32)
    if name == "tanh":
        return (1 - np.tanh(z) ** 2).astype(F32)
    if name == "gelu":
        from scipy.special import erf
        cdf = 0.5 * (1 + erf(z / np.sqrt(2.0)))
        pdf = np.exp(-0.5 * z * z) / np.sqrt(2 * np.pi)
        return (cdf + z * pdf).astype(F32)
    raise ValueError(name)


def dnn_forward(
    x: Array, params: Dict[str, Array], n_layers: int, activation: str = "relu",
    use_batch_norm: bool = True, training: bool = False, prefix: str = "mlp.",
    return_cache: bool = False,
):
    """DNN.forward (dnn.py:45-59): [Linear -> BatchNorm1d -> act -> Dropout] * n.
    Dropout is the identity here (eval mode, or p=0 in parity runs — survey §7.5).
    Module index inside nn.Sequential: 4*i (+1 for BN) with BN, 3*i without."""
    stride = 4 if use_batch_norm else 3
    h = x.astype(F32)
    cache = []
    for i in range(n_layers):
        W = params[f"{prefix}{stride * i}.weight"]
        b = params[f"{prefix}{stride * i}.bias"]
        lin_in = h
        z = (h @ W.T + b).astype(F32)
        bn = None
        if use_batch_norm:
            gk = f"{prefix}{stride * i + 1}."
            if training:
                mu = z.mean(axis=0, dtype=F32)
                var = ((z - mu) ** 2).mean(axis=0, dtype=F32)          # biased, as BN normalises
            else:
                mu, var = params[gk + "running_mean"], params[gk + "running_var"]
            rstd = (1.0 / np.sqrt(var + F32(BN_EPS))).astype(F32)
            zhat = ((z - mu) * rstd).astype(F32)
            bn = (zhat, rstd)
            z = (zhat * params[gk + "weight"] + params[gk + "bias"]).astype(F32)
        cache.append((lin_in, z, bn))
        h = _act(activation, z).astype(F32)
    return (h, cache) if return_cache else h


def dnn_backward(
    x: Array, params: Dict[str, Array], n_layers: int, d_out: Array, activation: str = "relu",
    use_batch_norm: bool = True, training: bool = False, prefix: str = "mlp.",
) -> Tuple[Array, Dict[str, Array]]:
    stride = 4 if use_batch_norm else 3
    _, cache = dnn_forward(x, params, n_layers, activation, use_batch_norm, training, prefix, True)
    grads: Dict[str, Array] = {}
    g = d_out.astype(F32)
    for i in reversed(range(n_layers)):
        lin_in, z, bn = cache[i]
        g = (g * _act_grad(activation, z)).astype(F32)
        if use_batch_norm:
            gk = f"{prefix}{stride * i + 1}."
            zhat, rstd = bn
            grads[gk + "weight"] = (g * zhat).sum(axis=0, dtype=F32)
            grads[gk + "bias"] = g.sum(axis=0, dtype=F32)
            gz = g * params[gk + "weight"]
            if training:
                g = rstd * (gz - gz.mean(axis=0, dtype=F32) - zhat * (gz * zhat).mean(axis=0, dtype=F32))
            else:
                g = gz * rstd
            g = g.astype(F32)
        W = params[f"{prefix}{stride * i}.weight"]
        grads[f"{prefix}{stride * i}.weight"] = (g.T @ lin_in).astype(F32)
        grads[f"{prefix}{stride * i}.bias"] = g.sum(axis=0, dtype=F32)
        g = (g @ W).astype(F32)
    return g, grads


def linear_forward(x: Array, params: Dict[str, Array], prefix: str) -> Array:
    return (x @ params[prefix + "weight"].T + params[prefix + "bias"]).astype(F32)


def bce_with_logits(logits: Array, labels: Array) -> Tuple[F32, Array]:
    """nn.BCEWithLogitsLoss() mean reduction (trainer.py:59, 221): (loss, d logits)."""
    z = logits.reshape(-1).astype(F32)
    y = labels.reshape(-1).astype(F32)
    loss = (np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))).mean(dtype=F32)
    sig = 1.0 / (1.0 + np.exp(-z))
    return F32(loss), ((sig - y) / F32(z.size)).astype(F32).reshape(logits.shape)


def _sub(params: Dict[str, Array], prefix: str) -> Dict[str, Array]:
    n = len(prefix)
    return {k[n:]: v for k, v in params.items() if k.startswith(prefix)}


def model_logits(model: str, fields, params: Dict[str, Array], batch, cfg: dict,
                 training: bool = False) -> Array:
    """BaseCTRModel.forward (base.py:59-68) + the three _forward_components:
    deepfm.py:30-42, xdeepfm.py:36-48, attention_deepfm.py:48-66.
    ``cfg`` keys: fm_dim, hidden_units, activation, use_batch_norm, and for
    xdeepfm cin_layer_sizes/cin_split_half, for attention_deepfm num_heads/
    num_layers/use_residual."""
    fo, fe, fl = embedding_forward(fields, _sub(params, "embedding."), batch, cfg["fm_dim"])
    n_hidden = len(cfg["hidden_units"])
    dnn_p = _sub(params, "dnn.")
    act, bn = cfg.get("activation", "relu"), cfg.get("use_batch_norm", True)
    if model == "deepfm":
        dnn_out = linear_forward(dnn_forward(fl, dnn_p, n_hidden, act, bn, training), params, "output_linear.")
        return (fo + fm_forward(fe) + dnn_out).astype(F32)
    if model == "xdeepfm":
        cin_out = cin_forward(fe, _sub(params, "cin."), cfg["cin_layer_sizes"], cfg["cin_split_half"])
        cin_lin = linear_forward(cin_out, params, "cin_linear.")
        dnn_out = linear_forward(dnn_forward(fl, dnn_p, n_hidden, act, bn, training), params, "dnn_linear.")
        return (fo + cin_lin + dnn_out).astype(F32)
    if model == "attention_deepfm":
        att = attention_forward(fe, _sub(params, "attention."), cfg["num_heads"], cfg["num_layers"],
                                cfg["use_residual"])
        dnn_in = np.concatenate([att.reshape(att.shape[0], -1), fl], axis=1)
        dnn_out = linear_forward(dnn_forward(dnn_in, dnn_p, n_hidden, act, bn, training), params, "output_linear.")
        return (fo + fm_forward(fe) + dnn_out).astype(F32)
    raise ValueError(f"Unknown model: {model}")


# --------------------------------------------------------------------------- #
# Row-sparse view of the embedding gradient, L2 term, clip, Adam
# (reference base.py:78-83, trainer.py:224-237).  The reference's step is dense;
# the row-wise form below is the build's documented fast mode:
# only rows touched by the batch are updated, L2 is applied lazily to them.
# --------------------------------------------------------------------------- #

def rowsparse_from_batch(ids: Array, g_rows: Array, g_first: Array):
    """Deterministic row-wise reduction for one SPARSE field.
    ids (n,) int64, g_rows (n,d), g_first (n,).  Returns (unique ids ascending
    without 0, summed row grads, summed first-order grads); contributions are
    added in increasing sample order."""
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    keep = sid != 0
    order, sid = order[keep], sid[keep]
    if sid.size == 0:
        return sid, np.zeros((0, g_rows.shape[1]), F32), np.zeros((0,), F32)
    starts = np.flatnonzero(np.r_[True, sid[1:] != sid[:-1]])
    uniq = sid[starts]
    ends = np.r_[starts[1:], sid.size]
    out2 = np.zeros((uniq.size, g_rows.shape[1]), F32)
    out1 = np.zeros((uniq.size,), F32)
    for u, (s, e) in enumerate(zip(starts, ends)):
        acc2 = np.zeros(g_rows.shape[1], F32)
        acc1 = F32(0)
        for p in order[s:e]:
            acc2 = (acc2 + g_rows[p]).astype(F32)
            acc1 = F32(acc1 + g_first[p])
        out2[u], out1[u] = acc2, acc1
    return uniq, out2, out1


def adam_update(w: Array, m: Array, v: Array, g: Array, step: int, lr: float,
                beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam single-tensor update (trainer.py:67-70, 237), in place, fp32.
    m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
    w -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    g = g.astype(F32)
    m *= F32(beta1)
    m += F32(1 - beta1) * g
    v *= F32(beta2)
    v += F32(1 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps)
    w -= F32(lr / bc1) * (m / denom)


def clip_coef(total_sq_norm: float, max_norm: float) -> F32:
    """clip_grad_norm_ scale (trainer.py:232-235): min(1, max_norm/(norm+1e-6))."""
    norm = math.sqrt(float(total_sq_norm))
    return F32(min(1.0, max_norm / (norm + 1e-6)))


def l2_reg_loss(params: Dict[str, Array], l2: float) -> F32:
    """BaseCTRModel.get_l2_reg_loss (base.py:78-83): lambda * sum ||p||_2^2 over every parameter of
    ``model.embedding`` — tables, DENSE-field Linears and projections (state_dict keys ``embedding.*``)."""
    total = 0.0
    for k, v in params.items():
        if k.startswith("embedding."):
            total += float((v.astype(np.float64) ** 2).sum())
    return F32(l2 * total)


def rowsparse_reduce_fast(ids: Array, g_rows: Array, g_first: Array):
    """Vectorised rowsparse_from_batch (same result up to fp32 summation order inside a
    run of duplicates): used by the CPU-baseline timing and the large-shape checks."""
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    keep = sid != 0
    order, sid = order[keep], sid[keep]
    if sid.size == 0:
        return sid, np.zeros((0, g_rows.shape[1]), F32), np.zeros((0,), F32)
    starts = np.flatnonzero(np.r_[True, sid[1:] != sid[:-1]])
    return (sid[starts], np.add.reduceat(g_rows[order], starts, axis=0).astype(F32),
            np.add.reduceat(g_first[order], starts).astype(F32))


def train_step_rowsparse(model: str, fields, params: Dict[str, Array], state: Dict[str, Array],
                         batch, labels: Array, cfg: dict, hp: dict, step: int,
                         exact_order: bool = False, info: Optional[dict] = None) -> F32:
    """One training step of ``model`` in {"deepfm", "xdeepfm", "attention_deepfm"} in the build's
    row-sparse mode, in place on ``params`` / ``state`` (``m/<key>``, ``v/<key>`` Adam moments): forward
    (embedding.py:76-126 + the model's ``_forward_components``: deepfm.py:30-42, xdeepfm.py:36-48,
    attention_deepfm.py:48-66), BCE + L2 on the non-table embedding parameters (trainer.py:221-225),
    backward, lazy L2 on touched rows, global-norm clip (trainer.py:232-235), Adam (trainer.py:237).
    hp: lr, l2, max_grad_norm, betas, eps.  ``cfg`` as for ``model_logits``.
    ``info`` (optional) receives logits, the squared global gradient norm, the clip coefficient, the
    per-field row gradients (ids, rows incl. the L2 term, first-order) and the dense gradients."""
    emb_p = _sub(params, "embedding.")
    fo, fe, fl = embedding_forward(fields, emb_p, batch, cfg["fm_dim"])
    B = fe.shape[0]
    n_hidden = len(cfg["hidden_units"])
    dnn_p = _sub(params, "dnn.")
    act, bn = cfg.get("activation", "relu"), cfg.get("use_batch_norm", True)
    grads: Dict[str, Array] = {}

    def head_backward(prefix, x_in, dz):
        grads[prefix + "weight"] = (dz.T @ x_in).astype(F32)
        grads[prefix + "bias"] = dz.sum(axis=0, dtype=F32)
        return (dz @ params[prefix + "weight"]).astype(F32)

    if model == "deepfm":
        h = dnn_forward(fl, dnn_p, n_hidden, act, bn, training=True)
        logits = fo + fm_forward(fe) + linear_forward(h, params, "output_linear.")
        loss, dz = bce_with_logits(logits, labels)
        d_fl, dnn_g = dnn_backward(fl, dnn_p, n_hidden, head_backward("output_linear.", h, dz), act, bn, training=True)
        d_fe = fm_backward(fe, dz) + d_fl.reshape(fe.shape)  # uniform schema: flat == fe reshaped
    elif model == "xdeepfm":
        cin_p = _sub(params, "cin.")
        sizes, split = cfg["cin_layer_sizes"], cfg["cin_split_half"]
        cin_out = cin_forward(fe, cin_p, sizes, split)
        h = dnn_forward(fl, dnn_p, n_hidden, act, bn, training=True)
        logits = fo + linear_forward(cin_out, params, "cin_linear.") + linear_forward(h, params, "dnn_linear.")
        loss, dz = bce_with_logits(logits, labels)
        d_cin, cin_g = cin_backward(fe, cin_p, sizes, split, head_backward("cin_linear.", cin_out, dz))
        grads.update({"cin." + k: v for k, v in cin_g.items()})
        d_fl, dnn_g = dnn_backward(fl, dnn_p, n_hidden, head_backward("dnn_linear.", h, dz), act, bn, training=True)
        d_fe = d_cin + d_fl.reshape(fe.shape)                # no FM term in xDeepFM (xdeepfm.py:36-48)
    elif model == "attention_deepfm":
        att_p = _sub(params, "attention.")
        heads, layers, res = cfg["num_heads"], cfg["num_layers"], cfg["use_residual"]
        att = attention_forward(fe, att_p, heads, layers, res)
        dnn_in = np.concatenate([att.reshape(B, -1), fl], axis=1)
        h = dnn_forward(dnn_in, dnn_p, n_hidden, act, bn, training=True)
        logits = fo + fm_forward(fe) + linear_forward(h, params, "output_linear.")
        loss, dz = bce_with_logits(logits, labels)
        d_in, dnn_g = dnn_backward(dnn_in, dnn_p, n_hidden, head_backward("output_linear.", h, dz), act, bn, training=True)
        n_att = att.shape[1] * att.shape[2]
        d_att, att_g = attention_backward(fe, att_p, heads, layers, res, d_in[:, :n_att].reshape(att.shape))
        grads.update({"attention." + k: v for k, v in att_g.items()})
        d_fe = fm_backward(fe, dz) + d_in[:, n_att:].reshape(fe.shape) + d_att   # FM uses the un-attended fe
    else:
        raise ValueError(f"Unknown model: {model}")
    grads.update({"dnn." + k: v for k, v in dnn_g.items()})
    l2 = F32(hp["l2"])
    rows = {}
    sq = 0.0
    for fidx, spec in enumerate(fields):
        name = spec["name"]
        k2, k1 = f"embedding.second_order_embeddings.{name}.weight", f"embedding.first_order_embeddings.{name}.weight"
        if spec["type"] == "sparse":
            red = rowsparse_from_batch if exact_order else rowsparse_reduce_fast
            uniq, r2, r1 = red(batch[name], d_fe[:, fidx, :], dz[:, 0])
            r2 = (r2 + 2 * l2 * params[k2][uniq]).astype(F32)
            r1 = (r1 + 2 * l2 * params[k1][uniq, 0]).astype(F32)
            rows[name] = (uniq, r2, r1)
            sq += float((r2.astype(np.float64) ** 2).sum() + (r1.astype(np.float64) ** 2).sum())
        else:
            x = batch[name].astype(F32)
            g = d_fe[:, fidx, :]
            b2k, b1k = k2.replace("weight", "bias"), k1.replace("weight", "bias")
            grads[k2] = ((x[:, None] * g).sum(axis=0, dtype=F32)[:, None] + 2 * l2 * params[k2]).astype(F32)
            grads[b2k] = (g.sum(axis=0, dtype=F32) + 2 * l2 * params[b2k]).astype(F32)
            grads[k1] = (np.array([[np.dot(x, dz[:, 0])]], dtype=F32) + 2 * l2 * params[k1]).astype(F32)
            grads[b1k] = (np.array([dz[:, 0].sum(dtype=F32)], dtype=F32) + 2 * l2 * params[b1k]).astype(F32)
    for g in grads.values():
        sq += float((g.astype(np.float64) ** 2).sum())
    coef = clip_coef(sq, hp["max_grad_norm"]) if hp.get("max_grad_norm") else F32(1.0)
    if info is not None:
        info.update(logits=logits, sq_norm=sq, coef=coef, rows=rows, grads=grads, d_fe=d_fe)
    b1, b2 = hp.get("betas", (0.9, 0.999))
    eps = hp.get("eps", 1e-8)
    for k, g in grads.items():
        adam_update(params[k], state["m/" + k], state["v/" + k], g * coef, step, hp["lr"], b1, b2, eps)
    for name, (uniq, r2, r1) in rows.items():
        for key, g in ((f"embedding.second_order_embeddings.{name}.weight", r2),
                       (f"embedding.first_order_embeddings.{name}.weight", r1[:, None])):
            w, m, v = params[key][uniq], state["m/" + key][uniq], state["v/" + key][uniq]
            adam_update(w, m, v, g * coef, step, hp["lr"], b1, b2, eps)
            params[key][uniq], state["m/" + key][uniq], state["v/" + key][uniq] = w, m, v
    return loss


def deepfm_train_step_rowsparse(fields, params, state, batch, labels, cfg, hp, step,
                                exact_order: bool = False, info: Optional[dict] = None) -> F32:
    """``train_step_rowsparse("deepfm", ...)`` (the name the round-1/2 tests and bench.py use)."""
    return train_step_rowsparse("deepfm", fields, params, state, batch, labels, cfg, hp, step, exact_order, info)
