"""Shared helpers for the test-suite: golden loading, tolerances, schema glue."""
from __future__ import annotations

import json
import os
from typing import Dict

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Floating-point bar of BASELINE.json: logits within 1e-4 relative of the reference
# CPU path.  `assert_close` applies it element-wise with an absolute floor tied to
# the magnitude of the tensor (an element that is ~0 by cancellation cannot be held
# to a relative bound).
RTOL = 1e-4


def load(name: str) -> Dict[str, np.ndarray]:
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def group(d: Dict[str, np.ndarray], prefix: str) -> Dict[str, np.ndarray]:
    n = len(prefix)
    return {k[n:]: v for k, v in d.items() if k.startswith(prefix)}


def fields_of(d) -> list:
    return json.loads(str(d["fields"]))


def cfg_of(d) -> dict:
    return json.loads(str(d["cfg"]))


def assert_close(got, want, rtol: float = RTOL, atol_scale: float = 1e-5, what: str = "", floor: float = 0.0):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    scale = max(float(np.abs(want).max()) if want.size else 0.0, 1e-30)
    err = np.abs(got - want)
    bound = rtol * np.abs(want) + atol_scale * scale + floor
    bad = err > bound
    if bad.any():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(
            f"{what}: {bad.sum()} / {bad.size} elements out of tolerance; worst at {i}: "
            f"got {got[i]!r} want {want[i]!r} (|err| {err[i]:.3e}, bound {bound[i]:.3e})")


def hashed_weights(shape, salt: int, scale: float) -> np.ndarray:
    """Same closed form as tools/make_golden.py::hashed_weights (exact integer arithmetic)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503)) % np.uint64(1 << 32)
    h = (h ^ (h >> np.uint64(15))) * np.uint64(2246822519) % np.uint64(1 << 32)
    h = (h ^ (h >> np.uint64(13))) % np.uint64(1 << 24)
    v = (h.astype(np.float64) / float(1 << 24) - 0.5) * 2.0 * scale
    return v.astype(np.float32).reshape(shape)


def cin_full_params(F=39, layer_sizes=(128, 128, 128), split_half=True):
    """Parameters of the `cin_criteo_full` golden case (not stored: recomputed)."""
    params = {}
    prev = F
    for li, size in enumerate(layer_sizes):
        k = prev * F
        params[f"conv_layers.{li}.weight"] = hashed_weights((size, k, 1), 2 * li + 1, 2.0 / np.sqrt(k))
        params[f"conv_layers.{li}.bias"] = hashed_weights((size,), 2 * li + 2, 0.1)
        prev = size - size // 2 if (split_half and li < len(layer_sizes) - 1) else size
    return params


# ---- glue between the golden "fields" description and the package's schema objects ----

from deepfm_amd.data.synthetic import random_fields_batch, schema_from_fields  # noqa: E402,F401  (package helpers)


def load_params(module, params: Dict[str, np.ndarray], device="cuda"):
    """Copy golden/oracle parameters (state_dict-keyed numpy arrays) into a torch module."""
    import torch
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}
    missing, unexpected = module.load_state_dict(sd, strict=True), None
    return module.to(device)


def to_device_batch(batch: Dict[str, np.ndarray], device="cuda"):
    import torch
    return {k: torch.from_numpy(v).to(device) for k, v in batch.items()}


def npy(t):
    return t.detach().cpu().numpy()


def assert_close_mostly(got, want, max_bad_frac: float, rtol: float = RTOL, atol_scale: float = 1e-5, what: str = ""):
    """assert_close that tolerates a bounded fraction of outliers.  Used only where a ReLU kink
    makes the gradient discontinuous: an activation within rounding distance of 0 may land on the
    other side of the kink in fp32-vs-split-bf16 arithmetic, which flips one (sample, d) column of
    the CIN gradients.  Everything else must still meet the normal bar."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    scale = max(float(np.abs(want).max()) if want.size else 0.0, 1e-30)
    bad = np.abs(got - want) > rtol * np.abs(want) + atol_scale * scale
    frac = bad.mean() if bad.size else 0.0
    assert frac <= max_bad_frac, f"{what}: {bad.sum()} / {bad.size} elements ({frac:.2e}) out of tolerance"
