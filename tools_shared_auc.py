"""The learnable synthetic CTR task shared by tools/make_auc_golden.py (reference run, build container)
and tests/test_gpu_auc_parity.py (this package, GPU box): same numpy generator, same split."""
import numpy as np

N_SPARSE, N_DENSE, VOCAB, DIM = 26, 13, 50, 16
N_TRAIN, N_TEST, BATCH, EPOCHS = 65536, 8192, 512, 4


def make_task(seed: int = 2024):
    """ids (N, 26) int64 in [1, VOCAB), dense (N, 13) float32 in [0, 1), labels (N,) float32.
    Teacher: per-field id effects + dense slopes + a few pairwise id interactions -> sigmoid -> Bernoulli."""
    rng = np.random.default_rng(seed)
    n = N_TRAIN + N_TEST
    ids = rng.integers(1, VOCAB, size=(n, N_SPARSE)).astype(np.int64)
    dense = rng.random((n, N_DENSE)).astype(np.float32)
    eff = rng.normal(0.0, 0.6, size=(N_SPARSE, VOCAB))
    slope = rng.normal(0.0, 0.8, size=N_DENSE)
    u = rng.normal(0.0, 0.6, size=(N_SPARSE, VOCAB, 4))
    logit = eff[np.arange(N_SPARSE)[None, :], ids].sum(1) + (dense - 0.5) @ slope
    for a, b in ((0, 1), (2, 3), (4, 5), (6, 7)):
        logit += (u[a, ids[:, a]] * u[b, ids[:, b]]).sum(1)
    logit -= 1.0
    labels = (rng.random(n) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    return ids, dense, labels


def epoch_order(epoch: int) -> np.ndarray:
    return np.random.default_rng(7 + epoch).permutation(N_TRAIN)
