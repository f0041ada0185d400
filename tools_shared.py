"""Schema builders shared by tests, bench.py and __graft_entry__ (plain-dict field lists)."""


# Distinct values of C1..C26 in the public Criteo display-advertising (Kaggle) training set: three fields with
# fewer than 11 ids, seven with fewer than 64, four with millions — what "Criteo-shaped" means for the
# row plan and the row gradients (runs of one id inside a batch from 1 to B/3).
CRITEO_KAGGLE_CARDINALITIES = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27,
                               14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def criteo_fields(vocab, dim: int, n_sparse: int = 26, n_dense: int = 13):
    """BASELINE.json Criteo shape: C1..C26 SPARSE then I1..I13 DENSE.  ``vocab``: one
    vocabulary size for every SPARSE field, or a list with one per field."""
    vocabs = list(vocab) if isinstance(vocab, (list, tuple)) else [vocab] * n_sparse
    if len(vocabs) != n_sparse:
        raise ValueError(f"{len(vocabs)} vocabulary sizes for {n_sparse} SPARSE fields")
    fs = [dict(name=f"C{i + 1}", type="sparse", vocab=int(vocabs[i]), dim=dim, max_len=1, combiner="mean")
          for i in range(n_sparse)]
    fs += [dict(name=f"I{i + 1}", type="dense", vocab=0, dim=dim, max_len=1, combiner="mean")
           for i in range(n_dense)]
    return fs
