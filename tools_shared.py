"""Schema builders shared by tests, bench.py and __graft_entry__ (plain-dict field lists)."""


def criteo_fields(vocab: int, dim: int, n_sparse: int = 26, n_dense: int = 13):
    """BASELINE.json Criteo shape: C1..C26 SPARSE then I1..I13 DENSE."""
    fs = [dict(name=f"C{i + 1}", type="sparse", vocab=vocab, dim=dim, max_len=1, combiner="mean")
          for i in range(n_sparse)]
    fs += [dict(name=f"I{i + 1}", type="dense", vocab=0, dim=dim, max_len=1, combiner="mean")
           for i in range(n_dense)]
    return fs
