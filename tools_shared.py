"""Schema builders shared by tests and tools: re-exported from the package (deepfm_amd/data/synthetic.py)."""
from deepfm_amd.data.synthetic import CRITEO_KAGGLE_CARDINALITIES, criteo_fields  # noqa: F401
