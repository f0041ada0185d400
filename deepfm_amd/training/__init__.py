"""Train-step tail for the HIP path: row-wise Adam over touched rows + data-parallel exchange."""
from deepfm_amd.training.rowsparse import RowSparseAdam  # noqa: F401
