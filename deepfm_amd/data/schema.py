"""Feature contract consumed by every layer on the hot path.

Host-side mirror of the reference's ``deepfm/data/schema.py:7-59`` (same class
names, field names, defaults and derived properties) so that schemas built for
the reference construct the MI355X modules unchanged.  Field order is the
insertion order of ``DatasetSchema.fields`` — it fixes the field axis of
``field_embeddings`` and the column order of ``flat_embeddings``
(reference ``deepfm/models/layers/embedding.py:26``).
"""

from __future__ import annotations

import enum
from dataclasses import dataclass, field as _dc_field
from typing import Dict, List


class FeatureType(enum.Enum):
    """Kind of input column (reference schema.py:7-10)."""

    SPARSE = "sparse"      # one int64 id per sample, id 0 = OOV / padding
    DENSE = "dense"        # one float32 value per sample
    SEQUENCE = "sequence"  # (B, L) int64 ids, 0-padded, pooled by `combiner`


@dataclass
class FieldSchema:
    """One input column (reference schema.py:13-21; defaults identical)."""

    name: str
    feature_type: FeatureType
    vocabulary_size: int = 0
    embedding_dim: int = 8
    group: str = ""
    max_length: int = 1
    combiner: str = "mean"


@dataclass
class DatasetSchema:
    """Ordered set of fields plus the label column (reference schema.py:24-59)."""

    fields: Dict[str, FieldSchema] = _dc_field(default_factory=dict)
    label_field: str = "label"

    def _of_type(self, kind: FeatureType) -> List[FieldSchema]:
        return [spec for spec in self.fields.values() if spec.feature_type is kind]

    @property
    def sparse_fields(self) -> List[FieldSchema]:
        return self._of_type(FeatureType.SPARSE)

    @property
    def dense_fields(self) -> List[FieldSchema]:
        return self._of_type(FeatureType.DENSE)

    @property
    def sequence_fields(self) -> List[FieldSchema]:
        return self._of_type(FeatureType.SEQUENCE)

    @property
    def num_fields(self) -> int:
        return len(self.fields)

    @property
    def total_embedding_dim(self) -> int:
        # width of flat_embeddings = sum of raw per-field dims
        return sum(spec.embedding_dim for spec in self.fields.values())
