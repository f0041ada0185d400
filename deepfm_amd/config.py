"""Experiment configuration: the construction API of the models.

Mirror of the reference's ``deepfm/config.py:13-86`` dataclasses (same names and
defaults) — ``BaseCTRModel(schema, config)`` reads ``config.feature.*``,
``config.dnn.*``, ``config.cin.*`` and ``config.attention.*`` exactly like the
reference (base.py:32-34, xdeepfm.py:20-25, attention_deepfm.py:27-33).

The reference builds the nested dataclasses from YAML with the third-party
``dacite`` package (config.py:110).  That package is not a dependency here:
``load_config`` uses a small recursive builder over ``dataclasses.fields``.
"""

from __future__ import annotations

import ast
import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, List, Optional, Union, get_type_hints


@dataclass
class DataConfig:
    dataset_name: str = "movielens"
    data_dir: str = "data/ml-100k"
    split_strategy: str = "temporal"
    temporal_val_ratio: float = 0.1
    temporal_test_ratio: float = 0.1
    neg_sampling_alpha: float = 0.75
    min_interactions: int = 3
    label_threshold: float = 4.0
    num_neg_train: int = 4
    num_neg_eval: int = 999


@dataclass
class FeatureConfig:
    fm_embed_dim: int = 16
    embedding_l2_reg: float = 1e-5


@dataclass
class FMConfig:
    # declared by the reference (config.py:33-36) but read by no model
    use_first_order: bool = True
    use_second_order: bool = True


@dataclass
class DNNConfig:
    hidden_units: List[int] = field(default_factory=lambda: [256, 128, 64])
    activation: str = "relu"
    dropout: float = 0.1
    use_batch_norm: bool = True


@dataclass
class CINConfig:
    layer_sizes: List[int] = field(default_factory=lambda: [128, 128])
    split_half: bool = True


@dataclass
class AttentionConfig:
    num_heads: int = 4
    attention_dim: int = 64
    num_layers: int = 1
    use_residual: bool = True


@dataclass
class TrainingConfig:
    num_epochs: int = 50
    batch_size: int = 4096
    lr: float = 1e-3
    optimizer: str = "adam"
    scheduler: str = "reduce_on_plateau"
    early_stopping_patience: int = 5
    metric: str = "auc"
    gradient_clip_norm: float = 1.0
    ranking_ks: List[int] = field(default_factory=lambda: [1, 5, 10, 20])


@dataclass
class ExperimentConfig:
    model_name: str = "deepfm"
    seed: int = 42
    device: str = "auto"
    output_dir: str = "outputs"
    data: DataConfig = field(default_factory=DataConfig)
    feature: FeatureConfig = field(default_factory=FeatureConfig)
    fm: FMConfig = field(default_factory=FMConfig)
    dnn: DNNConfig = field(default_factory=DNNConfig)
    cin: CINConfig = field(default_factory=CINConfig)
    attention: AttentionConfig = field(default_factory=AttentionConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)


def _build(cls: type, raw: dict) -> Any:
    """Instantiate dataclass ``cls`` from a (possibly nested) plain dict."""
    hints = get_type_hints(cls)
    known = {f.name for f in dataclasses.fields(cls)}
    unknown = set(raw) - known
    if unknown:
        raise ValueError(f"unknown keys for {cls.__name__}: {sorted(unknown)}")
    kwargs = {}
    for name, value in raw.items():
        target = hints[name]
        if dataclasses.is_dataclass(target) and isinstance(value, dict):
            kwargs[name] = _build(target, value)
        else:
            kwargs[name] = value
    return cls(**kwargs)


def _parse_value(text: str) -> Any:
    """CLI override literal → python value (bool / int / float / list / str);
    same precedence as the reference's ``_parse_value`` (config.py:113-131)."""
    lowered = text.lower()
    if lowered in ("true", "false"):
        return lowered == "true"
    for cast in (int, float):
        try:
            return cast(text)
        except ValueError:
            continue
    if text.startswith("[") and text.endswith("]"):
        try:
            return ast.literal_eval(text)
        except (ValueError, SyntaxError):
            pass
    return text


def load_config(
    yaml_path: Union[str, Path], overrides: Optional[List[str]] = None
) -> ExperimentConfig:
    """YAML file + ``a.b=c`` overrides → ExperimentConfig (reference config.py:89-110)."""
    import yaml

    with open(yaml_path) as handle:
        raw = yaml.safe_load(handle) or {}
    for item in overrides or []:
        dotted, _, literal = item.partition("=")
        node = raw
        *parents, leaf = dotted.strip().split(".")
        for key in parents:
            node = node.setdefault(key, {})
        node[leaf] = _parse_value(literal.strip())
    return _build(ExperimentConfig, raw)
