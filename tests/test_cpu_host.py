"""CPU: the C-ABI library loads and exports every declared symbol; host-side mirror logic
(schema / config / module construction / state_dict keys / fail-loud behaviour).  No compute."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.helpers import cfg_of, fields_of, group, load, schema_from_fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from deepfm_amd import _lib
    header = open(os.path.join(ROOT, "include", "deepfm_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(dfm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in deepfm_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.dfm_abi_version() == _lib.ABI_VERSION == 8


def test_cin_layout_helpers_match_reference_bookkeeping():
    from deepfm_amd import _lib
    lib = _lib.load()
    for sizes, split, want in (([64, 64], 0, 128), ([64, 64], 1, 96), ([128, 128, 128], 1, 256), ([5, 7, 3], 1, 2 + 3 + 3)):
        arr = (ctypes.c_int32 * len(sizes))(*sizes)
        assert lib.dfm_cin_output_dim(arr, len(sizes), split) == want


def test_schema_and_config_mirror():
    from deepfm_amd.config import ExperimentConfig, _parse_value
    from deepfm_amd.data.schema import DatasetSchema, FeatureType, FieldSchema
    f = FieldSchema("a", FeatureType.SPARSE)
    assert (f.vocabulary_size, f.embedding_dim, f.group, f.max_length, f.combiner) == (0, 8, "", 1, "mean")
    s = DatasetSchema(fields={"a": FieldSchema("a", FeatureType.SPARSE, 10, 16),
                              "b": FieldSchema("b", FeatureType.DENSE, embedding_dim=4),
                              "c": FieldSchema("c", FeatureType.SEQUENCE, 5, 16, max_length=3)})
    assert s.num_fields == 3 and s.total_embedding_dim == 36          # tests/test_schema.py:70-73
    assert [x.name for x in s.sparse_fields] == ["a"] and [x.name for x in s.dense_fields] == ["b"]
    assert [x.name for x in s.sequence_fields] == ["c"]
    c = ExperimentConfig()
    assert c.feature.fm_embed_dim == 16 and c.cin.layer_sizes == [128, 128] and c.attention.attention_dim == 64
    assert c.training.batch_size == 4096 and c.feature.embedding_l2_reg == 1e-5
    assert _parse_value("true") is True and _parse_value("3") == 3 and _parse_value("[1,2]") == [1, 2]


def test_load_config_yaml_and_overrides(tmp_path):
    from deepfm_amd.config import load_config
    p = tmp_path / "c.yaml"
    p.write_text("model_name: xdeepfm\ncin:\n  layer_sizes: [64]\ntraining:\n  lr: 0.01\n")
    cfg = load_config(p, ["training.batch_size=2048", "cin.split_half=false"])
    assert cfg.model_name == "xdeepfm" and cfg.cin.layer_sizes == [64] and cfg.cin.split_half is False
    assert cfg.training.batch_size == 2048 and cfg.training.lr == 0.01
    p.write_text("nope: 1\n")
    with pytest.raises(ValueError):
        load_config(p)


@pytest.mark.parametrize("case", ["model_deepfm", "model_xdeepfm", "model_attention_deepfm", "model_deepfm_movielens"])
def test_models_expose_reference_state_dict_keys_and_shapes(case):
    from deepfm_amd.models import create_model
    from tests.test_gpu_models_step import _config
    g = load(case)
    c = cfg_of(g)
    model = create_model(c["kind"], schema_from_fields(fields_of(g)), _config(c))
    want = group(g, "param/")
    sd = model.state_dict()
    assert sorted(sd) == sorted(want)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(want[k].shape), k
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in want.items()})   # checkpoint round trip


def test_same_seed_same_initial_weights_and_padding_row_zero():
    from deepfm_amd.models.layers.embedding import FeatureEmbedding
    g = load("emb_layers_test_schema")
    schema = schema_from_fields(fields_of(g))
    torch.manual_seed(3)
    a = FeatureEmbedding(schema, 16)
    torch.manual_seed(3)
    b = FeatureEmbedding(schema, 16)
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    for name in ("u", "i", "g"):
        assert float(a.second_order_embeddings[name].weight[0].abs().sum()) == 0.0     # embedding.py:66-74


def test_rowsparse_mode_refuses_non_uniform_schemas():
    """The row-sparse gradient path (packed records, row plans, fused steps) is for uniform SPARSE / DENSE schemas; a
    schema with a SEQUENCE field, a projection or mixed embedding widths trains in dense mode."""
    from deepfm_amd.models.layers.embedding import FeatureEmbedding
    g = load("emb_layers_test_schema")                      # SPARSE + SEQUENCE + projections
    emb = FeatureEmbedding(schema_from_fields(fields_of(g)), 16)
    with pytest.raises(NotImplementedError):
        emb.set_grad_mode("rowsparse")
    with pytest.raises(ValueError):
        emb.set_grad_mode("sparse")
    assert emb.set_grad_mode("dense") is emb
    uniform = [dict(name=f"C{i}", type="sparse", vocab=50, dim=8, max_len=1, combiner="mean") for i in range(3)] + \
              [dict(name="I0", type="dense", vocab=0, dim=8, max_len=1, combiner="mean")]
    assert FeatureEmbedding(schema_from_fields(uniform), 8).set_grad_mode("rowsparse").grad_mode == "rowsparse"


def test_cpu_tensors_fail_loudly_no_fallback():
    from deepfm_amd.models.layers.attention import MultiHeadSelfAttention
    from deepfm_amd.models.layers.cin import CIN
    from deepfm_amd.models.layers.fm import FMInteraction
    x = torch.randn(2, 3, 8)
    for layer in (FMInteraction(), CIN(3, 8, [4]), MultiHeadSelfAttention(8, 2, 8)):
        with pytest.raises(RuntimeError, match="HIP device only"):
            layer(x)
    with pytest.raises(ValueError):
        MultiHeadSelfAttention(8, num_heads=3, attention_dim=8)
    from deepfm_amd.models.layers.dnn import DNN
    with pytest.raises(ValueError):
        DNN(4, [])
    with pytest.raises(ValueError):
        DNN(4, [2], activation="nope")


# ---- data-parallel host logic (training/sharded.py, training/exchange.py) ------------------------
@pytest.mark.parametrize("num_sparse,world", [(26, 1), (26, 2), (26, 3), (26, 8), (26, 26), (7, 4)])
def test_field_shards_cover_every_field_once(num_sparse, world):
    from deepfm_amd.training.sharded import FieldShards
    sh = FieldShards(num_sparse, world)
    assert sum(sh.count) == num_sparse and max(sh.count) - min(sh.count) <= 1
    assert sh.first == [sum(sh.count[:r]) for r in range(world)]
    owners = [sh.owner(s) for s in range(num_sparse)]
    assert [r * 1000 + j for r, j in owners] == sorted(r * 1000 + j for r, j in owners)     # contiguous, in rank order
    assert sorted({r for r, _ in owners}) == list(range(world))


def test_field_shards_reject_more_ranks_than_fields():
    from deepfm_amd.training.sharded import FieldShards
    with pytest.raises(ValueError):
        FieldShards(3, 4)
    with pytest.raises(ValueError):
        FieldShards(3, 0)


def test_stacked_view_only_inside_one_storage():
    """W_q | W_k | W_v as one stacked weight: a view when the three lie back to back in ONE storage (the
    optimizer's flat buffer), None otherwise — even for allocations that happen to be neighbours."""
    import torch
    from deepfm_amd.models.layers.attention import stacked_view
    flat = torch.arange(3 * 8 * 4, dtype=torch.float32)
    parts = [flat[i * 32:(i + 1) * 32].view(8, 4) for i in range(3)]
    v = stacked_view(parts)
    assert v is not None and v.shape == (24, 4) and v.data_ptr() == flat.data_ptr()
    assert torch.equal(v, flat.view(24, 4))
    assert stacked_view([parts[0], parts[2]]) is None                       # a gap
    assert stacked_view([parts[0].clone(), parts[1].clone()]) is None       # separate storages
    biases = [flat[i * 8:(i + 1) * 8] for i in range(3)]
    assert stacked_view(biases).shape == (24,)
    assert stacked_view([parts[0], parts[1].t().contiguous().t()]) is None  # not contiguous


def test_exchange_without_a_process_group_is_a_copy():
    import torch
    from deepfm_amd.training import exchange
    src = torch.arange(10, dtype=torch.float32)
    dst = torch.zeros(10)
    exchange.all_to_all(dst, src, [10], [10])
    assert torch.equal(dst, src)
    out = torch.zeros(4)
    exchange.all_gather_flat(out, torch.tensor([1.0, 2.0, 3.0, 4.0]))
    assert out.tolist() == [1.0, 2.0, 3.0, 4.0]
    with pytest.raises(RuntimeError):
        exchange.all_to_all(dst, src, [5, 5], [5, 5])                       # several peers need a process group
