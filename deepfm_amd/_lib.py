"""ctypes binding of the C ABI declared in ``include/deepfm_hip.h``.

The library is built in-tree by ``deepfm_amd/csrc/Makefile`` (``__graft_entry__.build()``)
into ``deepfm_amd/lib/libdeepfm_hip.so``.  There is no CPU fallback: if the library is
missing, or a tensor is not on a HIP device, the callers raise.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFM_LIB_PATH: load another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("DFM_LIB_PATH") or os.path.join(_HERE, "lib", "libdeepfm_hip.so")

# == DFM_ABI_VERSION of include/deepfm_hip.h at the time SIGNATURES / the ctypes structs below were written:
# bumped together with the header whenever a struct layout or an argument list changes, so that a stale .so
# (the library is untracked and DFM_LIB_PATH can point anywhere) is refused instead of fed shifted arguments
ABI_VERSION = 8

MAX_FIELDS = 64
MAX_RANKS = 64
ROWPLAN_CHUNK = 4096
SPARSE, DENSE, SEQUENCE = 0, 1, 2
COMBINER = {"mean": 0, "sum": 1, "max": 2}

c_float_p = C.c_void_p  # device pointers travel as integers
c_int_p = C.c_void_p


class Field(C.Structure):
    """struct dfm_field"""
    _fields_ = [("kind", C.c_int32), ("dim", C.c_int32), ("vocab", C.c_int32), ("max_len", C.c_int32),
                ("combiner", C.c_int32), ("flat_offset", C.c_int32), ("stride2", C.c_int32),
                ("stride1", C.c_int32),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("proj", C.c_void_p)]


class FieldGrad(C.Structure):
    """struct dfm_field_grad"""
    _fields_ = [("w2", C.c_void_p), ("b2", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("proj", C.c_void_p)]


class Table(C.Structure):
    """struct dfm_table"""
    _fields_ = [("w2", C.c_void_p), ("m2", C.c_void_p), ("v2", C.c_void_p),
                ("w1", C.c_void_p), ("m1", C.c_void_p), ("v1", C.c_void_p),
                ("stride2", C.c_int32), ("stride1", C.c_int32)]


class BnBwd(C.Structure):
    """struct dfm_bn_bwd"""
    _fields_ = [("z", C.c_void_p), ("mean_rstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("dy", C.c_void_p), ("g_gamma", C.c_void_p), ("g_beta", C.c_void_p),
                ("seed", C.c_void_p), ("workspace", C.c_void_p), ("p_drop", C.c_float), ("salt", C.c_int32)]


class FmBwd(C.Structure):
    """struct dfm_fm_bwd"""
    _fields_ = [("g_fm", C.c_void_p), ("fm_sum", C.c_void_p), ("e", C.c_void_p), ("addend", C.c_void_p),
                ("dim", C.c_int32)]


class SlabRef(C.Structure):
    """struct dfm_slab_ref"""
    _fields_ = [("workspace", C.c_void_p), ("g_w", C.c_void_p), ("batch", C.c_int64), ("out_features", C.c_int32),
                ("in_features", C.c_int32), ("splits", C.c_int32), ("reserved", C.c_int32)]


class SplitJob(C.Structure):
    """struct dfm_split_job"""
    _fields_ = [("src", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("planes_f", C.c_void_p),
                ("planes_s", C.c_void_p)]


class PartialJob(C.Structure):
    """struct dfm_partial_job"""
    _fields_ = [("kind", C.c_int32), ("blocks", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32),
                ("accumulate", C.c_int32), ("reserved", C.c_int32), ("partial", C.c_void_p), ("out_w", C.c_void_p),
                ("out_b", C.c_void_p), ("ldw", C.c_int64)]


class HeadTail(C.Structure):
    """struct dfm_head_tail"""
    _fields_ = [("g_w", C.c_void_p), ("g_b", C.c_void_p), ("loss", C.c_void_p), ("g_b2", C.c_void_p)]


# name -> (restype, argtypes); must list every symbol of include/deepfm_hip.h
_P, _I, _L, _F, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "dfm_abi_version": (_I, []),
    "dfm_last_error": (C.c_char_p, []),
    "dfm_device_info": (_I, [_P, _P, C.c_char_p, _I]),
    "dfm_debug_empty_launch": (_I, [_P]),
    "dfm_embedding_plan_create": (_I, [C.POINTER(Field), _I, _I, C.POINTER(_P)]),
    "dfm_embedding_plan_destroy": (_I, [_P]),
    "dfm_embedding_plan_is_uniform": (_I, [_P]),
    "dfm_embedding_workspace_bytes": (_SZ, [_P, _L]),
    "dfm_embedding_forward": (_I, [_P, C.POINTER(_P), _L, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dfm_embedding_forward_staged": (_I, [_P, C.POINTER(_P), C.POINTER(_P), _P, _P, _L, _P, _P, _P, _P, _P, _P]),
    "dfm_embedding_forward_staged_update": (_I, [_P, _P, _P, C.POINTER(_P), C.POINTER(_P), _P, _P, _L, _P, _P, _P, _P, _P]),
    "dfm_graph_last_node": (_I, [_P, C.POINTER(_P)]),
    "dfm_gather_timing_begin": (_I, [_I]),
    "dfm_gather_timing_end": (_I, [C.POINTER(C.c_float), _I, C.POINTER(_I)]),
    "dfm_gather_set_shape": (_I, [_I]),
    "dfm_cin_set_mode": (_I, [_I]),
    "dfm_cin_get_mode": (_I, []),
    "dfm_embedding_backward_dense": (_I, [_P, C.POINTER(_P), _L, _P, _P, _P, C.POINTER(FieldGrad), _P, _P]),
    "dfm_embedding_backward_dense_fields": (_I, [_P, C.POINTER(_P), _L, _P, _P, _P, C.POINTER(FieldGrad), _P]),
    "dfm_rowplan_build": (_I, [C.POINTER(_P), C.POINTER(C.c_int32), _I, _L, _P, _P, _P, _P, _P, _P, _I, _P]),
    "dfm_rowplan_build_update": (_I, [_P, _P, C.POINTER(_P), C.POINTER(C.c_int32), _I, _L, _P, _P, _P, _P, _P, _P, _I]),
    "dfm_rowgrad_build": (_I, [C.POINTER(C.c_int32), _I, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dfm_rowadam_num_partials": (_L, [_I, _I, _I]),
    "dfm_rowadam_merge": (_I, [C.POINTER(Table), _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P]),
    "dfm_rowadam_apply": (_I, [C.POINTER(Table), _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P]),
    "dfm_dense_num_partials": (_L, [_L]),
    "dfm_dense_grad_prepare": (_I, [_P, _P, _L, _L, _F, _P, _P]),
    "dfm_grad_norm_finalize": (_I, [_P, _L, _F, _P, _P, _P, _P, _P]),
    "dfm_dense_adam": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P, _I, _P]),
    "dfm_cin_output_dim": (_I, [C.POINTER(C.c_int32), _I, _I]),
    "dfm_cin_saved_bytes": (_SZ, [C.POINTER(C.c_int32), _I, _I, _L, _I, _I]),
    "dfm_cin_backward_workspace_bytes": (_SZ, [C.POINTER(C.c_int32), _I, _I, _L, _I, _I]),
    "dfm_cin_forward_workspace_bytes": (_SZ, [C.POINTER(C.c_int32), _I, _I, _I, _I]),
    "dfm_cin_forward": (_I, [_P, _L, _I, _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int32), _I, _I, _P, _P, _P, _P]),
    "dfm_cin_backward": (_I, [_P, _L, _I, _I, C.POINTER(_P), C.POINTER(C.c_int32), _I, _I, _P, _P, _P,
                              C.POINTER(_P), C.POINTER(_P), _P, _P]),
    "dfm_attention_forward": (_I, [_P, _L, _I, _I, _I, _I, _I, C.POINTER(_P), _P, _P]),
    "dfm_attention_backward_workspace_bytes": (_SZ, [_L, _I, _I]),
    "dfm_attention_backward": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, C.POINTER(_P), _P, C.POINTER(_P), _P, _P]),
    "dfm_bn_workspace_bytes": (_SZ, [_L, _I]),
    "dfm_bn_relu_dropout_forward": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _F, _F, _F, _P, _I, _P, _P, _P, _P]),
    "dfm_bn_relu_dropout_backward": (_I, [_P, _P, _P, _P, _P, _L, _I, _F, _P, _I, _P, _P, _P, _P, _P]),
    "dfm_bce_workspace_bytes": (_SZ, [_L]),
    "dfm_bce_with_logits": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "dfm_gemm_workspace_bytes": (_SZ, [_I, _I, _I]),
    "dfm_gemm_f32": (_I, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _P, _I, _P, _P]),
    "dfm_weight_grad_workspace_bytes": (_SZ, [_L, _I, _I]),
    "dfm_weight_grad_f32": (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _L, _P, _I, _P, _P]),
    "dfm_attention_core_supported": (_I, [_I, _I, _I]),
    "dfm_attention_core_forward": (_I, [_P, _L, _I, _I, _I, _P, _P]),
    "dfm_attention_core_backward": (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "dfm_attention_qkv_core_supported": (_I, [_I, _I, _I, _I]),
    "dfm_attention_qkv_core_forward": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "dfm_attention_qkv_core_backward": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "dfm_attention_block_supported": (_I, [_I, _I, _I, _I]),
    "dfm_attention_block_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _L, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P, _L,
                                         _P]),
    "dfm_attention_block_backward": (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _P]),
    "dfm_layernorm_workspace_bytes": (_SZ, [_L, _I]),
    "dfm_layernorm_forward": (_I, [_P, _P, _L, _I, _P, _P, _F, _P, _P, _L, _L, _P]),
    "dfm_layernorm_backward": (_I, [_P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _L, _L, _P]),
    "dfm_linear_bn_workspace_bytes": (_SZ, [_L, _I]),
    "dfm_linear_bn_forward": (_I, [_P, _L, _P, _P, _L, _I, _I, _P, _P, _P]),
    "dfm_bn_relu_dropout_apply": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P, _I, _P, _P]),
    "dfm_bn_bwd_workspace_bytes": (_SZ, [_L, _I]),
    "dfm_bn_backward_apply": (_I, [C.POINTER(BnBwd), _L, _I, C.POINTER(HeadTail), _P, _P]),
    "dfm_linear1_supported": (_I, [_I]),
    "dfm_linear1_forward": (_I, [_P, _L, _I, _P, _P, _P, _P]),
    "dfm_linear1_backward_splits": (_I, [_L]),
    "dfm_linear1_backward": (_I, [_P, _P, _L, _I, _P, _P, _P, _P]),
    "dfm_head_bce": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _P, C.POINTER(BnBwd), _P]),
    "dfm_head_bn_bce": (_I, [_P, _P, _P, _P, _P, _F, _F, _L, _I, _P, _P, _P, _P, _P, _P, _P, C.POINTER(BnBwd), _P]),
    "dfm_linear_backward_workspace_bytes": (_SZ, [_L, _I, _I]),
    "dfm_linear_backward": (_I, [_P, _L, _I, _P, _I, _P, _P, C.POINTER(BnBwd), C.POINTER(FmBwd), _I, _P, _P]),
    "dfm_linear_backward_finish": (_I, [C.POINTER(SlabRef), _I, _P]),
    "dfm_linear_backward_splits": (_I, [_L, _I, _I]),
    "dfm_step_embedding_backward": (_I, [_P, _I, C.POINTER(_P), C.POINTER(FieldGrad), C.POINTER(C.c_int32), _I, _I, _I,
                                         _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _P]),
    "dfm_step_prepare_num_partials": (_L, [_I, _I, _I, _L]),
    "dfm_step_match_bytes": (_SZ, [_I, _I]),
    "dfm_step_prepare": (_I, [C.POINTER(Table), _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _L, _L,
                              C.POINTER(SlabRef), _I, _P, _I, _L, _P, _L, _P, _P]),
    "dfm_step_apply": (_I, [C.POINTER(Table), _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _P, _P, _P,
                            _L, _I, _P]),
    "dfm_step_apply_plan": (_I, [C.POINTER(Table), _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _P, _P, _P,
                                 _L, _I, _P, _L, _P, _I, _L, _P, _P, _P, _P, _P, _P]),
    "dfm_step_apply_plan_update": (_I, [_P, _P, C.POINTER(Table), _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P,
                                        _P, _P, _P, _P, _L, _I, _P, _L, _P, _I, _L, _P, _P, _P, _P, _P]),
    "dfm_weight_grad_partial_blocks": (_I, [_L]),
    "dfm_weight_grad_partials_f32": (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P]),
    "dfm_weight_grad_partials_pair_f32": (_I, [_P, _L, _P, _L, _I, _I, _P, _P, _L, _P, _L, _I, _I, _P, _L, _P]),
    "dfm_layernorm_partial_blocks": (_I, [_L]),
    "dfm_partials_finish": (_I, [C.POINTER(PartialJob), _I, _P]),
    "dfm_tower_set_mode": (_I, [_I]),
    "dfm_tower_get_mode": (_I, []),
    "dfm_planes_bytes": (_SZ, [_L, _L]),
    "dfm_tower_x6_supported": (_I, [_L, _I, _I]),
    "dfm_split_planes": (_I, [C.POINTER(SplitJob), _I, _P]),
    "dfm_linear_bn_forward_x6": (_I, [_P, _L, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "dfm_bn_relu_dropout_apply_planes": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P, _I, _P, _P, _P,
                                              _P]),
    "dfm_bn_backward_apply_planes": (_I, [C.POINTER(BnBwd), _L, _I, C.POINTER(HeadTail), _P, _P, _P, _P]),
    "dfm_linear_backward_x6_splits": (_I, [_L, _I, _I]),
    "dfm_linear_backward_x6_workspace_bytes": (_SZ, [_L, _I, _I]),
    "dfm_linear_backward_x6": (_I, [_P, _P, _L, _I, _P, _P, _I, _P, _P, C.POINTER(BnBwd), C.POINTER(FmBwd), _P, _P]),
    "dfm_fm_forward": (_I, [_P, _L, _I, _I, _P, _P]),
    "dfm_fm_backward": (_I, [_P, _P, _L, _I, _I, _P, _P]),
    "dfm_copy_2d": (_I, [_P, _L, _P, _L, _L, _I, _P]),
    "dfm_stage_record": (_I, [_P, _P, _L, _P]),
    "dfm_stage_record_update": (_I, [_P, _P, _P, _P, _L]),
    "dfm_shard_gather": (_I, [C.POINTER(Table), C.POINTER(C.c_int32), _I, _I, _I, _L, _P, _P, _P, _P, _P]),
    "dfm_shard_pack_segment": (_L, [_L, _I, _I, _L]),
    "dfm_shard_pack": (_I, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, C.POINTER(C.c_int32), _I, _I, _I, _L,
                            _P, _P, _P, _L, C.POINTER(SlabRef), _I, _P, _P]),
    "dfm_shard_rowgrad": (_I, [_I, _I, _I, _L, _P, _L, _P, _P, _P, _P, _P, _P]),
    "dfm_sum_floats": (_I, [_P, _L, _P, _P]),
    "dfm_embedding_grad_combine": (_I, [_P, _L, _P, _P, _P, _P, _L, _I, _I, _P, _P]),
}

_lib: Optional[C.CDLL] = None


class HipLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C deepfm_amd/csrc`). deepfm_amd has no CPU or eager-PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.dfm_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {lib.dfm_abi_version()} != binding {ABI_VERSION} "
                              f"(stale {LIB_PATH}? rebuild with make -C deepfm_amd/csrc)")
    _lib = lib
    return lib


ERR_UNSUPPORTED = 3        # enum dfm_status: DFM_ERR_UNSUPPORTED


def check(rc: int) -> None:
    if rc != 0:
        msg = load().dfm_last_error().decode("utf-8", "replace")
        raise HipLibraryError(f"deepfm_hip error {rc}: {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


def stream_handle() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_device(t, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on {t.device}: deepfm_amd runs on an MI355X HIP device only "
            "(there is no CPU fallback; use the reference implementation on CPU).")
