"""ctypes binding of libseg3d_hip.so (the C ABI declared in include/seg3d_hip.h).

There is no fallback: if the shared object is missing or a symbol does not resolve, loading
raises.  Build it with ``python __graft_entry__.py`` (hipcc --offload-arch=gfx950).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libseg3d_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "seg3d_hip.h")

OK, EINVAL, EWORKSPACE, ELAUNCH = 0, -1, -2, -3
REDUCE_SUM, REDUCE_MEAN, REDUCE_MAX = 0, 1, 2
ABI_VERSION = 40

_p, _i32, _i64, _sz, _f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); must match include/seg3d_hip.h (tests/test_boundary.py cross-checks the names)
SIGNATURES = {
    "seg3d_abi_version": (ctypes.c_int, []),
    "seg3d_last_error": (ctypes.c_char_p, []),
    "seg3d_grid_size": (ctypes.c_int, [_p, _p, _p]),
    "seg3d_voxelize_workspace_bytes": (_sz, [_i64]),
    "seg3d_voxelize_f32": (ctypes.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "seg3d_voxelize_f64": (ctypes.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "seg3d_voxelize_host_workspace_bytes": (_sz, [_i64]),
    "seg3d_voxelize_host_f32": (ctypes.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _sz]),
    "seg3d_voxelize_host_f64": (ctypes.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _sz]),
    "seg3d_cart2polar_f32": (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "seg3d_cart2polar_f64": (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "seg3d_group_index_workspace_bytes": (_sz, [_i64, _i64]),
    "seg3d_group_index": (ctypes.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    "seg3d_coord_hash_bytes": (_sz, [_i64]),
    "seg3d_coord_hash_build": (ctypes.c_int, [_p, _i64, _p, _p, _sz, _p]),
    "seg3d_rulebook_subm": (ctypes.c_int, [_p, _i64, _p, _p, _sz, _p, _p]),
    "seg3d_downsample_workspace_bytes": (_sz, [_i32, _p]),
    "seg3d_downsample_coords": (ctypes.c_int, [_p, _i64, _p, _i32, _p, _p, _i64, _p, _p, _sz, _p]),
    "seg3d_rulebook_strided": (ctypes.c_int, [_p, _i64, _i64, _p, _p, _sz, _p, _p, _p]),
    "seg3d_spconv_packed_bytes": (_sz, [_i32, _i32, _i32]),
    "seg3d_spconv_pack_weight": (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "seg3d_spconv_fwd": (ctypes.c_int, [_p, _p, _i64, _i64, _p, _i32, _p, _i32, _i32, _p, _p, _p]),
    "seg3d_spconv_fwd_act": (ctypes.c_int, [_p, _p, _i64, _i64, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    "seg3d_spconv_fwd_act_bf16": (ctypes.c_int, [_p, _i32, _p, _i64, _i64, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    "seg3d_spconv_presplit_bytes": (_sz, [_i64, _i32]),
    "seg3d_spconv_presplit": (ctypes.c_int, [_p, _i64, _i32, _p, _p]),
    "seg3d_spconv_fwd_presplit": (ctypes.c_int, [_p, _p, _i64, _i64, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    "seg3d_conv_plan_bytes": (_sz, [_i64]),
    "seg3d_conv_plan_workspace_bytes": (_sz, [_i64]),
    "seg3d_conv_plan_build": (ctypes.c_int, [_p, _p, _i64, _p, _p, _sz, _p]),
    "seg3d_spconv_tiled_supported": (_i32, [_i32, _i32]),
    "seg3d_spconv_fwd_tiled": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _p]),
    "seg3d_spconv_fwd_tiled_bf16": (ctypes.c_int, [_p, _i32, _p, _p, _i64, _i64, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _p]),
    "seg3d_debug_set_conv_nbt": (ctypes.c_int, [_i32]),
    "seg3d_debug_set_linear_stream": (ctypes.c_int, [_i32]),
    "seg3d_spconv_wgrad_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "seg3d_spconv_wgrad": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "seg3d_spconv_wgrad_partials": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _sz, _p, _p]),
    "seg3d_spconv_wgrad_partials_xbf16": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _sz, _p, _p]),
    "seg3d_linear_wgrad_partials_xbf16": (ctypes.c_int, [_p, _p, _i64, _i32, _i32, _i32, _p, ctypes.c_size_t, _p, _p]),
    "seg3d_linear_packed_bytes_f32": (ctypes.c_size_t, [_i32, _i32]),
    "seg3d_linear_pack_weight_f32": (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "seg3d_linear_fwd_f32": (ctypes.c_int, [_p, _i64, _p, _p, _i32, _i32, _p, _p]),
    "seg3d_debug_set_wgrad_lds": (ctypes.c_int, [_i32]),
    "seg3d_linear_packed_bytes_x6": (ctypes.c_size_t, [_i32, _i32]),
    "seg3d_linear_pack_weight_x6": (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "seg3d_linear_fwd_x6": (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "seg3d_pack_weights_batched": (ctypes.c_int, [_p, _i32, _i64, _p]),
    "seg3d_linear_wgrad_workspace_bytes": (ctypes.c_size_t, [_i64, _i32, _i32]),
    "seg3d_linear_wgrad": (ctypes.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_linear_wgrad_partials": (ctypes.c_int, [_p, _p, _i64, _i32, _i32, _i32, _p, ctypes.c_size_t, _p, _p]),
    "seg3d_reduce_partials": (ctypes.c_int, [_p, _i32, _i64, _i64, _p, _p, _p]),
    "seg3d_reduce_partials_batched": (ctypes.c_int, [_p, _i32, _i64, _p]),
    "seg3d_linear_packed_bytes": (_sz, [_i32, _i32, _i32]),
    "seg3d_linear_pack_weight": (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "seg3d_linear_fwd": (ctypes.c_int, [_p, _i64, _p, _p, _p, _i32, _i32, _p, _p]),
    "seg3d_linear_fwd_sum": (ctypes.c_int, [_p, _p, _i64, _p, _p, _i32, _i32, _p, _p]),
    "seg3d_linear_fwd_mul": (ctypes.c_int, [_p, _i64, _p, _p, _i32, _i32, _p, _p]),
    "seg3d_linear_layernorm_fwd": (ctypes.c_int, [_p, _i64, _p, _p, _p, _p, _p, ctypes.c_float, _i32, _i32, _p, _p]),
    "seg3d_window_partition_workspace_bytes": (_sz, [_i64, _i32, _p]),
    "seg3d_window_partition": (ctypes.c_int, [_p, _i64, _i32, _p, _p, _p, _i32, _p, _p, _p,
                                              _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "seg3d_pos_embed": (ctypes.c_int, [_p, _i64, _p, _p, _i32, _p, _p]),
    "seg3d_window_attn_supported": (ctypes.c_int, [_i32, _i32]),
    "seg3d_window_attn_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "seg3d_window_attn_fwd": (ctypes.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i32, _p, _i32,
                                             _i64, _i32, _i32, _i32, _p, _f, _f, _u64, _p, _p, _p, _sz, _p]),
    "seg3d_window_attn_bwd": (ctypes.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p,
                                             _i32, _i64, _i32, _i32, _i32, _p, _f, _f, _u64, _p, _p, _p, _i32, _i32, _i32,
                                             _p, _p, _sz, _p]),
    "seg3d_layernorm_fwd": (ctypes.c_int, [_p, _p, _p, _p, _p, _f, _i64, _i32, _p, _p, _p, _p]),
    "seg3d_layernorm_bwd_workspace_bytes": (ctypes.c_size_t, [_i64, _i32]),
    "seg3d_layernorm_bwd": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_layernorm_bwd_partials": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, ctypes.c_size_t, _p, _p]),
    "seg3d_batchnorm_workspace_bytes": (ctypes.c_size_t, [_i64, _i32]),
    "seg3d_colstats": (ctypes.c_int, [_p, _i64, _i32, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_batchnorm_stats": (ctypes.c_int, [_p, _i64, _i32, _f, _p, _p, _f, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_affine_act": (ctypes.c_int, [_p, _p, _p, _p, _i32, _i64, _i32, _p, _p]),
    "seg3d_gelu_fwd": (ctypes.c_int, [_p, _i64, _p, _p, _p]),
    "seg3d_batchnorm_bwd": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i64, _i32, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_batchnorm_bwd_reduce": (ctypes.c_int, [_p, _p, _p, _p, _p, _i32, _i64, _i32, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_batchnorm_bwd_apply": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i64, _i32, _p, _p, _p]),
    "seg3d_segment_reduce_fwd": (ctypes.c_int, [_p, _i32, _p, _p, _i64, _i32, _p, _p, _p]),
    "seg3d_segment_reduce_bwd": (ctypes.c_int, [_p, _i32, _p, _i64, _p, _p, _i64, _i32, _p, _p]),
    "seg3d_voxel_majority_labels": (ctypes.c_int, [_p, _p, _p, _i64, _i32, _p, _p]),
    "seg3d_cross_entropy_workspace_bytes": (ctypes.c_size_t, [_i64]),
    "seg3d_cross_entropy_fwd": (ctypes.c_int, [_p, _p, _i64, _i32, _i64, _f, _p, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_cross_entropy_bwd": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    "seg3d_lovasz_workspace_bytes": (ctypes.c_size_t, [_i64, _i32]),
    "seg3d_lovasz_softmax_fwd": (ctypes.c_int, [_p, _p, _i64, _i32, _i64, _i32, _p, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_lovasz_softmax_bwd": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p]),
    "seg3d_knn_level_workspace_bytes": (ctypes.c_size_t, [_i64]),
    "seg3d_knn_level_build": (ctypes.c_int, [_p, _i64, _p, _i32, _f, _p, _p, _p, _p, _p, _i64, _p, ctypes.c_size_t, _p]),
    "seg3d_parity_order": (ctypes.c_int, [_p, _i64, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_knn_query_order": (ctypes.c_int, [_p, _i64, _p, _i32, _f, _p, _p, ctypes.c_size_t, _p]),
    "seg3d_knn_grid_query": (ctypes.c_int, [_p, _i32, _p, _p, _i64, _p, _p, _i32, _i32, _p, _p, _p]),
    "seg3d_knn_query": (ctypes.c_int, [_p, _i64, _p, _i64, _p, _p, _i32, _i32, _p, _p, _p]),
    "seg3d_gather_rows": (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p]),
    "seg3d_class_context_fwd": (ctypes.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i64, _i32, _i32, _f, _p, _p, _p, _p, _p]),
    "seg3d_class_context_bwd": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i64, _i32, _i32, _f, _p, _p, _p, _p]),
    "seg3d_knn_attention_fwd": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _f, _p, _p, _p]),
    "seg3d_knn_attention_bwd": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _f, _p, _p, _p, _p, _p]),
}

_ERR = {EINVAL: "SEG3D_EINVAL (bad argument)", EWORKSPACE: "SEG3D_EWORKSPACE (workspace too small)",
        ELAUNCH: "SEG3D_ELAUNCH (HIP launch/runtime error)"}

_lib = None


class Seg3dError(RuntimeError):
    pass


def header_symbols():
    """Function names declared in include/seg3d_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(seg3d_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load the shared object and bind every entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Seg3dError(
            f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
            "Build it with `python __graft_entry__.py`.")
    # torch first: its bundled HIP runtime must be the one this library binds to (the streams and device pointers
    # handed over come from it); loading the library before torch would pull in a second runtime from /opt/rocm
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.seg3d_abi_version() != ABI_VERSION:
        raise Seg3dError(f"libseg3d_hip.so ABI {lib.seg3d_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning entry point; negative return codes raise Seg3dError."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != OK:
        detail = ""
        if rc == ELAUNCH:  # the runtime's own words (hipGetErrorString) for the failure behind SEG3D_ELAUNCH
            text = lib.seg3d_last_error()
            detail = f": {text.decode(errors='replace')}" if text else ""
        raise Seg3dError(f"{name} failed: {_ERR.get(rc, rc)}{detail}")


def query(name, *args):
    """Invoke a size_t-returning *_bytes query."""
    return int(getattr(load(), name)(*args))
