"""ctypes binding of the C ABI declared in include/fbk.h (libfbk.so, built by hipcc).

This module is plumbing: it declares argtypes/restypes for every exported symbol and
turns negative status codes into exceptions.  There is no CPU fallback anywhere in this
package: if libfbk.so is missing, or no gfx950 device is visible, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FBK_LIB_PATH") or os.path.join(_HERE, "csrc", "libfbk.so")  # override: A/B builds of the same ABI

FBK_OK = 0
FBK_E_INVALID, FBK_E_NODEVICE, FBK_E_HIP, FBK_E_NOMEM, FBK_E_CAPACITY, FBK_E_NOTFOUND = -1, -2, -3, -4, -5, -6
TYPE_NIL, TYPE_ARRAY, TYPE_BITMAP, TYPE_RUN = 0, 1, 2, 3
OP_AND, OP_OR, OP_XOR, OP_ANDNOT = 0, 1, 2, 3
SETOP_KEEP_BITMAP, SETOP_OPTIMIZE = 0, 1
QUERY_ACCUMULATE = 1


class FbkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"fbk error {code}: {msg}")
        self.code = code


class ContainerDesc(C.Structure):
    """Mirror of fbk_container_desc (include/fbk.h)."""

    _fields_ = [
        ("key", C.c_uint64),
        ("off", C.c_uint64),
        ("row", C.c_uint32),
        ("len", C.c_uint32),
        ("n", C.c_int32),
        ("type", C.c_uint8),
        ("pad", C.c_uint8 * 3),
    ]


class MatrixArgs(C.Structure):
    """Mirror of fbk_matrix_args (include/fbk.h): one group member's count-matrix arguments."""

    _fields_ = [
        ("a", C.c_void_p),
        ("rows_a", C.c_void_p),
        ("b", C.c_void_p),
        ("rows_b", C.c_void_p),
        ("filter", C.c_void_p),
        ("rows_f", C.c_void_p),
        ("n_shards", C.c_uint32),
        ("pad", C.c_uint32),
    ]


class BsiArgs(C.Structure):
    """Mirror of fbk_bsi_args (include/fbk.h): one group member's arguments of a BSI Sum."""

    _fields_ = [("batch", C.c_void_p), ("base_rows", C.c_void_p), ("filter", C.c_void_p), ("rows_f", C.c_void_p), ("n_shards", C.c_uint32), ("pad", C.c_uint32)]


class TopnArgs(C.Structure):
    """Mirror of fbk_topn_args (include/fbk.h): one group member's arguments of a TopN."""

    _fields_ = [("a", C.c_void_p), ("rows_a", C.c_void_p), ("filter", C.c_void_p), ("rows_f", C.c_void_p), ("n_shards", C.c_uint32), ("pad", C.c_uint32)]


REDUCE_HOST, REDUCE_PEER, REDUCE_RCCL = 0, 1, 2

_vp, _u32p, _u64p, _i32p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
_vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  Must list every symbol include/fbk.h declares
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    "fbk_abi_version": (C.c_int32, []),
    "fbk_last_error": (C.c_char_p, [_vp]),
    "fbk_last_error_r": (C.c_int32, [_vp, C.c_char_p, C.c_uint64, _i32p]),
    "fbk_ctx_fork": (C.c_int32, [_vp, _vpp]),
    "fbk_set_option": (C.c_int32, [_vp, C.c_char_p, C.c_int64]),
    "fbk_get_option": (C.c_int32, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "fbk_device_count": (C.c_int32, [_i32p]),
    "fbk_open": (C.c_int32, [C.c_int32, C.c_uint32, _vpp]),
    "fbk_close": (C.c_int32, [_vp]),
    "fbk_set_stream": (C.c_int32, [_vp, _vp]),
    "fbk_synchronize": (C.c_int32, [_vp]),
    "fbk_batch_upload": (C.c_int32, [_vp, C.POINTER(ContainerDesc), C.c_uint64, C.c_uint32, _vp, C.c_uint64, _vpp]),
    "fbk_batch_upload_dense": (C.c_int32, [_vp, _vp, C.c_uint32, _vpp]),
    "fbk_batch_upload_roaring": (C.c_int32, [_vp, _vp, C.c_uint64, _vpp, _vp, C.c_uint32, _u32p]),
    "fbk_rbf_find_root": (C.c_int32, [_vp, C.c_uint64, C.c_char_p, _u32p]),
    "fbk_batch_upload_rbf": (C.c_int32, [_vp, _vp, C.c_uint64, C.c_uint32, _vpp, _vp, C.c_uint32, _u32p]),
    "fbk_batch_roaring_size": (C.c_int32, [_vp, _vp, _u64p]),
    "fbk_batch_download_roaring": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, _u64p]),
    "fbk_batch_free": (C.c_int32, [_vp, _vp]),
    "fbk_batch_info": (C.c_int32, [_vp, _vp, _u32p, _u64p, _u64p]),
    "fbk_batch_download": (C.c_int32, [_vp, _vp, C.POINTER(ContainerDesc), C.c_uint64, _vp, C.c_uint64]),
    "fbk_cache_put": (C.c_int32, [_vp, C.c_char_p, C.c_uint64, _vp, _vp, C.c_uint32]),
    "fbk_cache_get": (C.c_int32, [_vp, C.c_char_p, C.c_uint64, _vpp, _vpp, _u32p]),
    "fbk_cache_release": (C.c_int32, [_vp, _vp]),
    "fbk_cache_invalidate": (C.c_int32, [_vp, C.c_char_p, _u32p]),
    "fbk_cache_configure": (C.c_int32, [_vp, C.c_uint64]),
    "fbk_cache_stats": (C.c_int32, [_vp, _u64p, _u64p, _u64p, _u64p, _u64p]),
    "fbk_count": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, _vp]),
    "fbk_count_range": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp]),
    "fbk_intersection_count": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "fbk_setop": (C.c_int32, [_vp, C.c_int32, _vp, _vp, _vp, _vp, C.c_uint64, C.c_uint32, _vpp, _vp]),
    "fbk_plan_create": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vpp]),
    "fbk_plan_free": (C.c_int32, [_vp, _vp]),
    "fbk_plan_intersection_count": (C.c_int32, [_vp, _vp]),
    "fbk_plan_intersection_count_total": (C.c_int32, [_vp, _vp, _vp]),
    "fbk_plan_intersection_count_accumulate": (C.c_int32, [_vp, _vp, _vp]),
    "fbk_plan_setop": (C.c_int32, [_vp, _vp, C.c_int32, C.c_uint32]),
    "fbk_plan_total": (C.c_int32, [_vp, _vp, _vp]),
    "fbk_plan_read": (C.c_int32, [_vp, _vp, _vp, _vp]),
    "fbk_plan_output": (C.c_int32, [_vp, _vp, _vpp]),
    "fbk_plan_detach_output": (C.c_int32, [_vp, _vp, _vpp]),
    "fbk_union_n": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vpp, _vp]),
    "fbk_union_n_intersection_count": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, _vp, _vp, _vp]),
    "fbk_fold_n": (C.c_int32, [_vp, C.c_int32, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vpp, _vp]),
    "fbk_fold_n_intersection_count": (C.c_int32, [_vp, C.c_int32, _vp, _vp, C.c_uint64, C.c_uint32, _vp, _vp, _vp]),
    "fbk_count_matrix": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, _vp, _vp]),
    "fbk_bsi_sum": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "fbk_bsi_min": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "fbk_bsi_max": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "fbk_bsi_distinct": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, C.c_uint64, _u64p]),
    "fbk_topk": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, C.c_uint32, _vp]),
    "fbk_topn": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _vp, _vp, C.c_uint32, _vp]),
    "fbk_batch_compact": (C.c_int32, [_vp, _vp, _vp]),
    "fbk_batch_memory": (C.c_int32, [_vp, _vp, _vp, _vp, _vp]),
    "fbk_topn_partials": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _vp, _vp]),
    "fbk_topk_bsi": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, _vpp, _u32p]),
    "fbk_flip": (C.c_int32, [_vp, _vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _vpp, _vp]),
    "fbk_rows": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp, C.c_uint64, _vp]),
    "fbk_shift": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_uint64, C.c_uint32, _vpp, _vp]),
    "fbk_bsi_add": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint64, C.c_uint32, _vpp]),
    "fbk_bsi_range": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_int32, C.c_uint32, C.c_int64, C.c_uint32, _vpp, _vp]),
    "fbk_bsi_between_sum_plan": (C.c_int32, [C.c_uint32, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "fbk_bsi_range_between_sum": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp]),
    "fbk_bsi_range_sum_plan": (C.c_int32, [C.c_int32, C.c_uint32, C.c_int64, _vp, _vp, _vp, _vp]),
    "fbk_bsi_range_sum": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_int32, C.c_uint32, C.c_int64, _vp, _vp, _vp, _vp]),
    "fbk_bsi_range_between": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.c_uint32, _vpp, _vp]),
    "fbk_query_count_matrix": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, _vpp]),
    "fbk_query_fold_intersection_count": (C.c_int32, [_vp, C.c_int32, _vp, _vp, C.c_uint64, C.c_uint32, _vp, _vp, _vpp]),
    "fbk_query_bsi_sum": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_int64, _vp, _vp, _vpp]),
    "fbk_query_bsi_range": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, C.c_int32, C.c_uint32, C.c_int64, _vpp]),
    "fbk_query_fold": (C.c_int32, [_vp, C.c_int32, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vpp]),
    "fbk_query_topn": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _vpp]),
    "fbk_query_output": (C.c_int32, [_vp, _vp, _vpp]),
    "fbk_query_run": (C.c_int32, [_vp, _vp, _vp, C.c_uint32]),
    "fbk_query_result": (C.c_int32, [_vp, _vp, _vpp, _u64p]),
    "fbk_query_read": (C.c_int32, [_vp, _vp, _vp, _vp]),
    "fbk_query_free": (C.c_int32, [_vp, _vp]),
    "fbk_group_open": (C.c_int32, [_i32p, C.c_uint32, C.c_uint32, _vpp]),
    "fbk_group_close": (C.c_int32, [_vp]),
    "fbk_group_size": (C.c_int32, [_vp, _u32p]),
    "fbk_group_member": (C.c_int32, [_vp, C.c_uint32, _vpp]),
    "fbk_group_set_reduce": (C.c_int32, [_vp, C.c_int32]),
    "fbk_group_plan_intersection_count_total": (C.c_int32, [_vp, _vpp, _u64p]),
    "fbk_group_count_matrix": (C.c_int32, [_vp, C.POINTER(MatrixArgs), C.c_uint32, C.c_uint32, _vp]),
    "fbk_group_reduce_u64": (C.c_int32, [_vp, _vpp, C.c_uint64, _vp]),
    "fbk_group_bsi_sum": (C.c_int32, [_vp, C.POINTER(BsiArgs), C.c_uint32, C.POINTER(C.c_int64), _u64p]),
    "fbk_group_topn": (C.c_int32, [_vp, C.POINTER(TopnArgs), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _vp, _vp, C.c_uint32, _u32p]),
    "fbk_group_last_error_r": (C.c_int32, [_vp, C.c_char_p, C.c_uint64, _i32p]),
    "fbk_comm_unique_id": (C.c_int32, [_vp]),
    "fbk_comm_init": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32]),
    "fbk_comm_all_reduce_u64": (C.c_int32, [_vp, _vp, C.c_uint64]),
    "fbk_comm_fence": (C.c_int32, [_vp]),
    "fbk_comm_close": (C.c_int32, [_vp]),
}
COMM_ID_BYTES = 128

BSI_EQ, BSI_NEQ, BSI_LT, BSI_LTE, BSI_GT, BSI_GTE = 1, 2, 3, 4, 5, 6
BSI_OPS = {"EQ": BSI_EQ, "NEQ": BSI_NEQ, "LT": BSI_LT, "LTE": BSI_LTE, "GT": BSI_GT, "GTE": BSI_GTE}

_lib = None


def load() -> C.CDLL:
    """Load libfbk.so (no GPU needed just to load it).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7
        # (same SONAME as /opt/rocm's).  If libfbk.so pulled in /opt/rocm's copy first and
        # torch were imported later, the process would hold two HSA runtimes and the
        # second one finds no GPU.  Importing torch first makes libfbk.so bind to the
        # already-loaded runtime.  (A Go host has no torch: /opt/rocm's runtime is used.)
        if "torch" not in sys.modules and os.environ.get("FBK_STANDALONE_HIP") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != FBK_OK:
        msg = load().fbk_last_error(None)
        raise FbkError(rc, msg.decode() if msg else "?")
