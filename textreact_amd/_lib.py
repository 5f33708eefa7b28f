"""ctypes binding of libtrxknn.so (include/trx_knn.h).  No fallback: if the HIP library is missing
or there is no GPU, calls raise -- the product never computes on the CPU."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_CSRC, os.environ.get("TRX_LIB", "libtrxknn.so"))  # TRX_LIB: diagnostic builds

METRIC_IP, METRIC_L2 = 0, 1
TIES_BY_ID, TIES_FAISS = 0, 1
DTYPE_F32, DTYPE_BF16, DTYPE_I8 = 0, 1, 2
DTYPE_I64, DTYPE_I32, DTYPE_I16, DTYPE_U8, DTYPE_F64 = 3, 4, 5, 6, 7      # host entry points only
MAX_K, FAST_MAX_K = 2048, 24

# every symbol include/trx_knn.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "trx_index_create", "trx_index_add", "trx_index_add_device", "trx_index_ntotal", "trx_index_dim",
    "trx_index_reset", "trx_index_destroy", "trx_index_search", "trx_index_search_device",
    "trx_index_search_device_s64", "trx_index_search_device_begin", "trx_index_search_finish",
    "trx_merge_topk_device", "trx_index_last_stats", "trx_search_stats_size",
    "trx_index_set_timing", "trx_last_error", "trx_version",
    "trx_merge_topk_device_s64", "trx_index_set_tie_rule", "trx_faiss_tie_order_device",
    "trx_host_convert", "trx_host_threads",
]


class SearchStats(ctypes.Structure):
    _fields_ = [("nq", ctypes.c_int64), ("n_uncertified", ctypes.c_int64), ("k_split", ctypes.c_int32),
                ("n_splits", ctypes.c_int32), ("exact_class", ctypes.c_int32), ("scan_launches", ctypes.c_int32),
                ("scan_ms", ctypes.c_float), ("total_ms", ctypes.c_float), ("late_fallback", ctypes.c_int32),
                ("n_rescored", ctypes.c_int32), ("n_rescanned", ctypes.c_int32), ("int8_scan", ctypes.c_int32)]


class TrxError(RuntimeError):
    pass


_lib = None


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h", ".cpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "trx_knn.h"))
    newest = max(os.path.getmtime(s) for s in srcs)
    nn = os.path.join(_CSRC, "libtrxnn.so")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest or not os.path.exists(nn) \
            or os.path.getmtime(nn) < newest:
        subprocess.check_call(["make", "-C", _CSRC, "-j4"] + (["-B"] if force else []))
    return _SO


def lib():
    """Load libtrxknn.so (building it if the toolchain is here and it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        try:
            build()
        except Exception as e:  # no toolchain on this box
            raise TrxError("libtrxknn.so is missing and could not be built: %s" % e)
    try:
        import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
    except Exception:
        pass
    L = ctypes.CDLL(_SO)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    L.trx_index_create.argtypes = [i32, i32, i32, ctypes.POINTER(vp)]
    L.trx_index_add.argtypes = [vp, vp, i64, i32]
    L.trx_index_add_device.argtypes = [vp, vp, i64, i32, vp]
    L.trx_index_ntotal.argtypes = [vp]; L.trx_index_ntotal.restype = i64
    L.trx_index_dim.argtypes = [vp]
    L.trx_index_reset.argtypes = [vp]
    L.trx_index_destroy.argtypes = [vp]; L.trx_index_destroy.restype = None
    L.trx_index_search.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.trx_index_search_device.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp]
    L.trx_index_search_device_s64.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp]
    L.trx_index_search_device_begin.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp]
    L.trx_index_search_finish.argtypes = [vp]
    L.trx_merge_topk_device.argtypes = [i32, i32, i64, i32, vp, vp, vp, vp, vp]
    L.trx_merge_topk_device_s64.argtypes = [i32, i32, i64, i32, vp, vp, vp, vp, vp, vp]
    L.trx_index_set_tie_rule.argtypes = [vp, i32]
    L.trx_faiss_tie_order_device.argtypes = [i64, i32, i32, vp, vp, vp, vp, vp]
    L.trx_host_convert.argtypes = [vp, i32, i64, vp, i32]
    L.trx_index_last_stats.argtypes = [vp, ctypes.POINTER(SearchStats)]
    L.trx_index_set_timing.argtypes = [vp, i32]
    L.trx_last_error.restype = ctypes.c_char_p
    L.trx_version.restype = ctypes.c_char_p
    if L.trx_search_stats_size() != ctypes.sizeof(SearchStats):
        raise TrxError("libtrxknn.so was built with another trx_search_stats (%d bytes, this binding has %d): rebuild it"
                       % (L.trx_search_stats_size(), ctypes.sizeof(SearchStats)))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().trx_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise AssertionError(msg)  # FAISS raises AssertionError on shape / argument errors
        raise TrxError("trxknn error %d: %s" % (rc, msg))
