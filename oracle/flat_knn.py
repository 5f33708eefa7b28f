"""ctypes front-end of the CPU oracle (oracle/flat_knn_ref.c) plus a NumPy cross-check.

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py; never from textreact_amd/.  PARITY UNPINNED: see the header of flat_knn_ref.c.

Reference call site restated: retrieve/retrieve_faiss.py:62-74 (index_and_search) and the
faiss.IndexFlat{IP,L2}.search convention of section 8b of SURVEY.md:
``search(x float32[Q,d], k) -> (D float32[Q,k], I int64[Q,k])``, best first, I = -1 and
D = +-FLT_MAX where fewer than k vectors exist.
"""
import ctypes
import os
import subprocess

import numpy as np

METRIC_IP, METRIC_L2 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile libtrxoracle.so next to the sources (gcc, a second or two)."""
    so = os.path.join(_HERE, "libtrxoracle.so")
    src = os.path.join(_HERE, "flat_knn_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtrxoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i32, i64 = ctypes.c_int, ctypes.c_int64
        L.trxo_knn_faiss.argtypes = [i32, _f32p, i64, _f32p, i64, i32, i32, _f32p, _i64p]
        L.trxo_knn_faiss.restype = i32
        L.trxo_knn_canonical.argtypes = [i32, _f32p, i64, _f32p, i64, i32, i32, _f32p, _i64p]
        L.trxo_knn_canonical.restype = i32
        L.trxo_scores_at.argtypes = [i32, _f32p, i64, _f32p, i32, i32, _i64p, _f64p]
        L.trxo_scores_at.restype = None
        L.trxo_heap_begin.argtypes = [i32, i32, i64, _f32p, _i64p]
        L.trxo_heap_end.argtypes = [i32, i32, i64, _f32p, _i64p]
        L.trxo_heap_add_block.argtypes = [i32, i32, i64, i64, i64, i64, _f32p, _f32p, _i64p]
        L.trxo_l2_from_ip_block.argtypes = [i64, i64, i64, i64, _f32p, _f32p, _f32p]
        L.trxo_norms_f32.argtypes = [_f32p, i64, i32, _f32p]
        L.trxo_merge_lists.argtypes = [i32, i32, i64, i32, _f32p, _i64p, _f32p, _i64p]
        L.trxo_set_threads.argtypes = [i32]
        for fn in (L.trxo_heap_begin, L.trxo_heap_end, L.trxo_heap_add_block,
                   L.trxo_l2_from_ip_block, L.trxo_norms_f32, L.trxo_merge_lists, L.trxo_set_threads):
            fn.restype = None
        _LIB = L
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _check(x, y):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    if x.ndim != 2 or y.ndim != 2 or x.shape[1] != y.shape[1]:
        raise AssertionError("dimension mismatch: queries %r vs database %r" % (x.shape, y.shape))
    return x, y


def _run(fn, metric, x, y, k):
    x, y = _check(x, y)
    nq, d = x.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    rc = fn(metric, x.ctypes.data_as(_f32p), nq, y.ctypes.data_as(_f32p), y.shape[0], d, k,
            D.ctypes.data_as(_f32p), I.ctypes.data_as(_i64p))
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return D, I


def knn_faiss(metric, x, y, k):
    """Literal restatement of IndexFlat{IP,L2}.search (fp32 blocks + strict-admission heap)."""
    return _run(lib().trxo_knn_faiss, metric, x, y, k)


def knn_canonical(metric, x, y, k):
    """The order-independent definition (fp64 fma chain, total order) the HIP path must match."""
    return _run(lib().trxo_knn_canonical, metric, x, y, k)


def scores_at(metric, x, y, I):
    """Canonical fp64 scores of the (query, id) pairs in I (NaN where id < 0)."""
    x, y = _check(x, y)
    I = np.ascontiguousarray(I, dtype=np.int64)
    out = np.empty(I.shape, dtype=np.float64)
    lib().trxo_scores_at(metric, x.ctypes.data_as(_f32p), x.shape[0], y.ctypes.data_as(_f32p),
                         x.shape[1], I.shape[1], I.ctypes.data_as(_i64p), out.ctypes.data_as(_f64p))
    return out


def merge_lists(metric, D_lists, I_lists):
    """Merge [nlists, nq, k] sorted lists with global ids into one [nq, k] list (canonical order)."""
    Dl = np.ascontiguousarray(D_lists, dtype=np.float32)
    Il = np.ascontiguousarray(I_lists, dtype=np.int64)
    nl, nq, k = Dl.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().trxo_merge_lists(metric, nl, nq, k, Dl.ctypes.data_as(_f32p), Il.ctypes.data_as(_i64p),
                           D.ctypes.data_as(_f32p), I.ctypes.data_as(_i64p))
    return D, I


def knn_faiss_blas(metric, x, y, k, bs_x=4096, bs_y=1024):
    """FAISS's own structure with the host BLAS doing the sgemm: 4096 x 1024 blocks of
    x @ y.T through numpy (MKL/OpenBLAS), the C heap handler on each block.  This is the
    cpu_baseline ("port") leg of bench.py -- same algorithm, fastest honest CPU form here.
    [faiss/utils/distances.cpp: exhaustive_inner_product_blas / exhaustive_L2sqr_blas]"""
    x, y = _check(x, y)
    L = lib()
    nq, d = x.shape
    nb = y.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    Dp, Ip = D.ctypes.data_as(_f32p), I.ctypes.data_as(_i64p)
    L.trxo_heap_begin(metric, k, nq, Dp, Ip)
    if nq and nb:
        if metric == METRIC_L2:
            xn = np.empty(nq, dtype=np.float32)
            yn = np.empty(nb, dtype=np.float32)
            L.trxo_norms_f32(x.ctypes.data_as(_f32p), nq, d, xn.ctypes.data_as(_f32p))
            L.trxo_norms_f32(y.ctypes.data_as(_f32p), nb, d, yn.ctypes.data_as(_f32p))
        for i0 in range(0, nq, bs_x):
            i1 = min(i0 + bs_x, nq)
            for j0 in range(0, nb, bs_y):
                j1 = min(j0 + bs_y, nb)
                blk = np.ascontiguousarray(x[i0:i1] @ y[j0:j1].T)
                bp = blk.ctypes.data_as(_f32p)
                if metric == METRIC_L2:
                    L.trxo_l2_from_ip_block(i0, i1, j0, j1, xn.ctypes.data_as(_f32p),
                                            yn.ctypes.data_as(_f32p), bp)
                L.trxo_heap_add_block(metric, k, i0, i1, j0, j1, bp, Dp, Ip)
    L.trxo_heap_end(metric, k, nq, Dp, Ip)
    return D, I


def knn_faiss_blas_mt(metric, x, y, k, workers, bs_x=4096, bs_y=1024, gemm=None):
    """knn_faiss_blas spread over the host's cores the way that scales on a many-core box: `workers` host threads, each
    taking every workers-th 1024-row corpus block -- the same fp32 sgemm of a 4096 x 1024 block (ONE BLAS thread per call)
    and the same strict-admission heap handler on it, into the worker's own heaps -- and one merge of the workers' lists at
    the end.  FAISS itself threads INSIDE the sgemm and over the queries of the handler; a 6.4 GFLOP block split over
    hundreds of BLAS threads is mostly synchronisation (bench.py records that rate too), so this form is the stronger CPU
    baseline.  Same arithmetic per block; on tie-free data the same result as knn_faiss_blas (an inner-product heap's order
    among EXACT ties depends on arrival order, which differs).
    gemm(a [m, d], b [n, d]) -> a @ b.T as a C-contiguous fp32 array: the sgemm of one block, called from `workers` threads at
    once (default: numpy's BLAS under threadpool_limits(1); OpenBLAS builds with a 64-thread table complain beyond 64 callers
    -- bench.py passes torch's MKL sgemm instead)."""
    from concurrent.futures import ThreadPoolExecutor
    from threadpoolctl import threadpool_limits
    x, y = _check(x, y)
    L = lib()
    nq, d = x.shape
    nb = y.shape[0]
    nblocks = (nb + bs_y - 1) // bs_y
    workers = max(1, min(int(workers), nblocks))
    if metric == METRIC_L2:
        xn = np.empty(nq, dtype=np.float32); yn = np.empty(nb, dtype=np.float32)
        L.trxo_norms_f32(x.ctypes.data_as(_f32p), nq, d, xn.ctypes.data_as(_f32p))
        L.trxo_norms_f32(y.ctypes.data_as(_f32p), nb, d, yn.ctypes.data_as(_f32p))
    Dw = np.empty((workers, nq, k), dtype=np.float32)
    Iw = np.empty((workers, nq, k), dtype=np.int64)

    def work(w):
        L.trxo_set_threads(1)           # this thread's OpenMP team: the handler runs inline
        Dp, Ip = Dw[w].ctypes.data_as(_f32p), Iw[w].ctypes.data_as(_i64p)
        L.trxo_heap_begin(metric, k, nq, Dp, Ip)
        for i0 in range(0, nq, bs_x):
            i1 = min(i0 + bs_x, nq)
            for b in range(w, nblocks, workers):
                j0, j1 = b * bs_y, min((b + 1) * bs_y, nb)
                blk = np.ascontiguousarray(x[i0:i1] @ y[j0:j1].T) if gemm is None else gemm(x[i0:i1], y[j0:j1])
                bp = blk.ctypes.data_as(_f32p)
                if metric == METRIC_L2:
                    L.trxo_l2_from_ip_block(i0, i1, j0, j1, xn.ctypes.data_as(_f32p), yn.ctypes.data_as(_f32p), bp)
                L.trxo_heap_add_block(metric, k, i0, i1, j0, j1, bp, Dp, Ip)
        L.trxo_heap_end(metric, k, nq, Dp, Ip)

    with threadpool_limits(limits=1, user_api="blas"):
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(work, range(workers)))
    return merge_lists(metric, Dw, Iw)


def knn_numpy(metric, x, y, k):
    """Independent NumPy statement of the canonical rule for SMALL inputs: fp64 scores by an
    explicit k-ordered loop (no BLAS, so the summation order is the defined one), then a stable
    lexsort on (score, id).  Cross-checks the C code in tests/test_oracle.py."""
    x, y = _check(x, y)
    nq, d = x.shape
    nb = y.shape[0]
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    S = np.zeros((nq, nb), dtype=np.float64)
    for c in range(d):  # products/differences of fp32 values: a*b is exact in fp64, so a*b + s
        if metric == METRIC_L2:  # rounds once, like fma; (x-y)^2 is NOT exact -> use math.fma-free
            t = xd[:, c, None] - yd[None, :, c]
            S = _fma(t, t, S)
        else:
            S = xd[:, c, None] * yd[None, :, c] + S
    D = np.full((nq, k), np.float32(np.finfo(np.float32).max if metric == METRIC_L2
                                    else -np.finfo(np.float32).max), dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    ids = np.arange(nb)
    for i in range(nq):
        key = S[i] if metric == METRIC_L2 else -S[i]
        order = np.lexsort((ids, key))[:k]
        D[i, :len(order)] = S[i, order].astype(np.float32)
        I[i, :len(order)] = order
    return D, I


def _fma(a, b, c):
    """Correctly rounded a*b+c for float64 arrays via error-free transformation (Dekker/Veltkamp
    split); exact enough to reproduce a hardware fma except in astronomically rare double-rounding
    cases, which the small test inputs (integers and 2^-3 grid values) never hit."""
    p = a * b
    # two-product error term via splitting
    split = 134217729.0  # 2^27 + 1
    ah = a * split; ah = ah - (ah - a); al = a - ah
    bh = b * split; bh = bh - (bh - b); bl = b - bh
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    s = p + c
    # two-sum error
    bb = s - p
    e2 = (p - (s - bb)) + (c - bb)
    return s + (e + e2)
