"""FAISS-shaped flat indexes backed by the gfx950 kernels.

Mirrors the object protocol the reference binds at retrieve/retrieve_faiss.py:65-71::

    index = faiss.IndexFlatL2(d)          # :65
    index.add(train_fps)                  # :66
    distance, rank = index.search(q, k)   # :71

so ``import textreact_amd.faiss_compat as faiss`` is the whole change on the reference side
(INTEGRATION.md).  Same names, argument meaning and error behaviour as faiss' Python wrapper:
``add``/``search`` take 2-D arrays of any numeric dtype (faiss.swigfaiss replacement_add converts
them to float32; the reference passes int64 / int8 fingerprints, which the library here narrows on
its own worker threads -- same values, same results), a
dimension mismatch raises AssertionError, ``search`` returns ``(D float32[nq,k], I int64[nq,k])``
best first with ``I = -1`` / ``D = +-3.4e38`` padding when fewer than k vectors are indexed.

Extension beyond faiss (device-resident data, used by bench.py and the sharded search): ``add`` and
``search`` also accept a ``torch.Tensor`` living on the index's GPU (float32 or bfloat16); search
then returns torch tensors on that GPU and nothing crosses PCIe.

There is no CPU implementation behind these classes: without libtrxknn.so and a GPU they raise.
"""
import ctypes

import numpy as np

from . import _lib

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class IndexFlat:
    """Exact (brute-force) index.  metric: METRIC_INNER_PRODUCT or METRIC_L2."""

    def __init__(self, d, metric=METRIC_L2, device=None, tie_rule=None):
        self.d = int(d)
        self.metric_type = int(metric)
        self.is_trained = True
        if device is None:
            device = _default_device()
        self.device = int(device)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().trx_index_create(self.d, self.metric_type, self.device, ctypes.byref(self._h)))
        self.tie_rule = "id"
        self.set_tie_rule(_default_tie_rule() if tie_rule is None else tie_rule)

    def set_tie_rule(self, rule):
        """'id' (default): equal scores in id order, the total order of include/trx_knn.h.  'faiss': what faiss.IndexFlat
        itself returns on exact score ties -- for L2 the same thing, for the inner product its min-heap's order (TRX_TIES_FAISS
        in include/trx_knn.h).  `TRX_TIE_RULE=faiss` in the environment makes it the default of every index."""
        assert rule in ("id", "faiss"), "tie_rule is 'id' or 'faiss'"
        _lib.check(_lib.lib().trx_index_set_tie_rule(self._h, _lib.TIES_FAISS if rule == "faiss" else _lib.TIES_BY_ID))
        self.tie_rule = rule

    # -- faiss surface ------------------------------------------------------------------------
    @property
    def ntotal(self):
        return int(_lib.lib().trx_index_ntotal(self._h))

    def add(self, x):
        if _is_torch(x):
            return self._add_torch(x)
        x, dt = self._host_arg(x)
        _lib.check(_lib.lib().trx_index_add(self._h, x.ctypes.data_as(ctypes.c_void_p), x.shape[0], dt))

    def search(self, x, k):
        k = int(k)
        assert k > 0, "k must be positive"
        if _is_torch(x):
            return self._search_torch(x, k)
        x, dt = self._host_arg(x)
        nq = x.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib.check(_lib.lib().trx_index_search(self._h, x.ctypes.data_as(ctypes.c_void_p), nq, dt, k,
                                               D.ctypes.data_as(ctypes.c_void_p), I.ctypes.data_as(ctypes.c_void_p)))
        return D, I

    def reset(self):
        _lib.check(_lib.lib().trx_index_reset(self._h))

    # -- extensions ---------------------------------------------------------------------------
    def set_timing(self, enabled=True):
        _lib.check(_lib.lib().trx_index_set_timing(self._h, 1 if enabled else 0))

    def last_stats(self):
        st = _lib.SearchStats()
        _lib.check(_lib.lib().trx_index_last_stats(self._h, ctypes.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def search_s64(self, x, k):
        """torch-only: (D, I, S) with S the fp64 canonical scores (row-sharded merge input)."""
        return self._search_torch(x, int(k), want_s64=True)

    def search_s64_begin(self, x, k):
        """stream-ordered: everything is enqueued on torch's current stream and (D, I, S) are returned at once; they are
        final after `search_finish()` (include/trx_knn.h, Threading).  Work enqueued on the same stream in between -- the
        all-gather and merge of the row-sharded search -- needs no host round trip."""
        import torch
        x, dt = self._torch_arg(x)
        nq, k = x.shape[0], int(k)
        dev = torch.device("cuda", self.device)
        D = torch.empty((nq, k), dtype=torch.float32, device=dev)
        I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        S = torch.empty((nq, k), dtype=torch.float64, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._keep = x          # the queries must outlive the enqueued work
        _lib.check(_lib.lib().trx_index_search_device_begin(self._h, ctypes.c_void_p(x.data_ptr()), nq, dt, k, ctypes.c_void_p(D.data_ptr()),
                                                            ctypes.c_void_p(I.data_ptr()), ctypes.c_void_p(S.data_ptr()), st))
        return D, I, S

    def search_finish(self):
        """-> True when the finish changed outputs AFTER the enqueued work (more certificate failures than the inline
        re-scan covers: rare): whatever was computed from them in between must be redone"""
        _lib.check(_lib.lib().trx_index_search_finish(self._h))
        self._keep = None
        return bool(self.last_stats()["late_fallback"])

    # -- internals ----------------------------------------------------------------------------
    _HOST_DTYPES = {"float32": _lib.DTYPE_F32, "int8": _lib.DTYPE_I8, "int64": _lib.DTYPE_I64, "int32": _lib.DTYPE_I32,
                    "int16": _lib.DTYPE_I16, "uint8": _lib.DTYPE_U8, "float64": _lib.DTYPE_F64}

    def _host_arg(self, x):
        """-> (C-contiguous array, dtype code).  faiss' wrapper converts every array to float32 on the calling thread; here an
        integer or float64 array -- the int64 difference fingerprints of retrieve_faiss.py:24-33, the int8 Morgan bits of :36-44 --
        goes to the library as it is (include/trx_knn.h: TRX_DTYPE_I64 ...), which blocks, converts (worker threads) and overlaps
        the copies itself: the values the index sees, and so the results, are those of the float32 conversion.  Any other dtype (float16, uint16 ...) is converted here."""
        x = np.asarray(x)
        assert x.ndim == 2, "expected a 2-D array"
        assert x.shape[1] == self.d, "dimension mismatch: got %d, index has d=%d" % (x.shape[1], self.d)
        if x.dtype == np.bool_:
            x = x.view(np.int8)
        dt = self._HOST_DTYPES.get(x.dtype.name) if x.dtype.isnative else None
        if dt is None:
            return np.ascontiguousarray(x, dtype=np.float32), _lib.DTYPE_F32
        return np.ascontiguousarray(x), dt

    def _torch_arg(self, x):
        import torch
        assert x.dim() == 2 and x.shape[1] == self.d, "dimension mismatch: got %r, index has d=%d" % (tuple(x.shape), self.d)
        assert x.is_cuda and x.device.index == self.device, "tensor must live on cuda:%d" % self.device
        if x.dtype == torch.bfloat16:
            dt = _lib.DTYPE_BF16
        else:
            if x.dtype != torch.float32:
                x = x.float()
            dt = _lib.DTYPE_F32
        return x.contiguous(), dt

    def _add_torch(self, x):
        import torch
        x, dt = self._torch_arg(x)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().trx_index_add_device(self._h, ctypes.c_void_p(x.data_ptr()), x.shape[0], dt,
                                                   ctypes.c_void_p(st)))

    def _search_torch(self, x, k, want_s64=False):
        import torch
        x, dt = self._torch_arg(x)
        nq = x.shape[0]
        dev = torch.device("cuda", self.device)
        D = torch.empty((nq, k), dtype=torch.float32, device=dev)
        I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L = _lib.lib()
        if want_s64:
            S = torch.empty((nq, k), dtype=torch.float64, device=dev)
            _lib.check(L.trx_index_search_device_s64(self._h, ctypes.c_void_p(x.data_ptr()), nq, dt, k,
                                                     ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                                     ctypes.c_void_p(S.data_ptr()), st))
            return D, I, S
        _lib.check(L.trx_index_search_device(self._h, ctypes.c_void_p(x.data_ptr()), nq, dt, k,
                                             ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), st))
        return D, I

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().trx_index_destroy(h)
            except Exception:
                pass
            self._h = None


class IndexFlatIP(IndexFlat):
    def __init__(self, d, device=None, tie_rule=None):
        super().__init__(d, METRIC_INNER_PRODUCT, device, tie_rule)


class IndexFlatL2(IndexFlat):
    def __init__(self, d, device=None, tie_rule=None):
        super().__init__(d, METRIC_L2, device, tie_rule)


def merge_topk(metric, S_lists, I_lists, faiss_ties_k=None):
    """Cross-shard merge on the GPU (include/trx_knn.h: trx_merge_topk_device).
    S_lists float64 [nlists, nq, k], I_lists int64 [nlists, nq, k] (global ids), both on one GPU.
    faiss_ties_k: the lists are canonical top-2k lists of an inner-product search and the result is FAISS' own top k of
    the union (trx_merge_topk_device_s64 + trx_faiss_tie_order_device)."""
    import torch
    assert S_lists.is_cuda and I_lists.is_cuda and S_lists.shape == I_lists.shape and S_lists.dim() == 3
    S_lists = S_lists.contiguous().double()
    I_lists = I_lists.contiguous().long()
    nl, nq, k = S_lists.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=S_lists.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=S_lists.device)
    st = ctypes.c_void_p(torch.cuda.current_stream(S_lists.device.index).cuda_stream)
    if faiss_ties_k is None:
        _lib.check(_lib.lib().trx_merge_topk_device(int(metric), nl, nq, k, ctypes.c_void_p(S_lists.data_ptr()),
                                                    ctypes.c_void_p(I_lists.data_ptr()), ctypes.c_void_p(D.data_ptr()),
                                                    ctypes.c_void_p(I.data_ptr()), st))
        return D, I
    S = torch.empty((nq, k), dtype=torch.float64, device=S_lists.device)
    _lib.check(_lib.lib().trx_merge_topk_device_s64(int(metric), nl, nq, k, ctypes.c_void_p(S_lists.data_ptr()),
                                                    ctypes.c_void_p(I_lists.data_ptr()), ctypes.c_void_p(D.data_ptr()),
                                                    ctypes.c_void_p(I.data_ptr()), ctypes.c_void_p(S.data_ptr()), st))
    kk = int(faiss_ties_k)
    Df = torch.empty((nq, kk), dtype=torch.float32, device=S_lists.device)
    If = torch.empty((nq, kk), dtype=torch.int64, device=S_lists.device)
    _lib.check(_lib.lib().trx_faiss_tie_order_device(nq, k, kk, ctypes.c_void_p(S.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                                     ctypes.c_void_p(Df.data_ptr()), ctypes.c_void_p(If.data_ptr()), st))
    return Df, If


def _default_tie_rule():
    import os
    rule = os.environ.get("TRX_TIE_RULE", "id").lower()
    if rule not in ("id", "faiss"):
        raise ValueError("TRX_TIE_RULE is 'id' or 'faiss', not %r" % rule)
    return rule


def _default_device():
    import os
    lr = os.environ.get("TRX_DEVICE", os.environ.get("LOCAL_RANK"))   # TRX_DEVICE: several ranks rehearse on one GPU
    if lr is not None:
        return int(lr)
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:
        pass
    return 0
