"""The host side of the host entry points (textreact_amd/csrc/knn_host.cpp): the storage-type change the library applies to
the arrays the reference hands FAISS (int64 difference fingerprints retrieve_faiss.py:24-27, int8 Morgan bits :36-44) is
numpy's astype, value for value -- checked here without a GPU through trx_host_convert -- and, on the GPU, every host dtype
gives the oracle's answer through trx_index_add / trx_index_search."""
import ctypes

import numpy as np
import pytest

from _data import reaction_fp_like

L2 = 1
CODES = {np.int64: 3, np.int32: 4, np.int16: 5, np.uint8: 6, np.float64: 7, np.float32: 0, np.int8: 2}


def _convert(x, to):
    from textreact_amd import _lib
    out = np.empty(x.shape, np.int8 if to == 2 else np.float32)
    rc = _lib.lib().trx_host_convert(x.ctypes.data_as(ctypes.c_void_p), CODES[x.dtype.type], x.size, out.ctypes.data_as(ctypes.c_void_p), to)
    return rc, out


@pytest.mark.parametrize("dt", [np.int64, np.int32, np.int16, np.uint8])
def test_integers_narrow_to_int8_exactly_or_say_they_do_not_fit(dt):
    rng = np.random.default_rng(1)
    lo, hi = (0, 128) if dt == np.uint8 else (-128, 128)
    for shape in ((1, 1), (3, 4097), (257, 2048), (5000, 1111)):       # below one task, ragged against the 4096-element task grid
        x = rng.integers(lo, hi, shape).astype(dt)
        rc, out = _convert(x, 2)
        assert rc == 0 and np.array_equal(out, x.astype(np.int8)) and np.array_equal(out.astype(dt), x)
    # one value beyond a signed byte anywhere -- first element, last element, the ends of the type -- and the answer is "no"
    info = np.iinfo(dt)
    for pos, val in ((0, 128), (-1, 128), (x.size // 2, info.max)) + (() if dt == np.uint8 else ((7, -129), (-2, info.min))):
        y = x.copy(); y.flat[pos] = val
        assert _convert(y, 2)[0] == 1
    # ... and the ends that do fit
    if dt != np.uint8:
        y = x.copy(); y.flat[0] = -128; y.flat[-1] = 127
        rc, out = _convert(y, 2)
        assert rc == 0 and np.array_equal(out.astype(dt), y)


@pytest.mark.parametrize("dt", [np.int64, np.int32, np.int16, np.uint8, np.float64])
def test_the_float32_route_is_numpys_astype(dt):
    rng = np.random.default_rng(2)
    if dt == np.float64:
        x = rng.standard_normal((301, 777)) * 10.0 ** rng.integers(-30, 30, (301, 777))
        x.flat[:6] = [np.inf, -np.inf, np.nan, 1e300, -1e300, 2.0 ** -150]      # overflow to inf, underflow to 0 / denormal: as astype
    else:
        info = np.iinfo(dt)
        x = rng.integers(info.min, info.max, (301, 777), dtype=dt, endpoint=True)
        x.flat[:4] = [info.max, info.min, min(info.max, 16777217), min(info.max, 2 ** 53 + 1) if dt == np.int64 else 0]
    rc, out = _convert(x, 0)
    with np.errstate(over="ignore"):
        want = x.astype(np.float32)
    assert rc == 0 and np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_convert_refuses_what_it_is_not_for():
    from textreact_amd import _lib
    x = np.zeros(8, np.float32); o = np.zeros(8, np.float32)
    L = _lib.lib()
    assert L.trx_host_convert(x.ctypes.data_as(ctypes.c_void_p), 0, 8, o.ctypes.data_as(ctypes.c_void_p), 2) == -1      # floats never narrow
    assert L.trx_host_convert(x.ctypes.data_as(ctypes.c_void_p), 99, 8, o.ctypes.data_as(ctypes.c_void_p), 0) == -1
    assert L.trx_host_convert(None, 3, 8, o.ctypes.data_as(ctypes.c_void_p), 0) == -1
    assert L.trx_host_convert(None, 3, 0, None, 0) == 0
    assert L.trx_host_threads() >= 1


def test_faiss_compat_hands_integer_arrays_over_as_they_are():
    """no float32 copy on the Python side for the dtypes the library takes; anything else still becomes float32 here"""
    import textreact_amd.faiss_compat as faiss
    from textreact_amd import _lib
    idx = faiss.IndexFlat.__new__(faiss.IndexFlat); idx.d = 16
    for dt, code in ((np.int64, _lib.DTYPE_I64), (np.int32, _lib.DTYPE_I32), (np.int16, _lib.DTYPE_I16), (np.uint8, _lib.DTYPE_U8),
                     (np.int8, _lib.DTYPE_I8), (np.bool_, _lib.DTYPE_I8), (np.float64, _lib.DTYPE_F64), (np.float32, _lib.DTYPE_F32)):
        x = np.ones((5, 16), dt)
        y, c = idx._host_arg(x)
        assert c == code and y.ctypes.data == x.ctypes.data, dt
    for dt in (np.float16, np.uint16, np.uint32, np.uint64, np.dtype(">i4")):
        y, c = idx._host_arg(np.ones((5, 16), dt))
        assert c == _lib.DTYPE_F32 and y.dtype == np.float32
    y, c = idx._host_arg(np.ones((5, 32), np.int64)[:, ::2])       # a strided view is made contiguous, in its own dtype
    assert c == _lib.DTYPE_I64 and y.flags.c_contiguous and y.dtype == np.int64
    with pytest.raises(AssertionError):
        idx._host_arg(np.ones((5, 17), np.int64))


# ---- GPU: every host dtype through trx_index_add / trx_index_search == the oracle on the float32 values -----------------

def _oracle(x, y, k):
    from oracle import flat_knn as oracle
    return oracle.knn_faiss(L2, x.astype(np.float32), y.astype(np.float32), k)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.int64, np.int32, np.int16, np.uint8, np.int8, np.float64])
def test_host_dtypes_like_the_reference(dt):
    """retrieve_faiss.py:66,71 with the array types numpy gives the reference's fingerprints.  Counts that fit a signed byte run
    the int8 scan whatever they were stored as; the answer is the oracle's on the float32 values."""
    import textreact_amd.faiss_compat as faiss
    y = reaction_fp_like(3000, 512, 3)
    if dt == np.uint8:
        y = np.abs(y)
    y = y.astype(dt)
    idx = faiss.IndexFlatL2(512)
    idx.add(y[:1000]); idx.add(y[1000:])
    D, I = idx.search(y[:300], 20)
    Dr, Ir = _oracle(y[:300], y, 20)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    assert (D[:, 0] == 0).all()
    st = idx.last_stats()
    assert st["exact_class"] == 1 and st["int8_scan"] >= 1, st


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.int64, np.int32, np.int16, np.uint8])
def test_integers_beyond_a_signed_byte_take_the_float32_route(dt):
    """a count of 200 in one row (and, for the signed types, -300 in a query): the block that holds it crosses as the float32
    FAISS would have seen; blocks before and after it may be int8; the index and the answers mix both"""
    import textreact_amd.faiss_compat as faiss
    y = np.abs(reaction_fp_like(4000, 256, 5)).astype(dt)
    y[2500, 17] = 200
    x = y[2400:2600].copy()
    if dt != np.uint8:
        x[5, 3] = -300 if dt != np.int8 else x[5, 3]
    idx = faiss.IndexFlatL2(256)
    for part in np.array_split(y, 4):      # the third call holds the 200
        idx.add(part)
    D, I = idx.search(x, 20)
    Dr, Ir = _oracle(x, y, 20)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)


@pytest.mark.gpu
def test_many_chunks_and_blocks_of_int64_queries():
    """more than one 65,536-query block and more than one 16 MiB staging chunk per block (70,000 x 512 int64 = 287 MB), a
    value beyond int8 in the LAST chunk of the second block only (that block is staged twice: narrow, then float32)"""
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = reaction_fp_like(2000, 512, 7).astype(np.int64)
    x = reaction_fp_like(70000, 512, 8).astype(np.int64)
    x[-1, -1] = 1000
    idx = faiss.IndexFlatL2(512)
    idx.add(y)
    D, I = idx.search(x, 5)
    assert idx.last_stats()["nq"] == 70000
    sel = np.r_[0:300, 65400:65700, 69700:70000]
    Dr, Ir = oracle.knn_faiss(L2, x[sel].astype(np.float32), y.astype(np.float32), 5)
    assert np.array_equal(I[sel], Ir) and np.array_equal(D[sel], Dr)
    # the same queries as float32 (the route faiss' wrapper takes): identical arrays
    D2, I2 = idx.search(x.astype(np.float32), 5)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)


@pytest.mark.gpu
def test_float64_embeddings_are_rounded_to_float32_first():
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    rng = np.random.default_rng(3)
    y = rng.standard_normal((5000, 96)); x = rng.standard_normal((200, 96))
    for metric, cls in ((0, faiss.IndexFlatIP), (1, faiss.IndexFlatL2)):
        idx = cls(96)
        idx.add(y)
        D, I = idx.search(x, 10)
        Dr, Ir = oracle.knn_canonical(metric, x.astype(np.float32), y.astype(np.float32), 10)
        assert np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32))


@pytest.mark.gpu
def test_odd_dimensions_through_the_integer_types_and_the_exact_scan():
    """the extended fuzzer's find (round 6): integer host rows reach the device as int8 and are widened to bf16 rows of d components --
    for an odd d every other row starts on an odd halfword, and the exact fp64 scan's 16-byte loads of such rows returned garbage
    (device bf16 queries of an odd d always could; nothing passed them until the host path did).  d = 65, bit vectors, L2, the
    two-scan path (k = 100) and k beyond the corpus (k = 256 over 7 rows): the oracle's answer for every host type"""
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    from _data import morgan_like
    for (seed, n, nq, k) in ((146, 7, 255, 256), (124, 5000, 63, 100)):
        y, x = morgan_like(n, 65, seed).astype(np.float32), morgan_like(nq, 65, seed + 1).astype(np.float32)
        Dr, Ir = oracle.knn_canonical(L2, x, y, k)
        for dt in (np.float32, np.int8, np.int64):
            idx = faiss.IndexFlatL2(65)
            idx.add(y.astype(dt))
            D, I = idx.search(x.astype(dt), k)
            assert np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), (dt, n, k, idx.last_stats())
    import torch
    yb, xb = torch.from_numpy(y).cuda().bfloat16(), torch.from_numpy(x).cuda().bfloat16()      # device bf16 rows of an odd d: the same loads
    idx = faiss.IndexFlatL2(65); idx.add(yb)
    D, I = idx.search(xb, k)
    assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy().view(np.uint32), Dr.view(np.uint32))


def test_conversions_from_several_threads_take_turns():
    """the worker pool serves one caller at a time: eight Python threads converting at once (ctypes releases the GIL) all get numpy's answer"""
    import threading
    rng = np.random.default_rng(5)
    xs = [rng.integers(-100, 100, (300, 1000 + 37 * i)).astype(np.int64) for i in range(8)]
    outs = [None] * 8

    def work(i):
        for _ in range(20):
            rc, out = _convert(xs[i], 2)
            assert rc == 0
        outs[i] = out
    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for i in range(8):
        assert outs[i] is not None and np.array_equal(outs[i], xs[i].astype(np.int8))


def test_conversion_properties_hold_for_arbitrary_arrays():
    """property test (hypothesis, 60 examples): for any int64 array, narrowing succeeds exactly when every value fits a signed byte and
    then reproduces the values; the float32 route equals numpy's astype bit for bit"""
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp

    @settings(max_examples=60, deadline=None)
    @given(hnp.arrays(np.int64, hnp.array_shapes(min_dims=2, max_dims=2, min_side=1, max_side=300),
                      elements=st.one_of(st.integers(-128, 127), st.integers(-2 ** 63, 2 ** 63 - 1), st.integers(-300, 300))))
    def check(x):
        x = np.ascontiguousarray(x)
        rc, out = _convert(x, 2)
        fits = bool(((x >= -128) & (x <= 127)).all())
        assert rc == (0 if fits else 1)
        if fits:
            assert np.array_equal(out.astype(np.int64), x)
        rc, of = _convert(x, 0)
        assert rc == 0 and np.array_equal(of.view(np.uint32), x.astype(np.float32).view(np.uint32))
    check()


@pytest.mark.gpu
def test_int64_values_whose_float32_squares_leave_float32():
    """a count of 2^62 in an int64 fingerprint (garbage upstream, but it is what the array holds): as float32 it is 4.6e18, its row's
    |row|^2 overflows float32 -- a hostile row that arrived through the integer route; the answer is the oracle's on
    the float32 values, both metrics"""
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = reaction_fp_like(3000, 128, 9).astype(np.int64); x = reaction_fp_like(100, 128, 10).astype(np.int64)
    y[77, 5] = 2 ** 62; y[78, 6] = -(2 ** 62); x[3, 5] = 2 ** 61
    for metric, cls in ((0, faiss.IndexFlatIP), (1, faiss.IndexFlatL2)):
        idx = cls(128); idx.add(y)
        D, I = idx.search(x, 10)
        with np.errstate(all="ignore"):
            Dr, Ir = oracle.knn_canonical(metric, x.astype(np.float32), y.astype(np.float32), 10)
        assert np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), (metric, idx.last_stats())
