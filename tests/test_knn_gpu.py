"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Bar: I identical (int64) and D identical (float32 bits) to oracle.knn_canonical on every input
class; on inputs whose fp32 partial sums are exact the oracle's literal FAISS restatement gives
the same answer too (checked in tests/test_oracle.py on the CPU side).
"""
import os
import sys

import numpy as np
import pytest

from _data import bf16_round, gaussian, grid, morgan_like, reaction_fp_like

pytestmark = pytest.mark.gpu

IP, L2 = 0, 1


def _index(metric, d):
    import textreact_amd.faiss_compat as faiss
    return faiss.IndexFlatIP(d) if metric == IP else faiss.IndexFlatL2(d)


def _check(metric, x, y, k, chunks=1, expect_exact_class=None):
    from oracle import flat_knn as oracle
    idx = _index(metric, y.shape[1])
    for part in np.array_split(y, chunks):
        idx.add(part)
    assert idx.ntotal == y.shape[0]
    D, I = idx.search(x, k)
    Dr, Ir = oracle.knn_canonical(metric, x, y, k)
    st = idx.last_stats()
    bad = np.argwhere(I != Ir)
    assert bad.size == 0, "index mismatch at %r (first: got %r want %r) stats=%r" % (
        bad[:5].tolist(), I[bad[0][0]].tolist() if bad.size else None, Ir[bad[0][0]].tolist() if bad.size else None, st)
    assert np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), "distance bits differ, stats=%r" % (st,)
    if expect_exact_class is not None:
        assert st["exact_class"] == int(expect_exact_class), st
    return st


@pytest.mark.parametrize("metric", [IP, L2])
def test_c0_gaussian_fp32(metric):
    # BASELINE.json configs[0]: 10k x 768 fp32 corpus, 1k queries, top-10.  Since round 4 the operand is the bf16 rounding
    # of the fp32 data (K = d), every row within twice the rounding's key error of a query's bound is listed, and the
    # candidates are certified on their exact scores (approx mode; the three-term split, K = 3d, is the next test)
    st = _check(metric, gaussian(1000, 768, 5678), gaussian(10000, 768, 1234), 10)
    assert st["k_split"] == 768 and st["n_uncertified"] == 0


@pytest.mark.parametrize("metric", [IP, L2])
def test_fp32_queries_against_a_bf16_index_and_back(metric):
    """the approx mode's other doors: fp32 queries that bf16 does not hold against an index of bf16 values (nothing about the
    index changes), bf16 queries against an fp32 index, fp32 rows added to an index that held bf16 values, ragged sizes"""
    yb, yf = bf16_round(gaussian(9000, 200, 3)), gaussian(4001, 200, 4)
    xf, xb = gaussian(333, 200, 5), bf16_round(gaussian(130, 200, 6))
    st = _check(metric, xf, yb, 10)
    assert st["k_split"] == 256 and st["n_uncertified"] == 0
    _check(metric, xb, yf, 10)
    _check(metric, xf, np.concatenate([yb, yf]), 10, chunks=3)


@pytest.mark.parametrize("metric", [IP, L2])
def test_gaussian_bf16_values(metric):
    # bf16-representable inputs -> plain operand, K = d
    st = _check(metric, bf16_round(gaussian(700, 768, 5678)), bf16_round(gaussian(20000, 768, 1234)), 10)
    assert st["k_split"] == 768 and st["n_uncertified"] == 0


@pytest.mark.parametrize("metric", [IP, L2])
def test_grid_ties(metric):
    _check(metric, grid(300, 128, 2), grid(9000, 128, 1), 10)


def _tie_inputs():
    """inner-product inputs with exact score ties, every fp32 partial sum exact (so the literal FAISS restatement's fp32
    scores ARE the scores): the 2^-3 grid at small d (ties everywhere), integer counts, bit vectors, and Gaussian rows
    repeated in blocks that straddle the k-th place (identical rows score identically in any arithmetic)"""
    yg, xg = grid(9000, 24, 41), grid(300, 24, 42)
    yg[4000:4040] = yg[7]; yg[8000:8013] = yg[7]; xg[0] = yg[7]; xg[1] = yg[4001]
    yield "grid", xg, yg, True
    yc = reaction_fp_like(6000, 512, 43, density=0.01); yc[100:160] = yc[5]
    yield "counts", yc[:200].copy(), yc, True
    ym = morgan_like(5000, 256, 44)
    yield "bits", ym[:150].copy(), ym, True
    yd = bf16_round(gaussian(6000, 64, 45))
    yd[1000:1017] = yd[2]; yd[3000:3009] = yd[2]; yd[5990:6000] = yd[2]
    xd = bf16_round(gaussian(100, 64, 46)); xd[0] = yd[2]; xd[1] = -yd[2]
    yield "duplicates", xd, yd, False


@pytest.mark.parametrize("k", [1, 3, 10, 12, 20])
def test_faiss_tie_rule_for_the_inner_product(k):
    """TRX_TIES_FAISS: on inputs with exact score ties the HIP index returns what faiss.IndexFlatIP's heap returns -- I and D of
    the literal restatement (oracle.knn_faiss: blocks, strict admission, (score, id) min-heap, heap_reorder) -- where the
    default rule returns (score desc, id asc); L2 is unaffected by the rule (FAISS' max-heap IS the total order)"""
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    told_apart = 0
    for name, x, y, exact in _tie_inputs():
        idx = faiss.IndexFlatIP(y.shape[1], tie_rule="faiss")
        idx.add(y[:len(y) // 2]); idx.add(y[len(y) // 2:])
        D, I = idx.search(x, k)
        Df, If = oracle.knn_faiss(IP, x, y, k)
        Dc, Ic = oracle.knn_canonical(IP, x, y, k)
        if exact:
            assert np.array_equal(I, If), (name, k, np.argwhere(I != If)[:3].tolist())
            assert np.array_equal(D.view(np.uint32), Df.view(np.uint32)), (name, k)
        else:       # Gaussian rows: only the duplicated block ties exactly; compare the queries that look at it
            assert np.array_equal(I[:2], If[:2]) and np.array_equal(np.sort(I, 1), np.sort(Ic, 1)), (name, k)
        told_apart += int(not np.array_equal(If, Ic))
        idx.set_tie_rule("id")
        D, I = idx.search(x, k)
        assert np.array_equal(I, Ic) and np.array_equal(D.view(np.uint32), Dc.view(np.uint32)), (name, k)
        l2 = faiss.IndexFlatL2(y.shape[1], tie_rule="faiss"); l2.add(y)
        D, I = l2.search(x, k)
        Df, If = oracle.knn_faiss(L2, x, y, k) if exact else oracle.knn_canonical(L2, x, y, k)
        assert np.array_equal(I, If) and np.array_equal(D.view(np.uint32), Df.view(np.uint32)), (name, k, "L2")
    assert told_apart >= (2 if k > 1 else 0)       # (k = 1: the heap keeps the first row it sees, which is the smallest id -- both rules agree)


def test_faiss_tie_rule_edges(monkeypatch):
    """fewer rows than k (pads behind, ties still FAISS-ordered), a corpus of identical rows, torch tensors on the device, the
    environment default, k beyond the rule's range"""
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = np.ones((6, 8), dtype=np.float32); x = np.ones((25, 8), dtype=np.float32)
    monkeypatch.setenv("TRX_TIE_RULE", "faiss")
    idx = faiss.IndexFlatIP(8); assert idx.tie_rule == "faiss"
    monkeypatch.delenv("TRX_TIE_RULE")
    idx.add(y)
    for k in (4, 6, 9):
        D, I = idx.search(x, k)
        Df, If = oracle.knn_faiss(IP, x, y, k)
        assert np.array_equal(I, If) and np.array_equal(D, Df), (k, I[0], If[0])
    y2 = grid(3000, 16, 51); x2 = grid(70, 16, 52)
    idx = faiss.IndexFlatIP(16, tie_rule="faiss"); idx.add(torch.from_numpy(y2).cuda())
    D, I = idx.search(torch.from_numpy(x2).cuda(), 10)
    Df, If = oracle.knn_faiss(IP, x2, y2, 10)
    assert np.array_equal(I.cpu().numpy(), If) and np.array_equal(D.cpu().numpy(), Df)
    D, I, S = idx.search_s64(torch.from_numpy(x2).cuda(), 10)
    assert np.array_equal(I.cpu().numpy(), If) and np.array_equal(S.float().cpu().numpy(), Df)
    with pytest.raises(AssertionError):
        idx.search(x2, 1025)
    assert faiss.IndexFlatIP(16).tie_rule == "id"


def test_reaction_fingerprints_l2_k20():
    # the reference's own call: IndexFlatL2, k = 20, d = 2048 integer counts, train searches itself
    y = reaction_fp_like(6000, 2048, 7)
    st = _check(L2, y[:500], y, 20, expect_exact_class=True)
    assert st["n_uncertified"] == 0


def test_morgan_fingerprints_l2_k20():
    y = morgan_like(8000, 1024, 8)
    y[100:140] = y[100]  # exact duplicates: distance-0 ties resolved by id
    _check(L2, y[:300], y, 20, expect_exact_class=True)


@pytest.mark.parametrize("metric,d,maker", [(L2, 2048, reaction_fp_like), (IP, 2048, reaction_fp_like), (L2, 1024, morgan_like), (IP, 1000, reaction_fp_like),
                                            (L2, 264, reaction_fp_like)])
def test_int8_form_of_the_scan_for_the_integer_class(metric, d, maker):
    """integer inputs that fit a signed byte run the scan in int8 (stats int8_scan): against the oracle like every other
    path, and bit-identical to the bf16 form of the same search (TRX_NO_I8), incl. a second add and a ragged query count"""
    y = maker(7000, d, 21)
    x = np.concatenate([y[:300], maker(133, d, 22)])
    form = 2 if maker is morgan_like else 1                # bit vectors take the fp4 form (next test), counts up to 10 the int8 form
    st = _check(metric, x, y, 20, chunks=2, expect_exact_class=True)
    assert st["int8_scan"] == form and st["n_uncertified"] == 0, st
    idx = _index(metric, d); idx.add(y[:5000]); idx.add(y[5000:])
    D8, I8 = idx.search(x, 20)
    assert idx.last_stats()["int8_scan"] == form
    os.environ["TRX_NO_I8"] = "1"
    try:
        D16, I16 = idx.search(x, 20)
        assert idx.last_stats()["int8_scan"] == 0
    finally:
        del os.environ["TRX_NO_I8"]
    assert np.array_equal(I8, I16) and np.array_equal(D8.view(np.uint32), D16.view(np.uint32))


def _e2m1_like(n, d, seed, density=0.1, signed=True):
    rng = np.random.default_rng(seed)
    vals = np.array([1, 1, 1, 2, 2, 3, 4, 6], dtype=np.float32)[rng.integers(0, 8, (n, d))]
    if signed:
        vals *= rng.choice(np.array([-1.0, 1.0], dtype=np.float32), (n, d))
    return ((rng.random((n, d)) < density) * vals).astype(np.float32)


@pytest.mark.parametrize("metric", [IP, L2])
@pytest.mark.parametrize("d,maker", [(1024, morgan_like), (1000, _e2m1_like), (2048, _e2m1_like), (300, morgan_like)])
def test_fp4_form_of_the_scan_for_bit_vectors_and_tiny_counts(metric, d, maker):
    """every value on both sides one of 0, +-1, +-2, +-3, +-4, +-6 (what E2M1 holds; the reference's Morgan fingerprints are
    0 / 1): the scan runs on v_mfma_f32_16x16x128_f8f6f4 with fp4 operands (stats int8_scan == 2) -- the oracle's answer, and
    bit-identical to the int8 form (TRX_NO_FP4) and the bf16 form (TRX_NO_I8) of the same search; a second add, a ragged query
    count, widths that are not a multiple of 32"""
    y = maker(7000, d, 51)
    x = np.concatenate([y[:300], maker(133, d, 52)])
    st = _check(metric, x, y, 20, chunks=2, expect_exact_class=True)
    assert st["int8_scan"] == 2 and st["n_uncertified"] == 0, st
    idx = _index(metric, d); idx.add(y[:5000]); idx.add(y[5000:])
    D4, I4 = idx.search(x, 20)
    assert idx.last_stats()["int8_scan"] == 2
    for env, want in (("TRX_NO_FP4", 1), ("TRX_NO_I8", 0)):
        os.environ[env] = "1"
        try:
            D, I = idx.search(x, 20)
            assert idx.last_stats()["int8_scan"] == want
        finally:
            del os.environ[env]
        assert np.array_equal(I4, I) and np.array_equal(D4.view(np.uint32), D.view(np.uint32)), env


def test_the_fp4_form_is_gated_on_the_device_and_by_the_corpus():
    y = _e2m1_like(6000, 512, 61)
    five = y[:200].copy(); five[:, 3] = 5.0                 # 5 is an int8 but not an E2M1 number
    assert _check(L2, y[:200], y, 10)["int8_scan"] == 2
    assert _check(L2, five, y, 10)["int8_scan"] == 1        # the QUERIES decide per search, on the device ...
    assert _check(IP, y[:200] * 0.5, y, 10)["int8_scan"] == 0
    y5 = y.copy(); y5[17, 0] = 5.0
    assert _check(L2, y[:200], y5, 10)["int8_scan"] == 1    # ... a corpus value outside the set rules the form out for good
    idx = _index(L2, 512); idx.add(y); idx.add(y5[:100])    # (also when it arrives with a later add)
    idx.search(y[:64], 10)
    assert idx.last_stats()["int8_scan"] == 1


def test_the_int8_form_is_gated_on_the_device():
    """what decides is the QUERIES of each search, on the device: integers up to 63 (L2 stages 2 x) or 127 (IP) take the int8
    form, larger counts, fractions or a narrow index take the bf16 form -- same answers either way"""
    y = reaction_fp_like(6000, 512, 31)
    big = y.copy(); big[:, :7] *= 9.0                       # counts up to 90: too large doubled, fine as they are
    for metric, x, want in ((L2, y[:200], 1), (L2, big[:200], 0), (IP, big[:200], 1), (L2, y[:200] + 0.5, 0), (IP, y[:200] * 0.25, 0)):
        st = _check(metric, x.astype(np.float32), y, 10)
        assert st["int8_scan"] == want, (metric, want, st)
    st = _check(L2, big[:200], big, 10)                     # corpus counts beyond 127? no: 90 -- but doubled queries are not bytes
    assert st["int8_scan"] == 0
    huge = y.copy(); huge[:, 0] = 200.0
    assert _check(IP, y[:100], huge, 10)["int8_scan"] == 0      # a corpus value that is not a signed byte: never
    assert _check(L2, reaction_fp_like(100, 128, 5), reaction_fp_like(3000, 128, 6), 10)["int8_scan"] == 0      # d < 256


@pytest.mark.parametrize("metric", [IP, L2])
def test_fingerprints_ip_and_l2_ties(metric):
    y = morgan_like(5000, 256, 9)
    _check(metric, y[:200], y, 20)


@pytest.mark.parametrize("nq", [1, 5, 19, 20, 257])
def test_small_query_counts(nq):
    _check(IP, gaussian(nq, 96, 3), gaussian(3000, 96, 4), 10)


@pytest.mark.parametrize("n,d", [(1, 8), (7, 100), (255, 33), (256, 64), (257, 65), (1000, 130)])
def test_ragged_shapes(n, d):
    for metric in (IP, L2):
        _check(metric, gaussian(33, d, 11), gaussian(n, d, 12), 10)


def test_k_larger_than_n_and_empty():
    import textreact_amd.faiss_compat as faiss
    idx = faiss.IndexFlatIP(16)
    D, I = idx.search(gaussian(3, 16, 1), 4)
    assert (I == -1).all() and (D == -np.finfo(np.float32).max).all()
    _check(IP, gaussian(9, 16, 1), gaussian(4, 16, 2), 6)
    _check(L2, gaussian(9, 16, 1), gaussian(4, 16, 2), 6)


def test_sorted_corpus_bursts():
    # adversarial order: similarity to every query increases with the row id, so each tile beats
    # everything before it and every append buffer overflows (dense rebuild path)
    rng = np.random.default_rng(5)
    base = rng.standard_normal(64).astype(np.float32)
    scale = np.linspace(0.1, 4.0, 5000, dtype=np.float32)[:, None]
    y = scale * base[None, :] + 0.01 * rng.standard_normal((5000, 64)).astype(np.float32)
    x = base[None, :] + 0.05 * rng.standard_normal((300, 64)).astype(np.float32)
    _check(IP, x, y, 10)


@pytest.mark.parametrize("k", [10, 20])
def test_adversarial_order_over_long_splits(k, monkeypatch):
    """the same ordering over 586 tiles in 4 splits of 146: every tile beats all earlier ones for every query, so own and
    shared thresholds always lag, lists fill and are compacted again and again -- still every row accounted for"""
    monkeypatch.setenv("TRX_NSPLITS", "4")
    rng = np.random.default_rng(15)
    n = 150_000
    base = rng.standard_normal(64).astype(np.float32)
    scale = np.linspace(0.1, 4.0, n, dtype=np.float32)[:, None]
    y = scale * base[None, :] + 0.01 * rng.standard_normal((n, 64)).astype(np.float32)
    x = base[None, :] + 0.05 * rng.standard_normal((300, 64)).astype(np.float32)
    st = _check(IP, x, y, k)
    assert st["n_splits"] == 4
    # L2: rows on a ray from the queries' neighbourhood, the radius shrinking with the row id
    u = rng.standard_normal(64).astype(np.float32); u /= np.linalg.norm(u)
    radius = np.linspace(40.0, 0.5, n, dtype=np.float32)[:, None]
    y2 = base[None, :] + radius * u[None, :] + 0.01 * rng.standard_normal((n, 64)).astype(np.float32)
    _check(L2, x, y2, k)


@pytest.mark.parametrize("n", [5000 + 37, 150_000 + 201])
def test_pad_rows_of_an_inner_product_index_never_count(n, monkeypatch):
    """inner product, NEGATIVE small-integer scores rising with the row id, n % 256 != 0: the zero pad rows of the last
    tile score 0, above every real row, and the lists are full when the last tile arrives (adversarial order), so the
    compaction sees them.  Exact-class inputs: no certificate stands behind the answer."""
    monkeypatch.setenv("TRX_NSPLITS", "4")
    rng = np.random.default_rng(23)
    d = 64
    x = np.zeros((300, d), dtype=np.float32); x[:, :8] = rng.integers(1, 4, (300, 8))
    # corpus row i: -(a few units) on the first 8 components, magnitude falling with i -> x . y negative, rising with i
    mag = np.linspace(9.0, 1.0, n).astype(np.int64)[:, None]
    y = np.zeros((n, d), dtype=np.float32); y[:, :8] = -(mag + rng.integers(0, 2, (n, 8)))
    _check(IP, x, y, 10, expect_exact_class=True)


def test_add_in_chunks_and_mode_transition():
    y = np.concatenate([bf16_round(gaussian(3000, 64, 1)), gaussian(2000, 64, 2)])  # exact block, then fp32 block
    _check(IP, gaussian(100, 64, 3), y, 10, chunks=5)
    _check(L2, gaussian(100, 64, 3), y, 10, chunks=5)


def test_bf16_corpus_fp32_queries():
    _check(IP, gaussian(100, 64, 3), bf16_round(gaussian(3000, 64, 1)), 10)


@pytest.mark.parametrize("k", [1, 12, 13, 24, 25, 64])
def test_k_range(k):
    _check(IP, gaussian(40, 48, 3), gaussian(2500, 48, 4), k)
    _check(L2, gaussian(40, 48, 3), gaussian(2500, 48, 4), k)


def _clustered(n, nq, d, seed):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((40, d)).astype(np.float32)
    y = c[rng.integers(0, 40, n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    x = c[rng.integers(0, 40, nq)] + 0.3 * rng.standard_normal((nq, d)).astype(np.float32)
    return y, x


@pytest.mark.parametrize("k", [25, 64, 100, 256])
@pytest.mark.parametrize("kind", ["bf16", "fp32", "fingerprints", "clustered"])
def test_two_scan_path_for_k_above_24(k, kind):
    """TRX_FAST_MAX_K < k <= TRX_WIDE_MAX_K: a first scan ranks 24 rows per query, a second lists every row above a threshold
    extrapolated from them and the wide re-score proves the k best; a query left with fewer than k rows (n_rescored) gets a
    better threshold from the rows it did find and a third scan (n_rescanned); only what fails that takes the exact scan.  The
    oracle's answer in every case; on continuous data, clustered or not, (almost) nobody needs the exact scan."""
    if kind == "fingerprints":
        y = reaction_fp_like(20000, 512, 61); x = np.concatenate([y[:200], reaction_fp_like(56, 512, 62)])
    elif kind == "clustered":      # 1,000-row clusters: the scores of a query fall off a cliff the first 24 know nothing about
        y, x = _clustered(40000, 300, 96, 65)
        y, x = bf16_round(y), bf16_round(x)
    else:
        y, x = gaussian(40000, 96, 63), gaussian(300, 96, 64)
        if kind == "bf16":
            y, x = bf16_round(y), bf16_round(x)
    for metric in (IP, L2):
        st = _check(metric, x, y, k)
        assert st["n_rescanned"] == st["n_rescored"], st                # every unproven query had room in the third scan
        if kind != "fingerprints":                                        # (count data ties by the hundred at the k-th place)
            assert st["n_rescored"] <= x.shape[0] // (3 if kind == "clustered" else 10), st
            assert st["n_uncertified"] <= x.shape[0] // 100, st


def test_two_scan_path_on_small_and_awkward_indexes():
    # fewer rows than k, fewer than 24, a crowd of near-duplicates around the k-th place, k = 257 (the exact scan for every query)
    for n, k in ((10, 25), (30, 100), (300, 256), (5000, 257)):
        _check(IP, gaussian(37, 48, 5), gaussian(n, 48, 6), k)
        _check(L2, gaussian(37, 48, 5), gaussian(n, 48, 6), k)
    y = gaussian(6000, 64, 1); c = gaussian(1, 64, 2)
    y[1000:1400] = c * (1.0 - 1e-7 * np.arange(400, dtype=np.float32)[:, None])
    _check(IP, np.repeat(c, 4, axis=0), y, 100)


@pytest.mark.parametrize("k", [5, 300, 2048])
def test_exact_scan_selection_radix_descent_sort_and_ties(k):
    """the exact scan's selection (k > TRX_WIDE_MAX_K, and every fall-back): radix descent to the k-th key, one sort; more rows
    TIED at the k-th score than the sort holds (12,000 copies of three rows; count data) -> the first of them in row order"""
    x = gaussian(9, 32, 11)
    _check(IP, x, gaussian(30000, 32, 12), k)
    _check(L2, x, gaussian(30000, 32, 12), k)
    three = gaussian(3, 32, 13)
    y = three[np.random.default_rng(14).integers(0, 3, 12000)]
    if k > 24:                                                           # (k <= 24 only reaches the exact scan as a fall-back)
        _check(IP, x, y, k)
        _check(L2, x, y, k)
    fp = reaction_fp_like(15000, 256, 15)
    _check(IP, fp[:7], fp, max(k, 257))
    _check(L2, fp[:7], fp, max(k, 257))
    _check(IP, x, gaussian(100, 32, 16), max(k, 257))                    # fewer rows than k


def test_near_duplicate_cluster_is_resolved_by_the_wide_rescore():
    # 200 rows within ~1e-7 of each other, far above everything else: the 32 candidates the select kernel re-scores cannot
    # prove the top 10 (the 33rd row ties with them), so the query is flagged -- and the second tier (round 4) re-scores ALL
    # rows its lists hold above their bound, the whole cluster, and certifies the answer without scanning the index
    rng = np.random.default_rng(6)
    y = gaussian(4000, 64, 1)
    c = gaussian(1, 64, 2)
    y[1000:1200] = c * (1.0 + 1e-7 * rng.standard_normal((200, 1)).astype(np.float32))
    x = np.repeat(c, 8, axis=0) + 1e-3 * gaussian(8, 64, 3)
    st = _check(IP, x, y, 10)
    assert st["n_rescored"] == 8 and st["n_uncertified"] == 0
    st = _check(L2, x, y, 10)
    assert st["n_rescored"] == 8 and st["n_uncertified"] == 0


def test_a_crowd_around_the_kth_place_is_resolved_by_the_wide_rescore():
    # the k-th place INSIDE a crowd that reaches down to the lists' bound: scores fall off smoothly (steps far below the rounding
    # bound) over 3000 rows.  The approx mode's error bound is 2^-7 |x||y|: the whole crowd lies inside its listing slack, is listed
    # by the FIRST scan and resolved by the wide re-score -- no second scan, no fp64 scan of the index.  (The three-term operand of
    # rounds 1-3, whose tight key error sends this crowd through the re-scan tier, is a lab build since round 6:
    # tools/experiments/lab_checks_knn.py; the re-scan tier itself stays exercised by tests/test_knn_hostile_gpu.py.)
    y = gaussian(4000, 64, 1)
    c = gaussian(1, 64, 2)
    y[500:3500] = c * (1.0 - 1e-7 * np.arange(3000, dtype=np.float32)[:, None])
    x = np.repeat(c, 4, axis=0)
    for metric in (IP, L2):
        st = _check(metric, x, y, 10)
        assert st["n_rescored"] == 4 and st["n_rescanned"] == 0 and st["n_uncertified"] == 0, st


@pytest.mark.parametrize("metric", [IP, L2])
def test_a_call_cut_into_several_scan_launches(metric, monkeypatch):
    """TRX_QUERY_BATCH (round 5's experiment: one round of workgroups per launch) cuts a call into batches of that many queries,
    each with its own bootstrap, scan and select over the shared workspaces: same answers, batch by batch -- also with the FAISS
    tie rule, whose scratch spans the whole call, and through the host entry point"""
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    monkeypatch.setenv("TRX_QUERY_BATCH", "256")
    st = _check(metric, bf16_round(gaussian(700, 96, 3)), bf16_round(gaussian(9000, 96, 4)), 10)
    assert st["scan_launches"] == 3 and st["n_uncertified"] == 0, st
    _check(metric, gaussian(513, 200, 5), gaussian(4000, 200, 6), 30)           # the two-scan path, fp32 data
    if metric == IP:
        y, x = grid(5000, 24, 7), grid(600, 24, 8)
        idx = faiss.IndexFlatIP(24, tie_rule="faiss"); idx.add(y)
        D, I = idx.search(x, 10)
        Df, If = oracle.knn_faiss(IP, x, y, 10)
        assert np.array_equal(I, If) and np.array_equal(D, Df)


def test_the_bootstrap_bound_changes_tiers_never_answers(monkeypatch):
    """round 5: the bootstrap publishes the 16th largest of a query's 32 tracked maxima instead of the minimum over its lanes'
    second bests (TRX_BOOT_J2=1 keeps the old rule).  A threshold is a hint: on benign data neither rule flags a query, and the
    answers are the oracle's (the crowd that tells the rules' tiers apart needs the lab build's three-term operand:
    tools/experiments/lab_checks_knn.py)"""
    for rule in ("1", None):
        if rule:
            monkeypatch.setenv("TRX_BOOT_J2", rule)
        else:
            monkeypatch.delenv("TRX_BOOT_J2")
        st = _check(IP, bf16_round(gaussian(700, 768, 5678)), bf16_round(gaussian(60000, 768, 1234)), 10)
        assert st["n_rescored"] == 0 and st["n_uncertified"] == 0, st


def test_a_crowd_wider_than_the_lists_still_takes_the_exact_scan():
    # 10,000 rows within the rounding bound of each other: more than a query's lists (and the wide re-score) hold
    y = gaussian(12000, 64, 1)
    c = gaussian(1, 64, 2)
    y[1000:11000] = c * (1.0 - 1e-8 * np.arange(10000, dtype=np.float32)[:, None])
    x = np.repeat(c, 4, axis=0)
    st = _check(IP, x, y, 10)
    assert st["n_uncertified"] == 4, st


def test_dimension_mismatch_raises():
    import textreact_amd.faiss_compat as faiss
    idx = faiss.IndexFlatL2(32)
    with pytest.raises(AssertionError):
        idx.add(np.zeros((4, 31), dtype=np.float32))
    idx.add(np.zeros((4, 32), dtype=np.float32))
    with pytest.raises(AssertionError):
        idx.search(np.zeros((2, 33), dtype=np.float32), 3)


def test_torch_device_path_bf16():
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = bf16_round(gaussian(30000, 768, 1234)); x = bf16_round(gaussian(1000, 768, 5678))
    idx = faiss.IndexFlatIP(768)
    idx.add(torch.from_numpy(y).cuda().bfloat16())
    D, I = idx.search(torch.from_numpy(x).cuda().bfloat16(), 10)
    Dr, Ir = oracle.knn_canonical(IP, x, y, 10)
    assert np.array_equal(I.cpu().numpy(), Ir)
    assert np.array_equal(D.cpu().numpy().view(np.uint32), Dr.view(np.uint32))


@pytest.mark.parametrize("dtype", ["bf16", "f32", "bits"])
def test_device_rows_that_do_not_start_on_16_bytes(dtype):
    """the operand builders and the row statistics read 16 bytes per lane when the rows allow it and element by element when
    they do not: rows that start 2 (bf16) or 4 (fp32) bytes into an allocation give the answers of the aligned rows"""
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    n, nq, d, k = 5000, 300, 64, 10
    if dtype == "bits":
        y = morgan_like(n, d, 5, density=0.2); x = y[:nq].copy(); metric, tdt = L2, torch.bfloat16
    else:
        y = gaussian(n, d, 5); x = gaussian(nq, d, 6); metric = IP
        tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        if dtype == "bf16":
            y, x = bf16_round(y), bf16_round(x)

    def off_by_one(a):
        flat = torch.zeros(a.size + 1, dtype=tdt, device="cuda")
        flat[1:] = torch.from_numpy(a).cuda().to(tdt).reshape(-1)
        v = flat[1:].view(a.shape)
        assert v.is_contiguous() and v.data_ptr() % 16 != 0
        return v

    res = []
    for conv in (lambda a: torch.from_numpy(a).cuda().to(tdt), off_by_one):
        idx = faiss.IndexFlatIP(d) if metric == IP else faiss.IndexFlatL2(d)
        idx.add(conv(y))
        D, I = idx.search(conv(x), k)
        res.append((D.cpu().numpy(), I.cpu().numpy()))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    Dr, Ir = oracle.knn_canonical(metric, x, y, k)
    assert np.array_equal(res[1][1], Ir) and np.array_equal(res[1][0].view(np.uint32), Dr.view(np.uint32))


def test_int_inputs_like_reference():
    # the reference hands faiss int64 / int8 arrays (retrieve_faiss.py:26,39)
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = reaction_fp_like(3000, 512, 3).astype(np.int64)
    idx = faiss.IndexFlatL2(512)
    idx.add(y)
    D, I = idx.search(y[:50].astype(np.int8), 20)
    Dr, Ir = oracle.knn_faiss(L2, y[:50].astype(np.float32), y.astype(np.float32), 20)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    assert (I[:, 0] == np.arange(50)).all() or (D[:, 0] == 0).all()  # self is a distance-0 neighbour


# ---- committed golden vectors through the C ABI -----------------------------------------------
import glob as _glob
import os as _os

_G = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(_glob.glob(_os.path.join(_G, "knn_*.npz"))))
def test_golden_vectors_on_gpu(path):
    z = np.load(path)
    m, k = int(z["metric"]), int(z["k"])
    idx = _index(m, z["y"].shape[1])
    idx.add(z["y"])
    D, I = idx.search(z["x"], k)
    assert np.array_equal(I, z["I_canonical"])
    assert np.array_equal(D.view(np.uint32), z["D_canonical"].view(np.uint32))


def test_merge_kernel_equals_oracle_merge():
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = grid(3000, 64, 3); x = grid(500, 64, 4)
    for metric in (IP, L2):
        parts = np.array_split(y, 5)
        offs = np.cumsum([0] + [len(p) for p in parts[:-1]])
        Sl, Il, Dl = [], [], []
        for part, off in zip(parts, offs):
            idx = _index(metric, 64)
            idx.add(torch.from_numpy(part).cuda())
            D, I, S = idx.search_s64(torch.from_numpy(x).cuda(), 10)
            Sl.append(S); Il.append(torch.where(I >= 0, I + int(off), I)); Dl.append(D)
        D, I = faiss.merge_topk(metric, torch.stack(Sl), torch.stack(Il))
        Dr, Ir = oracle.knn_canonical(metric, x, y, 10)
        assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)
        Dm, Im = oracle.merge_lists(metric, torch.stack(Dl).cpu().numpy(), torch.stack(Il).cpu().numpy())
        assert np.array_equal(Im, Ir)


def test_int8_and_bool_host_arrays_travel_as_bytes_and_give_the_float32_results():
    """the reference's Morgan fingerprints are int8 arrays (retrieve_faiss.py:36-44); faiss' wrapper would convert them to
    float32 on the host -- here they reach the library as bytes (TRX_DTYPE_I8) and are widened on the device: the same values,
    so the oracle's answer on the float32 conversion, bit for bit; negative values and a dimension that is not a multiple of 16"""
    from oracle import flat_knn as oracle
    for d, signed in ((1024, False), (100, True)):
        y = morgan_like(6000, d, 7); x = morgan_like(300, d, 8)
        if signed:
            y = reaction_fp_like(6000, d, 9, density=0.2); x = reaction_fp_like(300, d, 10, density=0.2)
        for metric in (IP, L2):
            Dr, Ir = oracle.knn_canonical(metric, x, y, 20)
            for cast in ((np.int8, np.bool_) if not signed else (np.int8,)):
                idx = _index(metric, d)
                idx.add(y.astype(cast)[:2500]); idx.add(y.astype(cast)[2500:])
                D, I = idx.search(x.astype(cast), 20)
                assert np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), (d, metric, cast)
                assert idx.last_stats()["exact_class"] == 1


def test_host_searches_longer_than_one_block_overlap_the_next_copy_and_keep_their_statistics():
    """trx_index_search takes host queries in blocks of 65,536: block c + 1 crosses PCIe while block c is searched; results in
    place, the statistics of the call are those of all blocks"""
    from oracle import flat_knn as oracle
    y = gaussian(3000, 64, 21); x = gaussian(65536 + 4500, 64, 22)
    for metric in (IP, L2):
        idx = _index(metric, 64)
        idx.add(y)
        D, I = idx.search(x, 10)
        st = idx.last_stats()
        assert st["nq"] == x.shape[0] and st["scan_launches"] >= 2, st
        sel = np.r_[0:300, 65536 - 150:65536 + 150, x.shape[0] - 300:x.shape[0]]
        Dr, Ir = oracle.knn_canonical(metric, x[sel], y, 10)
        assert np.array_equal(I[sel], Ir) and np.array_equal(D[sel].view(np.uint32), Dr.view(np.uint32))
        x64 = np.rint(x * 3).astype(np.int64); y64 = np.rint(y * 3).astype(np.int64)      # int64 (the difference fingerprints): the
        idx64 = _index(metric, 64); idx64.add(y64)                           # float32 conversion of block c + 1 runs beside block c
        D64, I64 = idx64.search(x64, 10)
        assert idx64.last_stats()["nq"] == x.shape[0]
        Dr, Ir = oracle.knn_canonical(metric, x64[sel].astype(np.float32), y64.astype(np.float32), 10)
        assert np.array_equal(I64[sel], Ir) and np.array_equal(D64[sel].view(np.uint32), Dr.view(np.uint32))
        x8 = np.clip(np.rint(x * 3), -100, 100).astype(np.int8)           # the same through the int8 transport
        idx8 = _index(metric, 64); idx8.add(np.clip(np.rint(y * 3), -100, 100).astype(np.int8))
        D8, I8 = idx8.search(x8, 10)
        Dr, Ir = oracle.knn_canonical(metric, x8[sel].astype(np.float32), np.clip(np.rint(y * 3), -100, 100).astype(np.float32), 10)
        assert np.array_equal(I8[sel], Ir) and np.array_equal(D8[sel].view(np.uint32), Dr.view(np.uint32))


def test_single_rank_sharded_wrapper():
    import torch
    from textreact_amd.sharded import ShardedFlatIndex
    from oracle import flat_knn as oracle
    y = gaussian(5000, 64, 1); x = gaussian(64, 64, 2)
    idx = ShardedFlatIndex(64, IP)
    idx.add_shard(torch.from_numpy(y).cuda(), 100, 5100)     # offset is applied to the ids
    D, I = idx.search(torch.from_numpy(x).cuda(), 10)
    Dr, Ir = oracle.knn_canonical(IP, x, y, 10)
    assert np.array_equal(I.cpu().numpy(), Ir + 100) and np.array_equal(D.cpu().numpy(), Dr)


def test_sharded_wrapper_over_fingerprints_runs_the_int8_form():
    """the stream-ordered begin / finish path of a shard (what ShardedFlatIndex drives) with the integer class: int8 scan, bf16
    device tensors in, ids offset by the shard's first row"""
    import torch
    from textreact_amd.sharded import ShardedFlatIndex
    from oracle import flat_knn as oracle
    y = reaction_fp_like(9000, 1024, 41); x = y[4000:4300]
    idx = ShardedFlatIndex(1024, L2)
    idx.add_shard(torch.from_numpy(y).cuda().to(torch.bfloat16), 700, 9700)
    D, I = idx.search(torch.from_numpy(x).cuda().to(torch.bfloat16), 20)
    Dr, Ir = oracle.knn_canonical(L2, x, y, 20)
    assert np.array_equal(I.cpu().numpy(), Ir + 700) and np.array_equal(D.cpu().numpy(), Dr)
    assert idx.local.last_stats()["int8_scan"] == 1


def test_cli_on_the_real_index(tmp_path):
    import json
    import pandas as pd
    import textreact_amd.retrieve_faiss as rf
    from oracle import flat_knn as oracle
    fps = reaction_fp_like(400, 2048, 3)
    pd.DataFrame({"id": np.arange(300), "canonical_rxn": ["C>>C"] * 300}).to_csv(tmp_path / "train.csv", index=False)
    pd.DataFrame({"id": np.arange(50) + 1000, "canonical_rxn": ["C>>C"] * 50}).to_csv(tmp_path / "val.csv", index=False)
    pd.DataFrame({"id": np.arange(50) + 2000, "canonical_rxn": ["C>>C"] * 50}).to_csv(tmp_path / "test.csv", index=False)
    for name, sl in (("train", slice(0, 300)), ("val", slice(300, 350)), ("test", slice(350, 400))):
        np.save(tmp_path / (name + ".npy"), fps[sl].astype(np.int64))     # int64 like the reference's arrays
    rf.main(["--data_path", str(tmp_path), "--train_file", "train.csv", "--valid_file", "val.csv", "--test_file",
             "test.csv", "--output_path", str(tmp_path / "out"), "--train_vectors", str(tmp_path / "train.npy"),
             "--valid_vectors", str(tmp_path / "val.npy"), "--test_vectors", str(tmp_path / "test.npy")])
    got = json.loads((tmp_path / "out" / "test.json").read_text())
    _, I = oracle.knn_faiss(L2, fps[350:], fps[:300], 20)           # the literal FAISS restatement
    assert [e["nn"] for e in got] == I.tolist() and [e["id"] for e in got] == list(range(2000, 2050))


def test_cli_tie_rule_flag_gives_the_order_of_the_faiss_heap(tmp_path, monkeypatch):
    """--metric ip over sparse integer count vectors: integer scores, ties in every list.  --tie_rule faiss writes the neighbour lists the
    literal restatement of FAISS's heap (oracle.knn_faiss) returns; without the flag ties go smaller id first"""
    import json
    import pandas as pd
    import textreact_amd.retrieve_faiss as rf
    from oracle import flat_knn as oracle
    monkeypatch.setenv("TRX_TIE_RULE", "id")                    # main() sets it; monkeypatch puts the environment back
    fps = reaction_fp_like(700, 256, 9)
    pd.DataFrame({"id": np.arange(600), "canonical_rxn": ["C>>C"] * 600}).to_csv(tmp_path / "train.csv", index=False)
    pd.DataFrame({"id": np.arange(50) + 1000, "canonical_rxn": ["C>>C"] * 50}).to_csv(tmp_path / "val.csv", index=False)
    pd.DataFrame({"id": np.arange(50) + 2000, "canonical_rxn": ["C>>C"] * 50}).to_csv(tmp_path / "test.csv", index=False)
    for name, sl in (("train", slice(0, 600)), ("val", slice(600, 650)), ("test", slice(650, 700))):
        np.save(tmp_path / (name + ".npy"), fps[sl].astype(np.float32))
    argv = ["--data_path", str(tmp_path), "--train_file", "train.csv", "--valid_file", "val.csv", "--test_file", "test.csv",
            "--train_vectors", str(tmp_path / "train.npy"), "--valid_vectors", str(tmp_path / "val.npy"),
            "--test_vectors", str(tmp_path / "test.npy"), "--metric", "ip"]
    rf.main(argv + ["--output_path", str(tmp_path / "faiss"), "--tie_rule", "faiss"])
    rf.main(argv + ["--output_path", str(tmp_path / "id"), "--tie_rule", "id"])
    got_f = [e["nn"] for e in json.loads((tmp_path / "faiss" / "test.json").read_text())]
    got_i = [e["nn"] for e in json.loads((tmp_path / "id" / "test.json").read_text())]
    _, If = oracle.knn_faiss(IP, fps[650:], fps[:600], 20)
    _, Ic = oracle.knn_canonical(IP, fps[650:], fps[:600], 20)
    assert got_f == If.tolist() and got_i == Ic.tolist()
    assert got_f != got_i                                        # the inputs do tell the two rules apart


# ---- BASELINE.json configs[1] at full size: size-independent properties + a sampled oracle check
def test_c1_full_size_properties():
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    n, d, nq, k = 1_000_000, 768, 65_536, 10
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    y = torch.randn((n, d), generator=g, device="cuda").bfloat16()
    x = torch.randn((nq, d), generator=g, device="cuda").bfloat16()
    x[:4096] = y[::244][:4096]                       # queries that ARE corpus rows
    for metric, cls in ((IP, faiss.IndexFlatIP), (L2, faiss.IndexFlatL2)):
        idx = cls(d)
        idx.add(y)
        D, I = idx.search(x, k)
        st = idx.last_stats()
        assert st["n_uncertified"] == 0, st        # every query certified exact by the fast path on the benchmark inputs
        Dh, Ih = D.cpu().numpy(), I.cpu().numpy()
        assert (Ih >= 0).all() and (Ih < n).all()
        assert all(len(set(r)) == k for r in Ih[:2000].tolist())                       # no duplicates
        dd = np.diff(Dh, axis=1)
        assert (dd >= 0).all() if metric == L2 else (dd <= 0).all()                    # best first
        if metric == L2:
            assert (Dh[:4096, 0] == 0).all() and (Ih[:4096, 0] == np.arange(4096) * 244).all()   # self at distance 0
        # idempotence: the same search again returns the same bits
        D2, I2 = idx.search(x, k)
        assert torch.equal(I, I2) and torch.equal(D, D2)
        # sampled exact check against the oracle (64 queries x the full corpus)
        sel = np.r_[0:8, 4096:4104, np.random.default_rng(0).integers(0, nq, 48)]
        xs = x[torch.from_numpy(sel).cuda()].float().cpu().numpy()
        Dr, Ir = oracle.knn_canonical(metric, xs, y.float().cpu().numpy(), k)
        assert np.array_equal(Ih[sel], Ir) and np.array_equal(Dh[sel].view(np.uint32), Dr.view(np.uint32))
        del idx


def test_bit_vectors_at_size_through_the_fp4_form():
    """retrieve/retro.sh at USPTO-full scale: 500,000 x 1024 Morgan-like bit vectors, the first 70,000 searching the set (two
    query batches), L2, k = 20, on fp4 operands -- every row finds itself at distance 0, results are sorted, idempotent, equal
    to the int8 form's, and 64 sampled queries equal the oracle over the FULL corpus bit for bit"""
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    n, d, nq, k = 500_000, 1024, 70_000, 20
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    y = (torch.rand((n, d), generator=g, device="cuda") < 0.05).to(torch.bfloat16)
    idx = faiss.IndexFlatL2(d); idx.add(y)
    D, I = idx.search(y[:nq], k)
    st = idx.last_stats()
    assert st["int8_scan"] == 2 and st["exact_class"] == 1 and st["n_uncertified"] == 0, st
    Dh, Ih = D.cpu().numpy(), I.cpu().numpy()
    assert (Dh[:, 0] == 0).all() and (np.diff(Dh, axis=1) >= 0).all() and (Ih >= 0).all() and (Ih < n).all()
    assert (Ih[:, 0] <= np.arange(nq)).all()                      # self, or an identical row with a smaller number
    D2, I2 = idx.search(y[:nq], k)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    os.environ["TRX_NO_FP4"] = "1"
    try:
        D8, I8 = idx.search(y[:nq], k)
        assert idx.last_stats()["int8_scan"] == 1
    finally:
        del os.environ["TRX_NO_FP4"]
    assert torch.equal(I, I8) and torch.equal(D, D8)
    sel = np.r_[0:8, 65530:65546, np.random.default_rng(0).integers(0, nq, 40)]
    Dr, Ir = oracle.knn_canonical(L2, y[torch.from_numpy(sel).cuda()].float().cpu().numpy(), y.float().cpu().numpy(), k)
    assert np.array_equal(Ih[sel], Ir) and np.array_equal(Dh[sel].view(np.uint32), Dr.view(np.uint32))


def test_more_than_one_query_batch_and_many_splits():
    # 70,000 queries -> two internal batches (65,536 + 4,464); small corpus -> many corpus splits
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = gaussian(3000, 64, 1)
    x = gaussian(70000, 64, 2)
    idx = faiss.IndexFlatIP(64)
    idx.add(y)
    D, I = idx.search(x, 5)
    sel = np.r_[0:50, 65500:65600, 69950:70000]
    Dr, Ir = oracle.knn_canonical(IP, x[sel], y, 5)
    assert np.array_equal(I[sel], Ir) and np.array_equal(D[sel].view(np.uint32), Dr.view(np.uint32))
    # a few hundred queries against a corpus of 100k rows: nsplits > 1, uneven last split
    y2 = gaussian(100_003, 96, 3); x2 = gaussian(300, 96, 4)
    for metric in (IP, L2):
        st = _check(metric, x2, y2, 10)
        assert st["n_splits"] > 1



def test_the_split_count_fills_whole_rounds_of_workgroups():
    """knn_api.hip: choose_splits -- 18 query tiles take 14 corpus splits (252 workgroups, one round of the 256 CUs), not the
    ceil(256 / 18) = 15 of the old rule (270: a second round for 14 workgroups); 50 tiles take 5 (250), not 6; a full batch of 256
    tiles keeps 4; the answers are the oracle's whatever the count"""
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    y = torch.randn((200_000, 768), generator=g, device="cuda").bfloat16()
    idx = faiss.IndexFlatIP(768); idx.add(y)
    yh = y.float().cpu().numpy()
    for nq, want in ((4464, 14), (12800, 5), (65536, 4)):
        x = torch.randn((nq, 768), generator=g, device="cuda").bfloat16()
        D, I = idx.search(x, 10)
        st = idx.last_stats()
        assert st["n_splits"] == want and st["n_uncertified"] == 0, st
        sel = np.random.default_rng(nq).integers(0, nq, 40)
        Dr, Ir = oracle.knn_canonical(IP, x[torch.from_numpy(sel).cuda()].float().cpu().numpy(), yh, 10)
        assert np.array_equal(I.cpu().numpy()[sel], Ir) and np.array_equal(D.cpu().numpy()[sel].view(np.uint32), Dr.view(np.uint32))


@pytest.mark.parametrize("nsplits", [1, 3, 4, 5, 8, 13])
def test_shared_thresholds_with_any_number_of_splits(nsplits, monkeypatch):
    """the splits of a query share thresholds through four slots (slot = split & 3, knn_scan.hip): exact for split counts
    below, at and above four and not a multiple of it, for both threshold depths (k <= 12: k' = 16; k = 20: k' = 32),
    on a corpus long enough (782 tiles) for the shared bound to take over from each split's own"""
    monkeypatch.setenv("TRX_NSPLITS", str(nsplits))
    y = gaussian(200_000, 64, 11); x = gaussian(520, 64, 12)
    for metric, k in ((IP, 10), (L2, 20)):
        st = _check(metric, x, y, k)
        assert st["n_splits"] == min(nsplits, 782) and st["n_uncertified"] == 0


def test_device_tensors_fp32_and_reset():
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = gaussian(4000, 80, 5); x = gaussian(60, 80, 6)
    idx = faiss.IndexFlatL2(80)
    idx.add(torch.from_numpy(y).cuda())
    D, I = idx.search(torch.from_numpy(x).cuda(), 7)
    Dr, Ir = oracle.knn_canonical(L2, x, y, 7)
    assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)
    idx.reset()
    assert idx.ntotal == 0
    idx.add(y[:10])
    D, I = idx.search(x[:3], 12)
    assert (I[:, 10:] == -1).all() and (I[:, :10] >= 0).all()


def test_values_outside_bf16_range_and_tiny_values():
    # large magnitudes and subnormal-ish small values through the split operand
    rng = np.random.default_rng(9)
    y = (rng.standard_normal((3000, 48)) * np.exp(rng.uniform(-20, 20, (3000, 1)))).astype(np.float32)
    x = rng.standard_normal((64, 48)).astype(np.float32)
    _check(IP, x, y, 10)
    _check(L2, x, y, 10)


# ---- N > 1 on one device: every rank drives the HIP index and the HIP merge kernel on cuda:0; only the
# transport differs from the real run (gloo instead of RCCL, since two ranks cannot share a GPU in RCCL)
def _rank_worker(rank, world, port, ret):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from textreact_amd.sharded import ShardedFlatIndex, shard_bounds
    from _data import gaussian as g_
    n, d, nq = 30011, 96, 300
    y, x = g_(n, d, 11), g_(nq, d, 12)
    y[5000:5040] = y[100]                                   # ties that straddle the shard boundaries
    y[20000:20020] = y[100]
    out = {}
    for metric in (0, 1):
        lo, hi = shard_bounds(n, world, rank)
        idx = ShardedFlatIndex(d, metric)
        idx.add_shard(torch.from_numpy(y[lo:hi]).cuda(), lo, n)
        D, I = idx.search(torch.from_numpy(x).cuda(), 10)
        out[metric] = (D.cpu().numpy(), I.cpu().numpy())
        D, I = idx.search(torch.from_numpy(x[:64]).cuda(), 100)      # the two-scan path's fp64 scores through the exchange and the merge
        out[("k100", metric)] = (D.cpu().numpy(), I.cpu().numpy())
        # the other decomposition (FAISS' IndexReplicas): all rows on every rank, a G-th of the queries each
        from textreact_amd.sharded import ReplicatedFlatIndex
        rep = ReplicatedFlatIndex(d, metric)
        rep.add(torch.from_numpy(y).cuda())
        D, I = rep.search(torch.from_numpy(x).cuda(), 10)
        out[("replicas", metric)] = (D.cpu().numpy(), I.cpu().numpy())
        if world == 2:      # round 6: the exchange on finished lists (no agreement collective): the same answer
            fidx = ShardedFlatIndex(d, metric, stream_ordered=False)
            fidx.add_shard(torch.from_numpy(y[lo:hi]).cuda(), lo, n)
            D, I = fidx.search(torch.from_numpy(x).cuda(), 10)
            out[("finish_first", metric)] = (D.cpu().numpy(), I.cpu().numpy())
        if world == 4:      # round 6: the rows x queries grid, 2 x 2 -- row shard r % 2, query slice r // 2, the exchange inside a column
            gidx = ShardedFlatIndex(d, metric, row_groups=2)
            glo, ghi = gidx.shard_rows(n)
            gidx.add_shard(torch.from_numpy(y[glo:ghi]).cuda(), glo, n)
            D, I = gidx.search(torch.from_numpy(x).cuda(), 10)
            out[("grid", metric)] = (D.cpu().numpy(), I.cpu().numpy())
    # FAISS' own order among exact inner-product ties, through the shards: canonical top 2k per shard, merged, the rule once
    import textreact_amd.faiss_compat as fc
    from _data import grid as grid_
    yt, xt = grid_(20003, 32, 31), grid_(200, 32, 32)
    yt[7000:7030] = yt[3]; yt[15000:15025] = yt[3]; xt[0] = yt[3]
    lo, hi = shard_bounds(len(yt), world, rank)
    idx = ShardedFlatIndex(32, 0, local_index=fc.IndexFlatIP(32, tie_rule="faiss"))
    idx.add_shard(torch.from_numpy(yt[lo:hi]).cuda(), lo, len(yt))
    D, I = idx.search(torch.from_numpy(xt).cuda(), 10)
    out["faiss_ties"] = (D.cpu().numpy(), I.cpu().numpy())
    ret[rank] = out
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_multi_rank_sharded_search_on_one_gpu(world):
    import socket
    import torch.multiprocessing as mp
    from oracle import flat_knn as oracle
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_rank_worker, args=(world, port, ret), nprocs=world, join=True)
    n, d, nq = 30011, 96, 300
    y, x = gaussian(n, d, 11), gaussian(nq, d, 12)
    y[5000:5040] = y[100]; y[20000:20020] = y[100]
    for metric in (IP, L2):
        Dr, Ir = oracle.knn_canonical(metric, x, y, 10)
        for r in range(world):
            for key in (metric, ("replicas", metric)) + ((("grid", metric),) if world == 4 else ()) + ((("finish_first", metric),) if world == 2 else ()):
                D, I = ret[r][key]
                assert np.array_equal(I, Ir), (key, r)
                assert np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), (key, r)
        Dr, Ir = oracle.knn_canonical(metric, x[:64], y, 100)
        for r in range(world):
            D, I = ret[r][("k100", metric)]
            assert np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), ("k100", metric, r)
    yt, xt = grid(20003, 32, 31), grid(200, 32, 32)
    yt[7000:7030] = yt[3]; yt[15000:15025] = yt[3]; xt[0] = yt[3]
    Df, If = oracle.knn_faiss(IP, xt, yt, 10)
    assert not np.array_equal(If, oracle.knn_canonical(IP, xt, yt, 10)[1])       # the input does tell the two rules apart
    for r in range(world):
        D, I = ret[r]["faiss_ties"]
        assert np.array_equal(I, If) and np.array_equal(D.view(np.uint32), Df.view(np.uint32)), ("faiss_ties", r)


def test_pad_queries_of_the_last_query_tile_cost_nothing():
    """round 4: a query count that is not a multiple of 256 is padded with zero rows; under the inner product every corpus
    row scores exactly 0 against a zero query -- one tie group of the whole corpus -- and the pad columns' lists used to fill
    and be compacted at every tile: 40,000 queries took 190 ms where 40,192 took 12.  The scan never lists for pad columns now."""
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    y = torch.randn((204800, 768), generator=g, device="cuda").bfloat16()
    x = torch.randn((40192, 768), generator=g, device="cuda").bfloat16()
    idx = faiss.IndexFlatIP(768); idx.add(y); idx.set_timing(True)
    ms = {}
    for nq in (40192, 40000, 40192, 40000):
        D, I = idx.search(x[:nq], 10)
        ms[nq] = idx.last_stats()["scan_ms"]
    assert ms[40000] < 1.5 * ms[40192], ms
    rows = [0, 1, 39935, 39936, 39999]              # first tile, and the real queries that share the last tile with the pad rows
    _, want = oracle.knn_canonical(IP, x[rows].float().cpu().numpy(), y.float().cpu().numpy(), 10)
    assert np.array_equal(I[rows].cpu().numpy(), want)


def test_indexes_share_one_workspace_across_streams():
    """round 4: the large search workspaces are one set per device, shared by every index of the process; a search enqueues
    behind the event the previous user recorded.  Two indexes of different shapes, searched on two different streams,
    interleaved and with one search begun and not yet finished while the other index runs: every result is the oracle's"""
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    ya, xa = gaussian(30000, 96, 1), gaussian(700, 96, 2)
    yb, xb = bf16_round(gaussian(8000, 768, 3)), bf16_round(gaussian(300, 768, 4))
    a, b = faiss.IndexFlatIP(96), faiss.IndexFlatL2(768)
    a.add(ya); b.add(yb)
    xa_d, xb_d = torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda().bfloat16()
    Da_r, Ia_r = oracle.knn_canonical(IP, xa, ya, 10)
    Db_r, Ib_r = oracle.knn_canonical(L2, xb, yb, 10)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(4):
        with torch.cuda.stream(s1):
            Da, Ia, Sa = a.search_s64_begin(xa_d, 10)            # enqueued on s1, not waited for
        with torch.cuda.stream(s2):
            Db, Ib = b.search(xb_d, 10)                          # the other index, another stream, the same workspaces
        a.search_finish()
        assert np.array_equal(Ia.cpu().numpy(), Ia_r) and np.array_equal(Da.cpu().numpy().view(np.uint32), Da_r.view(np.uint32))
        assert np.array_equal(Ib.cpu().numpy(), Ib_r) and np.array_equal(Db.cpu().numpy().view(np.uint32), Db_r.view(np.uint32))
    del a
    D2, I2 = b.search(xb_d, 10)                                  # the surviving index keeps the workspaces
    assert np.array_equal(I2.cpu().numpy(), Ib_r)


def _late_worker(rank, world, port, ret):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from textreact_amd.sharded import ShardedFlatIndex, shard_bounds
    y, x = _near_duplicate_problem()
    lo, hi = shard_bounds(len(y), world, rank)
    idx = ShardedFlatIndex(y.shape[1], 0)
    idx.add_shard(torch.from_numpy(y[lo:hi]).cuda(), lo, len(y))
    D, I = idx.search(torch.from_numpy(x).cuda(), 10)
    ret[rank] = (D.cpu().numpy(), I.cpu().numpy(), idx.late_fallbacks, idx.local.last_stats()["n_uncertified"])
    dist.destroy_process_group()


def _near_duplicate_problem():
    """8 crowds of 5,000 rows in the first shard, each within the certificate's slack of its query (float32 steps of 1e-8:
    many exact duplicates among them): more rows than a query's candidate lists or the wide re-score hold, so neither the 32
    re-scored candidates, nor the wide re-score, nor the fixed-threshold re-scan can prove the top 10, and those 8 queries take
    the exact scan -- more of them than the 4 inline slots of a batch (csrc/knn_api.hip INLINE_FALLBACK).  (Smaller crowds are
    resolved on the way: test_near_duplicate_cluster_is_resolved_by_the_wide_rescore, test_a_crowd_..._by_the_rescan.)"""
    rng = np.random.default_rng(21)
    y = rng.standard_normal((100000, 64)).astype(np.float32)
    x = rng.standard_normal((40, 64)).astype(np.float32)
    fall = (1.0 - 1e-8 * np.arange(5000, dtype=np.float32))[:, None]
    for c in range(8):
        base = 3.0 * rng.standard_normal(64).astype(np.float32)
        y[5000 * c:5000 * (c + 1)] = base[None] * fall
        x[c] = base
    return y, x


def test_late_fallback_repeats_the_exchange():
    """more certificate failures in a batch than the inline slots -> trx_index_search_finish re-does queries AFTER
    the exchange was enqueued; the one-word all-reduce makes every rank repeat the exchange.  Two ranks on one GPU (gloo);
    the result must still be the unsharded oracle's, on both ranks -- also on the rank whose own shard had no failure."""
    import socket
    import torch.multiprocessing as mp
    from oracle import flat_knn as oracle
    y, x = _near_duplicate_problem()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_late_worker, args=(2, port, ret), nprocs=2, join=True)
    Dr, Ir = oracle.knn_canonical(IP, x, y, 10)
    assert ret[0][2] == 1 and ret[0][3] > 4, "the problem no longer reaches the late path: %r" % (ret[0][2:],)
    assert ret[1][2] == 0                                   # rank 1 had nothing to re-do and repeated the exchange all the same
    for r in (0, 1):
        assert np.array_equal(ret[r][1], Ir), r
        assert np.array_equal(ret[r][0].view(np.uint32), Dr.view(np.uint32)), r


def test_sharded_search_over_a_one_rank_rccl_group():
    """the transport branch of sharded.py / live.py that only RCCL takes (device tensors through all_to_all_single with split
    sizes, all_gather_into_tensor, the int32 MAX all-reduce of the late flag), on the one GPU a test box has: a one-rank nccl
    group with the exchange forced on.  What two ranks would add -- uneven splits -- runs over gloo in the tests above."""
    import torch
    import torch.distributed as dist
    from oracle import flat_knn as oracle
    from textreact_amd.sharded import ShardedFlatIndex
    from textreact_amd.live import all_gather_rows
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    assert not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for (y, x) in ((gaussian(5000, 64, 1), gaussian(301, 64, 2)), _near_duplicate_problem()):
            for metric in (IP, L2):
                idx = ShardedFlatIndex(64, metric, exchange_always=True)
                idx.add_shard(torch.from_numpy(y).cuda(), 100, len(y) + 100)
                D, I = idx.search(torch.from_numpy(x).cuda(), 10)
                Dr, Ir = oracle.knn_canonical(metric, x, y, 10)
                assert np.array_equal(I.cpu().numpy(), Ir + 100) and np.array_equal(D.cpu().numpy().view(np.uint32), Dr.view(np.uint32))
        assert idx.late_fallbacks == 1                     # the near-duplicate problem took the repeated exchange over RCCL too (L2)
        e = torch.randn(37, 768, device="cuda").bfloat16()
        assert torch.equal(all_gather_rows(e, 37, 0, 1, always=True), e)
    finally:
        dist.destroy_process_group()


def test_sharded_cli_writes_the_files_one_gpu_writes(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 -m textreact_amd.retrieve_faiss ...: the train vectors row-sharded
    over two ranks (both on this box's GPU, gloo transport), the HIP index and HIP merge on each; rank 0's train / val /
    test.json are byte for byte the files of the single-GPU run (retrieve/retrieve_faiss.py:112-130)"""
    import subprocess
    import pandas as pd
    import textreact_amd.retrieve_faiss as rf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fps = reaction_fp_like(1101, 2048, 3)
    fps[700:720] = fps[3]                      # ties that straddle the shard boundary at row 501
    fps[40:45] = fps[3]
    pd.DataFrame({"id": np.arange(1001), "canonical_rxn": ["C>>C"] * 1001, "year": 2000 + np.arange(1001) % 20}).to_csv(tmp_path / "train.csv", index=False)
    pd.DataFrame({"id": np.arange(50) + 5000, "canonical_rxn": ["C>>C"] * 50}).to_csv(tmp_path / "val.csv", index=False)
    pd.DataFrame({"id": np.arange(50) + 6000, "canonical_rxn": ["C>>C"] * 50}).to_csv(tmp_path / "test.csv", index=False)
    for name, sl in (("train", slice(0, 1001)), ("val", slice(1001, 1051)), ("test", slice(1051, 1101))):
        np.save(tmp_path / (name + ".npy"), fps[sl].astype(np.int64))
    for case, extra in enumerate(([], ["--before", "2013"], ["--replicas"])):
        if case == 2:      # int8 arrays (what morgan_fingerprint returns): bytes over PCIe, widened on the device; and the replicated form
            for name, sl in (("train", slice(0, 1001)), ("val", slice(1001, 1051)), ("test", slice(1051, 1101))):
                np.save(tmp_path / (name + ".npy"), fps[sl].astype(np.int8))
        argv = ["--data_path", str(tmp_path), "--train_file", "train.csv", "--valid_file", "val.csv", "--test_file", "test.csv",
                "--train_vectors", str(tmp_path / "train.npy"), "--valid_vectors", str(tmp_path / "val.npy"),
                "--test_vectors", str(tmp_path / "test.npy")] + extra
        one, two = tmp_path / ("one%d" % case), tmp_path / ("two%d" % case)
        assert rf.main(argv + ["--output_path", str(one)]) == 0
        env = dict(os.environ, TRX_DIST_BACKEND="gloo", TRX_DEVICE="0")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29731", "-m", "textreact_amd.retrieve_faiss"] + argv + ["--output_path", str(two)],
                           cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        for name in ("train.json", "val.json", "test.json"):
            assert (two / name).read_bytes() == (one / name).read_bytes(), name
        assert r.stdout.count("Faiss nearest neighbor search") == 3        # rank 0 alone prints


# ---- the literal FAISS restatement (and FAISS itself when a box has it) on Gaussian C0 -----------------------------
def _audit_near_ties(metric, x, y, I_hip, I_ref, what):
    """Every position where the HIP result and an fp32-accumulating reference differ must be a provable near-tie:
    the two rows' canonical fp64 scores differ by no more than fp32 rounding of the accumulation can explain
    (the same relation tests/test_oracle.py::test_gaussian_differences_are_provable_near_ties proves CPU-side)."""
    from oracle import flat_knn as oracle
    bad = np.argwhere(I_hip != I_ref)
    if bad.size:
        Sh = oracle.scores_at(metric, x, y, I_hip)
        Sr = oracle.scores_at(metric, x, y, I_ref)
        for q, t in bad:
            assert abs(Sh[q, t] - Sr[q, t]) <= 64 * np.finfo(np.float32).eps * max(1.0, abs(Sr[q, t])), (what, q, t)
        # and the two answers are the same SET except where the k-th and (k+1)-th straddle such a tie
        assert bad.shape[0] <= I_ref.size // 200, (what, bad.shape[0])
    return int(bad.shape[0])


@pytest.mark.parametrize("metric", [IP, L2])
def test_c0_gaussian_vs_literal_faiss_restatement(metric):
    # BASELINE.json configs[0] (10k x 768 fp32, 1k queries, top-10) against the oracle's literal FAISS statement:
    # 4096 x 1024 fp32 blocks (plain loops and host-BLAS sgemm), strict-admission (value, id) heap, heap_reorder
    from oracle import flat_knn as oracle
    x, y, k = gaussian(1000, 768, 5678), gaussian(10000, 768, 1234), 10
    idx = _index(metric, 768)
    idx.add(y)
    D, I = idx.search(x, k)
    for name, fn in (("knn_faiss", oracle.knn_faiss), ("knn_faiss_blas", oracle.knn_faiss_blas)):
        Df, If = fn(metric, x, y, k)
        _audit_near_ties(metric, x, y, I, If, name)
        assert np.allclose(D, Df, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("metric", [IP, L2])
def test_real_faiss_when_the_box_has_it(metric):
    # FAISS is absent from the build container; if a GPU box ships it, it IS the reference
    # (retrieve/retrieve_faiss.py:14,65-71) and this test pins the product to it; otherwise the skip says so.
    faiss = pytest.importorskip("faiss", reason="faiss: unavailable on this box (parity stays pinned to the restatement only)")
    print("faiss version", getattr(faiss, "__version__", "?"))
    x, y, k = gaussian(1000, 768, 5678), gaussian(10000, 768, 1234), 10
    ref = faiss.IndexFlatIP(768) if metric == IP else faiss.IndexFlatL2(768)
    ref.add(y)
    Df, If = ref.search(x, k)
    idx = _index(metric, 768)
    idx.add(y)
    D, I = idx.search(x, k)
    _audit_near_ties(metric, x, y, I, If, "faiss " + getattr(faiss, "__version__", "?"))
    # integer fingerprints (the reference's real inputs): exact arithmetic on both sides -> identical, ties included (L2)
    fy = reaction_fp_like(5000, 2048, 3); fx = fy[:200].copy()
    ref = faiss.IndexFlatL2(2048); ref.add(fy)
    Df, If = ref.search(fx, 20)
    idx = _index(L2, 2048); idx.add(fy)
    D, I = idx.search(fx, 20)
    assert np.array_equal(I, If) and np.array_equal(D, Df)


def test_hip_search_against_an_independent_brute_force_library():
    """no oracle in the loop: the HIP index against scikit-learn's NearestNeighbors(algorithm='brute') in float64 on C0-sized
    Gaussian data (10,000 x 768, 1,000 queries, k = 10) -- same neighbours in the same order, squared distances to fp32 rounding"""
    sk = pytest.importorskip("sklearn.neighbors")
    x, y = gaussian(1000, 768, 5678), gaussian(10000, 768, 1234)
    dist, ind = sk.NearestNeighbors(n_neighbors=10, algorithm="brute", metric="euclidean").fit(y.astype(np.float64)).kneighbors(x.astype(np.float64))
    idx = _index(L2, 768)
    idx.add(y)
    D, I = idx.search(x, 10)
    assert np.array_equal(I, ind)
    assert np.allclose(D, dist ** 2, rtol=2e-5, atol=1e-3)
