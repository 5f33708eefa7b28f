"""Hostile inputs through the C ABI against the oracle: non-finite components (NaN, +inf, -inf) in corpus rows and in queries,
an all-NaN query, fewer than k rows with a finite score, magnitudes whose scores overflow float32, k beyond the limits, a search
on an empty index, an add after a search, d = 1.  A NaN / +-inf row in a corpus of embeddings is the most likely bad input a
real run of retrieve/retrieve_faiss.py:62-74 meets.

Expected values: oracle.knn_canonical (oracle/flat_knn_ref.c: a NaN score never ranks; +-inf scores rank like any other value;
D = (float) of the fp64 score, so a finite fp64 score beyond float32 comes back as +-inf).  What FAISS itself does in each
case is listed in INTEGRATION.md ("Non-finite inputs").  Each test names the fall-back tier it reaches."""
import numpy as np
import pytest

from _data import bf16_round, gaussian, reaction_fp_like

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

IP, L2 = 0, 1
NAN, INF = np.float32(np.nan), np.float32(np.inf)


def _index(metric, d):
    import textreact_amd.faiss_compat as faiss
    return faiss.IndexFlatIP(d) if metric == IP else faiss.IndexFlatL2(d)


def _same(D, I, Dr, Ir, what=""):
    bad = np.argwhere(I != Ir)
    assert bad.size == 0, "%s: ids differ at %r: got %r want %r" % (what, bad[:4].tolist(), I[bad[0][0]].tolist(), Ir[bad[0][0]].tolist())
    assert np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), "%s: distance bits differ" % what


def _run(metric, x, y, k, chunks=1):
    from oracle import flat_knn as oracle
    idx = _index(metric, y.shape[1])
    for part in np.array_split(y, chunks):
        idx.add(part)
    D, I = idx.search(x, k)
    with np.errstate(all="ignore"):
        Dr, Ir = oracle.knn_canonical(metric, x, y, k)
    return D, I, Dr, Ir, idx.last_stats()


def _poison(y, x):
    """corpus rows and queries with non-finite components, in place: a NaN component, +inf, -inf, a whole row of NaN, a row of
    float32's largest values (its scores are finite in fp64 and infinite as float32)"""
    big = np.float32(3.0e38)
    y[5, 7] = NAN; y[17, 3] = INF; y[29, 11] = -INF; y[41, :] = NAN; y[53, :] = big; y[54, :] = -big
    y[2000, 0] = NAN; y[2001, -1] = INF
    x[3, 2] = NAN; x[7, 5] = INF; x[11, 9] = -INF; x[13, :] = NAN; x[19, :] = big
    return y, x


@pytest.mark.parametrize("metric", [IP, L2])
@pytest.mark.parametrize("kind", ["fp32", "bf16_values", "integers"])
def test_non_finite_components_in_rows_and_queries(metric, kind):
    """fp32 data (approximate operand + certificate), bf16-exact data (plain operand), small integers (the exact class until a
    non-finite value joins them): every query's neighbours and distances are the oracle's, the all-NaN query gets k pads"""
    if kind == "integers":
        y = reaction_fp_like(3000, 64, 3).astype(np.float32); x = reaction_fp_like(200, 64, 4).astype(np.float32)
    else:
        y, x = gaussian(3000, 64, 1), gaussian(200, 64, 2)
        if kind == "bf16_values":
            y, x = bf16_round(y), bf16_round(x)
    y, x = _poison(y, x)
    if kind == "bf16_values":
        y[53, :] = bf16_round(np.full(64, 3.0e38, np.float32)); y[54, :] = -y[53, :]; x[19, :] = y[53, :]
    D, I, Dr, Ir, st = _run(metric, x, y, 10, chunks=3)
    _same(D, I, Dr, Ir, "%s %s" % (kind, "IP" if metric == IP else "L2"))
    assert (I[13] == -1).all()                                            # the all-NaN query: nothing ranks
    assert st["nq"] == 200
    # tiers: queries whose certificate cannot hold (a non-finite key or bound) go down to the exact fp64 scan; the rest stay
    # on the fast path
    # tiers: the five hostile queries (3, 7, 11, 13, 19) are never certified and go down every tier to the
    # exact fp64 scan (n_uncertified); the eight hostile rows are folded in by merge_special_kernel; everybody else keeps
    # the fast path unless a zeroed operand row (key 0) sits among the rows behind its threshold (the sparse integers)
    assert st["n_uncertified"] >= 5 and st["n_rescored"] >= 5, st
    if kind != "integers":
        assert st["n_uncertified"] == 5 and st["n_rescored"] == 5 and st["n_rescanned"] == 5, st


@pytest.mark.parametrize("metric", [IP, L2])
def test_fewer_than_k_rows_have_a_finite_score(metric):
    """30 rows of which 25 hold a NaN: 5 neighbours, then pads (I = -1, D = -+FLT_MAX) -- for k = 10 (fast path) and k = 40
    (more than the index holds: two-scan path); reaches the exact scan for the queries it cannot certify"""
    y = gaussian(30, 32, 5); x = gaussian(20, 32, 6)
    y[5:, 3] = NAN
    for k in (10, 40):
        D, I, Dr, Ir, st = _run(metric, x, y, k)
        _same(D, I, Dr, Ir, "k=%d" % k)
        assert (I[:, 5:] == -1).all() and (I[:, :5] >= 0).all()
        fm = np.finfo(np.float32).max
        assert (D[:, 5:] == (fm if metric == L2 else -fm)).all()


@pytest.mark.parametrize("metric", [IP, L2])
def test_scores_beyond_float32(metric):
    """|values| ~ 1e25: every score is finite in fp64 (the order the oracle ranks by) and +-inf as float32"""
    rng = np.random.default_rng(9)
    y = (gaussian(2000, 48, 7) * np.float32(1e25)); x = (gaussian(64, 48, 8) * np.float32(1e25))
    D, I, Dr, Ir, st = _run(metric, x, y, 10)
    _same(D, I, Dr, Ir)
    assert np.isinf(D).any()


@pytest.mark.parametrize("metric", [IP, L2])
def test_denormal_and_zero_rows(metric):
    y = gaussian(1500, 40, 3) * np.float32(1e-41); x = gaussian(40, 40, 4) * np.float32(1e-3)
    y[100:120] = 0.0; x[5] = 0.0
    D, I, Dr, Ir, st = _run(metric, x, y, 10)
    _same(D, I, Dr, Ir)


def test_k_beyond_the_limits_is_refused():
    from textreact_amd import _lib
    idx = _index(IP, 16)
    idx.add(gaussian(100, 16, 1))
    x = gaussian(3, 16, 2)
    with pytest.raises(AssertionError, match=r"k must be in \[1, 2048\]"):
        idx.search(x, _lib.MAX_K + 1)
    with pytest.raises(AssertionError):
        idx.search(x, 0)
    D, I = idx.search(x, _lib.MAX_K)                   # the limit itself: 100 rows, 1948 pads
    assert (I[:, 100:] == -1).all() and (np.sort(I[:, :100], axis=1) == np.arange(100)).all()
    idx.set_tie_rule("faiss")                          # the FAISS tie rule searches for 2k
    with pytest.raises(AssertionError, match="1024"):
        idx.search(x, 1025)


@pytest.mark.parametrize("metric", [IP, L2])
def test_empty_index_then_add_after_search(metric):
    """a search on an empty index returns pads; rows added AFTER a search are found by the next one (the int8 / fp4 copies of the
    operand are rebuilt), also across the plain -> approx transition of the index"""
    from oracle import flat_knn as oracle
    idx = _index(metric, 24)
    x = bf16_round(gaussian(50, 24, 2))
    D, I = idx.search(x, 5)
    fm = np.finfo(np.float32).max
    assert (I == -1).all() and (D == (fm if metric == L2 else -fm)).all()
    assert idx.search(np.zeros((0, 24), np.float32), 5)[1].shape == (0, 5)
    parts = [np.round(gaussian(700, 24, 3) * 4), bf16_round(gaussian(900, 24, 4)), gaussian(800, 24, 5)]     # integers, bf16 values, fp32
    have = np.zeros((0, 24), np.float32)
    for p in parts:
        idx.add(p.astype(np.float32))
        have = np.concatenate([have, p.astype(np.float32)])
        D, I = idx.search(x, 7)
        Dr, Ir = oracle.knn_canonical(metric, x, have, 7)
        _same(D, I, Dr, Ir, "after %d rows" % len(have))
    idx.reset()
    assert idx.ntotal == 0 and (idx.search(x, 3)[1] == -1).all()


@pytest.mark.parametrize("metric", [IP, L2])
def test_one_dimension(metric):
    """d = 1: the operand is one column padded to a whole K-step; ties everywhere (L2 of integers), ids break them"""
    y = np.round(gaussian(5000, 1, 1) * 3).astype(np.float32); x = np.round(gaussian(300, 1, 2) * 3).astype(np.float32)
    D, I, Dr, Ir, st = _run(metric, x, y, 10)
    _same(D, I, Dr, Ir)
    y2, x2 = gaussian(5000, 1, 3), gaussian(300, 1, 4)
    D, I, Dr, Ir, st = _run(metric, x2, y2, 20)
    _same(D, I, Dr, Ir)


def test_non_finite_values_through_the_integer_and_float64_host_types():
    """float64 rows with NaN / inf / values beyond float32 (they become +-inf as float32, as numpy's astype makes them) through
    trx_index_add / trx_index_search of the float64 dtype"""
    from oracle import flat_knn as oracle
    y = gaussian(1000, 20, 1).astype(np.float64); x = gaussian(30, 20, 2).astype(np.float64)
    y[3, 1] = np.nan; y[4, 2] = np.inf; y[5, 3] = 1e300; y[6, 4] = -1e300; x[2, 0] = np.nan; x[3, 1] = -np.inf
    with np.errstate(all="ignore"):
        yf, xf = y.astype(np.float32), x.astype(np.float32)
    for metric in (IP, L2):
        idx = _index(metric, 20); idx.add(y)
        D, I = idx.search(x, 10)
        with np.errstate(all="ignore"):
            Dr, Ir = oracle.knn_canonical(metric, xf, yf, 10)
        _same(D, I, Dr, Ir)


@pytest.mark.parametrize("metric", [IP, L2])
def test_more_hostile_rows_than_the_index_folds_in(metric):
    """1,100 rows with a NaN (MAX_SPECIAL = 1,024): every search of the index is the exact fp64 scan (n_uncertified = nq) -- slow,
    and the oracle's answer; 1,000 of them: still the fast path plus merge_special_kernel"""
    y = gaussian(6000, 32, 5); x = gaussian(64, 32, 6)
    y[100:1100, 3] = NAN
    D, I, Dr, Ir, st = _run(metric, x, y, 10)
    _same(D, I, Dr, Ir, "1000 hostile rows")
    assert st["n_uncertified"] == 0, st
    y[1100:1200, 5] = INF
    D, I, Dr, Ir, st = _run(metric, x, y, 10, chunks=4)
    _same(D, I, Dr, Ir, "1100 hostile rows")
    assert st["n_uncertified"] == 64, st


def test_hostile_rows_with_the_faiss_tie_rule_and_device_tensors():
    """TRX_TIES_FAISS (the search runs for 2k, then the tie kernel) and torch tensors on the GPU (bf16 rows with NaN / inf): the
    hostile rows are merged before the tie rule is applied; equal to the oracle's heap replay on exact-arithmetic data"""
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = np.round(gaussian(4000, 24, 7) * 2).astype(np.float32); x = np.round(gaussian(100, 24, 8) * 2).astype(np.float32)
    y[7, 0] = INF; y[9, 1] = -INF; y[11, :] = NAN; y[13, 2] = np.float32(2.0 ** 100)
    idx = faiss.IndexFlatIP(24, tie_rule="faiss")
    idx.add(torch.from_numpy(y).cuda().bfloat16())
    D, I = idx.search(torch.from_numpy(x).cuda().bfloat16(), 10)
    with np.errstate(all="ignore"):
        Df, If = oracle.knn_faiss(IP, x, y, 10)
    # FAISS' own fp32 arithmetic turns inf * 0 into NaN like the canonical chain does; rows 7 / 9 score +-inf where x is nonzero
    assert np.array_equal(I.cpu().numpy(), If), (I.cpu().numpy()[:3], If[:3])
    assert np.array_equal(D.cpu().numpy().view(np.uint32), Df.view(np.uint32))


def test_sharded_search_folds_in_each_shards_hostile_rows():
    """ShardedFlatIndex over three shards on this GPU (no exchange: the shards are merged here): shard-local special ids become
    global ids through the merge like everybody else's"""
    import torch
    import textreact_amd.faiss_compat as faiss
    from oracle import flat_knn as oracle
    y = gaussian(9000, 48, 3); x = gaussian(80, 48, 4)
    y[10, 0] = INF; y[3000, 1] = NAN; y[3001, :] = np.float32(3e38); y[8999, 2] = -INF
    for metric in (IP, L2):
        Sl, Il = [], []
        for lo in (0, 3000, 6000):
            idx = _index(metric, 48); idx.add(y[lo:lo + 3000])
            D, I, S = idx.search_s64(torch.from_numpy(x).cuda(), 10)
            Sl.append(S); Il.append(torch.where(I >= 0, I + lo, I))
        D, I = faiss.merge_topk(metric, torch.stack(Sl), torch.stack(Il))
        with np.errstate(all="ignore"):
            Dr, Ir = oracle.knn_canonical(metric, x, y, 10)
        _same(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, "three shards")
