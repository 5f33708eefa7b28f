"""CPU tests of the oracle itself (it is the checker of every GPU parity test, so it is pinned
first): committed golden vectors, an independent NumPy statement of the canonical rule, properties
any exact flat search must have, and the relation between the literal FAISS restatement and the
canonical rule (identical on exact-arithmetic inputs, near-tie-only differences otherwise)."""
import glob
import os

import numpy as np
import pytest

from _data import bf16_round, gaussian, grid, morgan_like, reaction_fp_like
from oracle import flat_knn as oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IP, L2 = 0, 1
FMAX = np.finfo(np.float32).max


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "knn_*.npz"))))
def test_golden_vectors(path):
    z = np.load(path)
    m, k = int(z["metric"]), int(z["k"])
    Df, If = oracle.knn_faiss(m, z["x"], z["y"], k)
    Dc, Ic = oracle.knn_canonical(m, z["x"], z["y"], k)
    assert np.array_equal(If, z["I_faiss"]) and np.array_equal(Df.view(np.uint32), z["D_faiss"].view(np.uint32))
    assert np.array_equal(Ic, z["I_canonical"]) and np.array_equal(Dc.view(np.uint32), z["D_canonical"].view(np.uint32))


@pytest.mark.parametrize("metric", [IP, L2])
@pytest.mark.parametrize("maker", [grid, lambda n, d, s: reaction_fp_like(n, d, s, 0.05), morgan_like])
def test_canonical_c_equals_numpy_statement(metric, maker):
    y = maker(400, 48, 3)
    x = maker(30, 48, 4)
    Dc, Ic = oracle.knn_canonical(metric, x, y, 12)
    Dn, In = oracle.knn_numpy(metric, x, y, 12)
    assert np.array_equal(Ic, In) and np.array_equal(Dc, Dn)


@pytest.mark.parametrize("metric", [IP, L2])
def test_canonical_is_sorted_complete_and_matches_bruteforce_scores(metric):
    x, y = gaussian(40, 64, 1), gaussian(1500, 64, 2)
    k = 15
    D, I = oracle.knn_canonical(metric, x, y, k)
    S = oracle.scores_at(metric, x, y, I)
    assert np.array_equal(S.astype(np.float32), D)
    # best-first, ties by id
    d = np.diff(S, axis=1)
    assert (d >= 0).all() if metric == L2 else (d <= 0).all()
    # nothing outside the list beats the k-th entry
    full = ((x[:, None, :].astype(np.float64) - y[None].astype(np.float64)) ** 2).sum(-1) if metric == L2 \
        else x.astype(np.float64) @ y.astype(np.float64).T
    for q in range(x.shape[0]):
        rest = np.delete(full[q], I[q])
        assert (rest >= S[q, -1] - 1e-9).all() if metric == L2 else (rest <= S[q, -1] + 1e-9).all()
        assert len(set(I[q].tolist())) == k


def test_l2_faiss_restatement_equals_canonical_on_exact_inputs():
    # integer fingerprints: every fp32 partial sum exact -> FAISS's blocks+heap == (dist, id) order
    for y in (reaction_fp_like(3000, 512, 5), morgan_like(3000, 256, 6)):
        y[50:90] = y[50]
        x = y[:64].copy()
        for k in (1, 20):
            Df, If = oracle.knn_faiss(L2, x, y, k)
            Dc, Ic = oracle.knn_canonical(L2, x, y, k)
            assert np.array_equal(If, Ic) and np.array_equal(Df, Dc)
            assert (Df[:, 0] == 0).all()          # the query is in the corpus
        # both code paths of the restatement: nq < 20 is the sequential one
        assert np.array_equal(oracle.knn_faiss(L2, x[:7], y, 20)[1], oracle.knn_canonical(L2, x[:7], y, 20)[1])


def test_ip_faiss_restatement_tie_order_is_a_heap_artefact():
    # equal scores: the min-heap keeps first-seen ids and emits larger ids first; the product's DEFAULT rule is
    # (score desc, id asc) and its TRX_TIES_FAISS mode returns the heap's answer (next test; tests/test_knn_gpu.py:
    # test_faiss_tie_rule_for_the_inner_product); the SETS agree when ties do not straddle k
    y = np.ones((6, 4), dtype=np.float32)
    x = np.ones((25, 4), dtype=np.float32)
    Df, If = oracle.knn_faiss(IP, x, y, 6)
    Dc, Ic = oracle.knn_canonical(IP, x, y, 6)
    assert np.array_equal(np.sort(If, 1), np.sort(Ic, 1)) and np.array_equal(Df, Dc)
    assert Ic[0].tolist() == [0, 1, 2, 3, 4, 5]
    assert If[0].tolist() != Ic[0].tolist()


def faiss_ip_order_from_canonical(Dc, Ic, k):
    """The closed form the product's TRX_TIES_FAISS mode computes (knn_select.hip: faiss_tie_kernel), restated in numpy: from
    the canonical top 2k (score desc, id asc) of an inner-product search, what FAISS' min-heap returns.  a = rows above the
    k-th score, G = the rows tied at it (ids ascending); of G only the rows among the first k of (better u G) by id were
    admitted (the heap was not yet full), each better row that came later evicted the smallest id among them, and
    heap_reorder emits (score desc, id desc)."""
    nq = Dc.shape[0]
    D = np.full((nq, k), -np.finfo(np.float32).max, dtype=np.float32); I = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        d, i = Dc[q], Ic[q]
        nvalid = int((i >= 0).sum())
        if nvalid <= k:
            res = [(d[t], i[t]) for t in range(nvalid)]
        else:
            sk = d[k - 1]
            a = int((d[:nvalid] > sk).sum())
            better = [(d[t], i[t]) for t in range(a)]
            tied = [i[t] for t in range(a, min(nvalid, a + k)) if d[t] == sk]
            first_k = sorted([(idv, 1) for _, idv in better] + [(g, 0) for g in tied])[:k]
            admitted = [idv for idv, is_better in first_k if not is_better]
            res = better + [(sk, g) for g in admitted[len(admitted) - (k - a):]]
        res.sort(key=lambda t: (-t[0], -t[1]))
        D[q, :len(res)] = [t[0] for t in res]; I[q, :len(res)] = [t[1] for t in res]
    return D, I


def test_ip_tie_order_of_the_faiss_heap_has_a_closed_form():
    """the divergence the test above documents is closed by a mode of the product (TRX_TIES_FAISS); this pins the rule that
    mode computes against the heap replay of the literal restatement, on inputs with exact ties of every kind"""
    rng = np.random.default_rng(0)
    differ = 0
    for trial in range(120):
        n = int(rng.integers(5, 400)); d = int(rng.choice([4, 8, 16, 32])); k = int(rng.integers(1, 25)); nq = 20
        if trial % 3 == 0:
            y = rng.integers(-2, 3, (n, d)).astype(np.float32); x = rng.integers(-2, 3, (nq, d)).astype(np.float32)
        elif trial % 3 == 1:      # Gaussian rows repeated: identical rows score identically
            base = rng.standard_normal((max(2, n // 5), d)).astype(np.float32)
            y = base[rng.integers(0, len(base), n)]; x = rng.standard_normal((nq, d)).astype(np.float32)
        else:
            y = grid(n, d, trial); x = grid(nq, d, trial + 1000)
        Df, If = oracle.knn_faiss(IP, x, y, k)
        Dc, Ic = oracle.knn_canonical(IP, x, y, 2 * k)
        D2, I2 = faiss_ip_order_from_canonical(Dc, Ic, k)
        assert np.array_equal(I2, If), (trial, n, d, k)
        if trial % 3 != 1:
            assert np.array_equal(D2, Df), (trial, n, d, k)
        differ += int(not np.array_equal(If, Ic[:, :k]))
    assert differ > 60


@pytest.mark.parametrize("metric", [IP, L2])
def test_gaussian_differences_are_provable_near_ties(metric):
    x, y = gaussian(300, 128, 11), gaussian(20000, 128, 12)
    k = 10
    for fn in (oracle.knn_faiss, oracle.knn_faiss_blas):
        Df, If = fn(metric, x, y, k)
        Dc, Ic = oracle.knn_canonical(metric, x, y, k)
        bad = np.argwhere(If != Ic)
        if bad.size:
            Sf = oracle.scores_at(metric, x, y, If)
            Sc = oracle.scores_at(metric, x, y, Ic)
            for q, t in bad:
                # a mismatch is only allowed where the canonical scores differ by fp32 rounding
                assert abs(Sf[q, t] - Sc[q, t]) <= 64 * np.finfo(np.float32).eps * max(1.0, abs(Sc[q, t]))
        assert np.allclose(Df, Dc, rtol=1e-5, atol=1e-4)


def test_blas_variant_equals_plain_restatement_on_exact_inputs():
    y = grid(2500, 64, 1); x = grid(100, 64, 2)
    for metric in (IP, L2):
        a = oracle.knn_faiss(metric, x, y, 10)
        b = oracle.knn_faiss_blas(metric, x, y, 10, bs_x=64, bs_y=512)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


def test_padding_and_edges():
    x = gaussian(21, 8, 1)
    for metric, pad in ((IP, -FMAX), (L2, FMAX)):
        for fn in (oracle.knn_faiss, oracle.knn_canonical):
            D, I = fn(metric, x, gaussian(3, 8, 2), 5)
            assert (I[:, 3:] == -1).all() and (D[:, 3:] == pad).all() and (I[:, :3] >= 0).all()
            D, I = fn(metric, x, np.zeros((0, 8), np.float32), 4)
            assert (I == -1).all() and (D == pad).all()
    with pytest.raises(AssertionError):
        oracle.knn_canonical(IP, np.zeros((2, 5), np.float32), np.zeros((3, 6), np.float32), 1)


def test_merge_lists_equals_unsharded():
    y = grid(900, 32, 3); x = grid(40, 32, 4)     # ties across shards
    for metric in (IP, L2):
        D, I = oracle.knn_canonical(metric, x, y, 10)
        parts, offs = np.array_split(y, 3), [0, 300, 600]
        Dl, Il = [], []
        for part, off in zip(parts, offs):
            d, i = oracle.knn_canonical(metric, x, part, 10)
            Dl.append(d); Il.append(np.where(i >= 0, i + off, i))
        Dm, Im = oracle.merge_lists(metric, np.stack(Dl), np.stack(Il))
        assert np.array_equal(Im, I) and np.array_equal(Dm, D)


@pytest.mark.parametrize("metric", [0, 1])
def test_oracle_agrees_with_an_independent_exact_search(metric):
    """FAISS is not installed here (the oracle's header says 'parity unpinned'), but scikit-learn is: its
    brute-force NearestNeighbors is an independent exact k-NN.  On Gaussian data (no exact ties) both of
    the oracle's statements must return its neighbours, in its order, up to pairs whose scores differ by
    less than float32 can resolve."""
    from sklearn.neighbors import NearestNeighbors
    from _data import gaussian
    y, x = gaussian(4000, 96, 41), gaussian(150, 96, 42)
    k = 10
    if metric == 1:
        nn = NearestNeighbors(n_neighbors=k, algorithm="brute", metric="euclidean").fit(y.astype(np.float64))
        _, ids = nn.kneighbors(x.astype(np.float64))
    else:   # inner product: brute force on float64 scores
        ids = np.argsort(-(x.astype(np.float64) @ y.astype(np.float64).T), axis=1, kind="stable")[:, :k]
    s = (x.astype(np.float64) @ y.astype(np.float64).T) if metric == 0 else \
        -((x.astype(np.float64)[:, None, :] - y.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    for name in ("knn_canonical", "knn_faiss"):
        _, I = getattr(oracle, name)(metric, x, y, k)
        diff = np.argwhere(I != ids)
        for q, j in diff:   # only near-ties may differ
            assert abs(s[q, I[q, j]] - s[q, ids[q, j]]) <= 1e-4 * max(1.0, abs(s[q, ids[q, j]])), (name, q, j)
        assert len(diff) <= 0.01 * ids.size, (name, len(diff))


def test_oracle_agrees_with_an_independent_brute_force_library():
    """scikit-learn's NearestNeighbors(algorithm='brute') -- a third implementation of exact flat search, written by nobody
    here and unrelated to FAISS -- finds the same neighbours as the oracle's canonical rule and its literal FAISS restatement
    on Gaussian data (no ties), and the same distances up to float rounding.  This does not pin the oracle to FAISS (nothing in
    this image can: DESIGN.md section 1); it rules out a shared misreading of what "the k nearest rows" means."""
    sk = pytest.importorskip("sklearn.neighbors")
    y, x = gaussian(3000, 64, 11), gaussian(200, 64, 12)
    nn = sk.NearestNeighbors(n_neighbors=10, algorithm="brute", metric="euclidean").fit(y.astype(np.float64))
    dist, ind = nn.kneighbors(x.astype(np.float64))
    Dc, Ic = oracle.knn_canonical(L2, x, y, 10)
    Df, If = oracle.knn_faiss(L2, x, y, 10)
    assert np.array_equal(Ic, ind) and np.array_equal(If, ind)
    assert np.allclose(Dc, dist ** 2, rtol=1e-5, atol=1e-5) and np.allclose(Df, dist ** 2, rtol=1e-4, atol=1e-4)
    # inner product: the same library's machinery on the augmented vectors (max x.y == min |x' - y'|^2 with y' = [y, sqrt(M - |y|^2)], x' = [x, 0])
    n2 = (y.astype(np.float64) ** 2).sum(1)
    ya = np.concatenate([y.astype(np.float64), np.sqrt(n2.max() - n2)[:, None]], 1)
    xa = np.concatenate([x.astype(np.float64), np.zeros((len(x), 1))], 1)
    _, ind_ip = sk.NearestNeighbors(n_neighbors=10, algorithm="brute", metric="euclidean").fit(ya).kneighbors(xa)
    _, Ic_ip = oracle.knn_canonical(IP, x, y, 10)
    assert np.array_equal(Ic_ip, ind_ip)
