"""Tanimoto brute force (retrieve/retrieve.py) without a GPU: the oracle's vectorised statement against the
sparse-vector walk of RDKit's calcVectParams, the ranking rule, and the C ABI of libtrxtani.so (builds, loads, exports
what include/trx_tanimoto.h declares, rejects bad arguments, never computes on the CPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import tanimoto as oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fingerprints(rng, n, d, density=0.03, lo=-4, hi=5):
    x = rng.integers(lo, hi, (n, d))
    x[rng.random((n, d)) >= density] = 0
    return x.astype(np.int64)


def test_dense_statement_equals_the_sparse_walk():
    rng = np.random.default_rng(1)
    c = fingerprints(rng, 60, 96, density=0.3)
    c[7] = 0                                           # an empty fingerprint
    for qi in (0, 7, 13):
        s = oracle.similarities(c[qi], c)
        for i in range(len(c)):
            v1 = {j: int(v) for j, v in enumerate(c[qi]) if v}
            v2 = {j: int(v) for j, v in enumerate(c[i]) if v}
            assert s[i] == oracle.sparse_similarity(v1, v2)
    assert oracle.similarities(c[7], c)[7] == 0.0       # both empty: denominator 0 -> 0.0
    assert oracle.similarities(c[3], c)[3] == 1.0


def test_signs_do_not_matter_only_magnitudes():
    a = np.array([[2, -1, 0, 3]]); b = np.array([[-2, 1, 5, 0]])
    # and = min(2,2) + min(1,1) = 3; |a| = 6, |b| = 8 -> 3 / 11
    assert oracle.similarities(a[0], b)[0] == 3.0 / 11.0
    assert oracle.sparse_similarity({0: 2, 1: -1, 3: 3}, {0: -2, 1: 1, 2: 5}) == 3.0 / 11.0


def test_rank_is_a_stable_argsort_read_backwards():
    s = np.array([0.5, 0.25, 0.5, 1.0, 0.25, 0.5])
    assert oracle.rank(s, 6).tolist() == [3, 5, 2, 0, 4, 1]
    sim, rk = oracle.search(np.array([[1, 0, 0, 0]]), np.array([[1, 0, 0, 0], [0, 1, 0, 0], [1, 0, 0, 0]]), k=5)
    assert rk.tolist() == [[2, 0, 1]] and sim.tolist() == [[1.0, 1.0, 0.0]]


def _declared():
    src = open(os.path.join(ROOT, "include", "trx_tanimoto.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(trx_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_the_header():
    from textreact_amd import tanimoto
    assert _declared() == sorted(tanimoto.SYMBOLS)
    L = tanimoto.lib()
    for sym in _declared():
        assert hasattr(L, sym), sym
    src = open(os.path.join(ROOT, "include", "trx_tanimoto.h")).read()
    assert "retrieve/retrieve.py" in src and ":34-40" in src and ":55-62" in src


def test_argument_errors_and_no_cpu_path():
    import torch
    from textreact_amd import tanimoto
    L = tanimoto.lib()
    assert L.trx_tanimoto_packed_bytes(130, 2048) == 192 * 2048
    assert L.trx_tanimoto_packed_bytes(5, 6) == -1
    assert L.trx_tanimoto_pack(None, 0, 5, 6, 6, 0, None, None, None, None) == -1
    assert b"d % 4" in L.trx_tanimoto_last_error()
    assert L.trx_tanimoto_scores(None, None, 1 << 27, 8, None, None, 1, None, 1 << 27, None, None) == -1
    assert b"2^27" in L.trx_tanimoto_last_error()
    if not torch.cuda.is_available():
        with pytest.raises(tanimoto.TrxTanimotoError):
            tanimoto.TanimotoIndex(2048)
