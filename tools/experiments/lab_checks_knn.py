"""Lab checks of the three-term bf16 split of fp32 data (rounds 1-3's operand: corpus [hi|lo|hi], queries [hi|hi|lo], K = 3d; 200 ms
against 75 for the headline search, so the product uses the bf16 rounding + a listing slack instead).  Since round 6 the form
exists in the lab build only:
    make -C textreact_amd/csrc knnlab && TRX_LIB=libtrxknn_lab.so python -m pytest tools/experiments/lab_checks_knn.py -q
Its tight key error (2^-16) is what sends the crowds below through the fixed-threshold re-scan tier and lets that tier SUCCEED;
these were tests/test_knn_gpu.py::*split* until the switch left libtrxknn.so."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _data import bf16_round, gaussian  # noqa: E402
from test_knn_gpu import IP, L2, _check  # noqa: E402

pytestmark = pytest.mark.skipif(os.environ.get("TRX_LIB") != "libtrxknn_lab.so", reason="needs the lab build: TRX_LIB=libtrxknn_lab.so")


@pytest.mark.parametrize("metric", [IP, L2])
def test_c0_gaussian_fp32_through_the_three_term_split(metric, monkeypatch):
    """rounds 1-3's form for fp32 data, the lab build's TRX_FP32_SPLIT=1: corpus [hi|lo|hi], queries [hi|hi|lo], K = 3d"""
    monkeypatch.setenv("TRX_FP32_SPLIT", "1")
    st = _check(metric, gaussian(1000, 768, 5678), gaussian(10000, 768, 1234), 10)
    assert st["k_split"] == 3 * 768 and st["n_uncertified"] == 0


@pytest.mark.parametrize("split", [True, False])
def test_a_crowd_around_the_kth_place_is_resolved_by_the_rescan(split, monkeypatch):
    # (split: the three-term operand of rounds 1-3, whose tight key error this crowd was built against.  The approx mode's
    # error bound is 2^-7 |x||y|: the whole crowd lies inside its listing slack, is listed by the FIRST scan and resolved by the
    # wide re-score -- same answers, no second scan)
    if split:
        monkeypatch.setenv("TRX_FP32_SPLIT", "1")
        # (which tier settles a query depends on where the lists' bound falls inside the crowd; this case was built against the
        # bootstrap bound of rounds 2-4 -- the main scan's rule -- and keeps it, so that the re-scan tier stays exercised here;
        # round 5's tighter bootstrap bound puts the same queries through the wide re-score alone: next test)
        monkeypatch.setenv("TRX_BOOT_J2", "1")
    # the k-th place INSIDE a crowd that reaches down to the lists' bound: scores fall off smoothly (steps far below the rounding
    # bound) over 3000 rows, so whatever the bound is, rows just under it tie with the k-th and the wide re-score cannot
    # certify.  Tier 3 scans again with the threshold fixed at (k-th exact score so far) - 2 eps: the ~1,400 rows above it
    # are listed by construction and re-scored; no fp64 scan of the index
    y = gaussian(4000, 64, 1)
    c = gaussian(1, 64, 2)
    y[500:3500] = c * (1.0 - 1e-7 * np.arange(3000, dtype=np.float32)[:, None])
    x = np.repeat(c, 4, axis=0)
    for metric in (IP, L2):
        st = _check(metric, x, y, 10)
        assert st["n_rescored"] == 4 and st["n_rescanned"] == (4 if split else 0) and st["n_uncertified"] == 0, st


def test_the_bootstrap_bound_changes_tiers_never_answers(monkeypatch):
    """round 5: the bootstrap publishes the 16th largest of a query's 32 tracked maxima instead of the minimum over its lanes'
    second bests (TRX_BOOT_J2=1 keeps the old rule).  A threshold is a hint: the same crowd, the three-term operand, both
    rules -- identical answers (the oracle's), whichever tiers they take; and on benign data neither flags a query"""
    monkeypatch.setenv("TRX_FP32_SPLIT", "1")
    y = gaussian(4000, 64, 1)
    c = gaussian(1, 64, 2)
    y[500:3500] = c * (1.0 - 1e-7 * np.arange(3000, dtype=np.float32)[:, None])
    x = np.repeat(c, 4, axis=0)
    tiers = []
    for rule in ("1", None):
        if rule:
            monkeypatch.setenv("TRX_BOOT_J2", rule)
        else:
            monkeypatch.delenv("TRX_BOOT_J2")
        for metric in (IP, L2):
            st = _check(metric, x, y, 10)
            assert st["n_uncertified"] == 0, st
            tiers.append((st["n_rescored"], st["n_rescanned"]))
    monkeypatch.delenv("TRX_FP32_SPLIT")
    for rule in ("1", None):
        if rule:
            monkeypatch.setenv("TRX_BOOT_J2", rule)
        else:
            monkeypatch.delenv("TRX_BOOT_J2")
        st = _check(IP, bf16_round(gaussian(700, 768, 5678)), bf16_round(gaussian(60000, 768, 1234)), 10)
        assert st["n_rescored"] == 0 and st["n_uncertified"] == 0, st
    assert tiers[0][1] == 4          # (the old rule's bound sends this crowd through the re-scan: what the test above pins)


@pytest.mark.parametrize("split", [True, False])
def test_a_crowd_wider_than_the_lists_still_takes_the_exact_scan(split, monkeypatch):
    # 10,000 rows within the rounding bound of each other: more than a query's lists (and the wide re-score) hold
    if split:
        monkeypatch.setenv("TRX_FP32_SPLIT", "1")
    y = gaussian(12000, 64, 1)
    c = gaussian(1, 64, 2)
    y[1000:11000] = c * (1.0 - 1e-8 * np.arange(10000, dtype=np.float32)[:, None])
    x = np.repeat(c, 4, axis=0)
    st = _check(IP, x, y, 10)
    assert st["n_uncertified"] == 4 and (st["n_rescanned"] == 4 or not split), st


