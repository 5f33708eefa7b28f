"""randomised shapes / densities / magnitudes through TanimotoIndex against the numpy oracle (bit-identical similarities,
identical ranks): python3 tools/tani_fuzz.py [n_cases] [seed]"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tanimoto as oracle  # noqa: E402  (the checker)
from textreact_amd.tanimoto import TanimotoIndex  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
for it in range(n_cases):
    d = rng.choice([8, 64, 256, 1024, 2048, 2052])
    n = rng.choice([1, 63, 64, 65, 1000, 4097, 9000, 30000])
    nq = rng.choice([1, 15, 16, 17, 63, 64, 65, 130])
    k = rng.choice([1, 10, 100, 1000])
    density = rng.choice([0.01, 0.03, 0.2, 0.6])
    hi = rng.choice([1, 3, 12])
    if density * d * hi > 30000:
        hi = 3
    r = np.random.default_rng(it)
    def make(m):
        x = r.integers(-hi, hi + 1, (m, d))
        x[r.random((m, d)) >= density] = 0
        return x
    distinct = rng.choice([None, None, 20])            # sometimes a heavily tied corpus
    corpus = make(n) if distinct is None else make(distinct)[r.integers(0, distinct, n)]
    queries = make(nq)
    if rng.random() < 0.5:
        queries[0] = corpus[r.integers(0, n)]
    idx = TanimotoIndex(d)
    idx.add(corpus)
    sim, rank = idx.search(queries, k)
    ws, wr = oracle.search(queries, corpus, k)
    ok = np.array_equal(rank.cpu().numpy(), wr) and np.array_equal(sim.cpu().numpy().view(np.uint64), ws.view(np.uint64))
    if not ok:
        fails += 1
        print("FAIL", it, dict(d=d, n=n, nq=nq, k=k, density=density, hi=hi, distinct=distinct))
print(n_cases, "cases,", fails, "failures")
