import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import textreact_amd.faiss_compat as faiss
from oracle import flat_knn as oracle
from _data import bf16_round, gaussian
metric, d, n, nq, k, seed = [int(a) for a in sys.argv[1:7]]
y, x = bf16_round(gaussian(n, d, 2 * seed)), bf16_round(gaussian(nq, d, 2 * seed + 1))
idx = faiss.IndexFlat(d, metric)
idx.add(y)
for rep in range(3):
    D, I = idx.search(x, k)
Dr, Ir = oracle.knn_canonical(metric, x, y, k)
print("ok" if np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32)) else "MISMATCH", idx.last_stats())
