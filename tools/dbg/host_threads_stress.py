"""round 6: the host entry points from several Python threads at once -- every thread its own index, int64 / int8 / float32 arrays,
adds and searches interleaved (the staging pipeline and the worker pool serve one caller at a time; the device workspaces are
shared): every answer equals the oracle's.  python3 tools/dbg/host_threads_stress.py [threads [rounds]]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import textreact_amd.faiss_compat as faiss
from oracle import flat_knn as oracle
from _data import gaussian, reaction_fp_like, morgan_like
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
errors = []

def work(t):
    try:
        rng = np.random.default_rng(t)
        for r in range(rounds):
            kind = (t + r) % 3
            d = [512, 96, 1024][kind]
            if kind == 0:
                y = reaction_fp_like(4000 + 37 * t, d, 10 * t + r).astype(np.int64); x = y[:300 + t]
            elif kind == 1:
                y = gaussian(6000, d, 10 * t + r); x = gaussian(200 + t, d, 10 * t + r + 1)
            else:
                y = morgan_like(5000, d, 10 * t + r).astype(np.int8); x = y[100:400]
            metric = (t + r) & 1
            idx = faiss.IndexFlat(d, metric)
            for part in np.array_split(y, 1 + (r % 3)):
                idx.add(part)
            D, I = idx.search(x, 10)
            Dr, Ir = oracle.knn_canonical(metric, x.astype(np.float32), y.astype(np.float32), 10)
            if not (np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32))):
                errors.append((t, r, kind, metric))
    except Exception as e:      # noqa: BLE001
        errors.append((t, repr(e)))

ts = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
[t.start() for t in ts]; [t.join() for t in ts]
print("threads", nthreads, "rounds", rounds, "errors", errors)
sys.exit(1 if errors else 0)
