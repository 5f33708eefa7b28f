"""stress the index life cycle: many short-lived indexes, searches with exact ties (flagged queries -> wide re-score -> exact scan),
k > 24 (exact path for every query), interleaved"""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import textreact_amd.faiss_compat as faiss
from _data import grid, gaussian
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
y = grid(3000, 64, 3); x = grid(500, 64, 4)
yg = gaussian(20000, 96, 1); xg = gaussian(300, 96, 2)
for r in range(rounds):
    for metric in (0, 1):
        parts = np.array_split(y, 5)
        keep = []
        for part in parts:
            idx = faiss.IndexFlat(64, metric)
            idx.add(torch.from_numpy(part).cuda())
            D, I, S = idx.search_s64(torch.from_numpy(x).cuda(), 10)
            keep.append((D, I, S))
        big = faiss.IndexFlat(96, metric)
        big.add(yg)
        big.search(xg, 10)
        big.search(xg[:7], 40)          # k > 24: the exact path
        del big
    if r % 10 == 0:
        gc.collect(); torch.cuda.synchronize(); print("round", r, "ok", flush=True)
print("done")
