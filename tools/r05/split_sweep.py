"""round 5: search time against the number of corpus splits (TRX_NSPLITS, read per call) for query counts that do not fill
whole rounds of 256 workgroups at 4 splits; bf16, IP, k = 10, Gaussian.  python3 tools/r05/split_sweep.py"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import textreact_amd.faiss_compat as faiss
g = torch.Generator(device="cuda"); g.manual_seed(2)
shapes = [(40000, 204800), (25000, 1000000), (4464, 800000), (65536, 125000), (12800, 500000)]
for nq, n in shapes:
    y = torch.randn((n, 768), generator=g, device="cuda").bfloat16()
    x = torch.randn((nq, 768), generator=g, device="cuda").bfloat16()
    idx = faiss.IndexFlatIP(768); idx.add(y)
    nqt = (nq + 255) // 256
    row = {"nq": nq, "n": n, "query_tiles": nqt, "corpus_tiles": (n + 255) // 256, "ms_by_splits": {}}
    cands = sorted(set([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 15, 16, 20, 24, 32] + [max(4, -(-256 // nqt))]))
    for s in ["default"] + cands:
        if s == "default":
            os.environ.pop("TRX_NSPLITS", None)
        else:
            os.environ["TRX_NSPLITS"] = str(s)
        ts = []
        for rep in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); D, I = idx.search(x, 10); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        row["ms_by_splits"][str(s)] = round(min(ts[1:]), 3)
        if s == "default":
            row["default_splits"] = idx.last_stats()["n_splits"]
    os.environ.pop("TRX_NSPLITS", None)
    print(json.dumps(row), flush=True)
    del idx, x, y
