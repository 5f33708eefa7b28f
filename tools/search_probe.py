#!/usr/bin/env python3
"""time one flat-index search at a given shape (corpus rows, queries) on Gaussian bf16 data: wall per call + library stats.
    python tools/search_probe.py 204800 40000 [k]          (under rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import textreact_amd.faiss_compat as faiss

n, nq = int(sys.argv[1]), int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
g = torch.Generator(device="cuda"); g.manual_seed(5)
y = torch.randn((n, 768), generator=g, device="cuda").bfloat16()
x = torch.randn((nq, 768), generator=g, device="cuda").bfloat16()
idx = faiss.IndexFlatIP(768)
idx.add(y)
idx.set_timing(True)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    D, I = idx.search(x, k)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = idx.last_stats()
    print("call %d: %.2f ms wall; scan %.2f ms in %d launch(es), total %.2f ms, splits %d, uncertified %d" % (
        i, (t1 - t0) * 1e3, st["scan_ms"], st["scan_launches"], st["total_ms"], st["n_splits"], st["n_uncertified"]), flush=True)
