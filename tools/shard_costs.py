"""what one shard of the strong-scaling split costs on one GPU: the 65,536-query top-10 search of bench.py against
1,000,000 / 500,000 / 250,000 / 125,000 rows (the shard of 1, 2, 4, 8 GPUs), scan kernel and whole step, median of 10
event-timed steps each; optionally with other bootstrap lengths (TRX_BOOT_TILES) for the short shards.
    python3 tools/shard_costs.py > gpurun_out/shard_costs.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import textreact_amd.faiss_compat as faiss  # noqa: E402

dev = torch.device("cuda", 0)
queries = bench.make_rows(65536, 768, 5678, dev)
rows = []
for n in (1_000_000, 500_000, 250_000, 125_000):
    shard = bench.make_rows(n, 768, 1234, dev)
    for boot in (None, 4, 8, 32) if n <= 250_000 else (None,):
        if boot is None:
            os.environ.pop("TRX_BOOT_TILES", None)
        else:
            os.environ["TRX_BOOT_TILES"] = str(boot)
        idx = faiss.IndexFlatIP(768, device=0)
        idx.add(shard)
        idx.set_timing(True)
        for _ in range(2):
            idx.search(queries, 10)
        steps, scans = [], []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); idx.search(queries, 10); b.record(); b.synchronize()
            steps.append(a.elapsed_time(b)); scans.append(idx.last_stats()["scan_ms"])
        steps.sort(); scans.sort()
        rows.append({"corpus_rows": n, "boot_tiles": boot if boot is not None else "default", "scan_ms_median": scans[5], "step_ms_median": steps[5],
                     "step_ms_min": steps[0], "n_splits": idx.last_stats()["n_splits"]})
        del idx
    del shard
os.environ.pop("TRX_BOOT_TILES", None)
# round 6: the cells of a rows x queries grid over 8 (and 4) ranks -- sharded.ShardedFlatIndex(row_groups=Gr): Gr row shards x Gq query slices
grid = []
for gr, gq in ((4, 2), (2, 4), (1, 8), (2, 2), (1, 4), (1, 2)):
    n, nq = 1_000_000 // gr, 65536 // gq
    shard = bench.make_rows(n, 768, 1234, dev)
    idx = faiss.IndexFlatIP(768, device=0)
    idx.add(shard); idx.set_timing(True)
    q = queries[:nq].contiguous()
    for _ in range(2):
        idx.search(q, 10)
    steps, scans = [], []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); idx.search(q, 10); b.record(); b.synchronize()
        steps.append(a.elapsed_time(b)); scans.append(idx.last_stats()["scan_ms"])
    steps.sort(); scans.sort()
    grid.append({"row_groups": gr, "query_groups": gq, "ranks": gr * gq, "corpus_rows": n, "queries": nq, "scan_ms_median": scans[5],
                 "step_ms_median": steps[5], "step_ms_min": steps[0], "n_splits": idx.last_stats()["n_splits"]})
    del idx, shard
base = next(r for r in rows if r["corpus_rows"] == 1_000_000)["step_ms_median"]
for r in rows:
    r["ideal_ms"] = base * r["corpus_rows"] / 1_000_000
for r in grid:
    r["ideal_ms"] = base / r["ranks"]
print(json.dumps({"what": "65,536 queries x 768, exact IP top-10, one MI355X; step = query statistics + bootstrap + scan + select (+ inline fall-back slots)",
                  "rows": rows, "grid_cells": grid}, indent=1))
