"""round 5: what the corpus splits per query tile (TRX_NSPLITS) and the bootstrap cost at the shard sizes of the strong-scaling
split: 65,536 queries x 768 against 1,000,000 / 250,000 / 125,000 rows, scan kernel and whole step, median of 10 event-timed
steps, all on one box in one process.
    python3 tools/r05/shard_sweep.py > gpurun_out/r05/shard_sweep.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
import textreact_amd.faiss_compat as faiss  # noqa: E402

dev = torch.device("cuda", 0)
queries = bench.make_rows(65536, 768, 5678, dev)
sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1_000_000, 250_000, 125_000]
envs = [{}] + [{"TRX_NSPLITS": str(s)} for s in (1, 2, 8)] + [{"TRX_NO_BOOT": "1"}]
rows = []
for n in sizes:
    shard = bench.make_rows(n, 768, 1234, dev)
    idx = faiss.IndexFlatIP(768, device=0)
    idx.add(shard)
    idx.set_timing(True)
    for rep in range(2):
        for env in envs:
            for k_, v_ in env.items():
                os.environ[k_] = v_
            for _ in range(2):
                idx.search(queries, 10)
            steps, scans = [], []
            for _ in range(10):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); idx.search(queries, 10); b.record(); b.synchronize()
                steps.append(a.elapsed_time(b)); scans.append(idx.last_stats()["scan_ms"])
            steps.sort(); scans.sort()
            rows.append({"corpus_rows": n, "env": env, "rep": rep, "scan_ms_median": scans[5], "step_ms_median": steps[5], "step_ms_min": steps[0],
                         "n_splits": idx.last_stats()["n_splits"], "uncertified": idx.last_stats()["n_uncertified"]})
            for k_ in env:
                del os.environ[k_]
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    del idx, shard
print(json.dumps({"what": "65,536 queries x 768, exact IP top-10, one MI355X", "rows": rows}, indent=1))
