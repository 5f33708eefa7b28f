#!/bin/bash
# round 5, call 14: operand builders with 16-byte loads: the kNN tests, index.add timing (old library from the commit before
# against this tree's), the fingerprint / morgan workloads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_knn_gpu.py -x -q > $O/t_knn6.log 2>&1; grep -h "passed\|failed" $O/t_knn6.log
for lib in libtrxknn_head.so libtrxknn.so; do
TRX_LIB=$lib python - <<'PY'
import os, time, torch
import textreact_amd.faiss_compat as faiss
g = torch.Generator(device="cuda"); g.manual_seed(1)
for dt, n, d in ((torch.bfloat16, 1_000_000, 768), (torch.float32, 1_000_000, 768), (torch.bfloat16, 500_000, 1024)):
    y = torch.randn((n, d), generator=g, device="cuda").to(dt) if d == 768 else (torch.rand((n, d), generator=g, device="cuda") < 0.05).to(dt)
    ts = []
    for rep in range(3):
        idx = faiss.IndexFlatL2(d) if d == 1024 else faiss.IndexFlatIP(d)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.add(y); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 2 and d == 1024:
            q = y[:70000]
            idx.search(q, 20); torch.cuda.synchronize(); t0 = time.perf_counter()
            idx.search(q, 20); torch.cuda.synchronize()
            print(os.environ.get("TRX_LIB"), "search 70000 x 1024 bit vectors ms", round((time.perf_counter() - t0) * 1e3, 2))
        del idx
    print(os.environ.get("TRX_LIB"), str(dt), n, d, "add ms", [round(t, 2) for t in ts])
PY
done
python bench.py --workload fingerprint --no-cpu-baseline > $O/fingerprint_bench2.jsonl 2> $O/fingerprint_bench2.err; cut -c1-200 $O/fingerprint_bench2.jsonl
python bench.py --workload morgan --n-corpus 800000 --no-cpu-baseline > $O/morgan_bench2.jsonl 2> $O/morgan_bench2.err; cut -c1-200 $O/morgan_bench2.jsonl
