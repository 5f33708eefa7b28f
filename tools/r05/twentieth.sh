#!/bin/bash
# round 5, call 20: after choose_splits -- kNN / live / sharded tests, the fuzzer, the workloads whose launches it changes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_knn_gpu.py tests/test_live_gpu.py tests/test_c2_rehearsal_gpu.py tests/test_bench_gpu.py -x -q > $O/t_knn9.log 2>&1; grep -h "passed\|failed" $O/t_knn9.log
python tools/knn_fuzz.py 200 11 > $O/knn_fuzz9.log 2>&1; tail -1 $O/knn_fuzz9.log
python bench.py --workload morgan --n-corpus 800000 > $O/morgan_bench.jsonl 2> $O/morgan_bench.err
TRX_NO_FP4=1 python bench.py --workload morgan --n-corpus 800000 --no-cpu-baseline >> $O/morgan_bench.jsonl 2>> $O/morgan_bench.err
python bench.py --workload fingerprint > $O/fingerprint_bench.jsonl 2> $O/fingerprint_bench.err
python tools/bigk_probe.py 1000000 16384 > $O/bigk_probe.jsonl 2> $O/bigk_probe.err
python bench_predictor.py --live > $O/live_bench.jsonl 2> $O/live_bench.err
python tools/fp32_search_ab.py > $O/fp32_search.jsonl 2> $O/fp32_search.err
python - <<'PY'
import json
for f in ("morgan_bench", "fingerprint_bench", "live_bench"):
    for l in open("gpurun_out/r05/%s.jsonl" % f):
        if l.startswith("{"):
            r = json.loads(l)
            print(f, round(r.get("value", 0)), r.get("ms_per_step"), (r.get("ms") or {}).get("search") if isinstance(r.get("ms"), dict) else None)
PY
