#!/bin/bash
# round 5, call 16: the headline profile + bench lines + shard costs + the new misaligned-rows test, one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_knn_gpu.py -x -q -k "do_not_start_on_16_bytes or cli" > $O/t_knn8.log 2>&1; grep -h "passed\|failed" $O/t_knn8.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_probe.jsonl 2> $O/bench_probe.err
python - <<'PY'
import json
for l in open("gpurun_out/r05/bench_probe.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print("probe", round(r["value"]), r["roofline"]["launch_ms"], r["roofline"]["frac"])
PY
bash profiles/run_profile.sh r05 > $O/run_profile.log 2>&1
python bench.py > $O/bench.jsonl 2> $O/bench.err
python bench.py --steps 20 --warmup 2 --no-cpu-baseline >> $O/bench.jsonl 2>> $O/bench.err
python tools/shard_costs.py > $O/shard_costs.json 2> $O/shard_costs.err
python bench_predictor.py > $O/predictor_bench.jsonl 2> $O/predictor_bench.err
grep -c "^{" $O/bench.jsonl $O/predictor_bench.jsonl
