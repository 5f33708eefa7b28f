mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_knn_gpu.py -q -m gpu -x -k "multi_rank_sharded or c2_ or bench" 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 900 python -m pytest tests/test_bench_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 900 python tools/shard_costs.py > gpurun_out/r06/shard_costs.json 2> gpurun_out/r06/shard_costs.err; tail -2 gpurun_out/r06/shard_costs.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06/shard_costs.json"))
for r in j["rows"]: print(r)
for r in j["grid_cells"]: print(r)
PY
