#!/bin/bash
# round 5, tenth GPU call: the measured rounding bound of the approximate mode -- tests, fuzz, fp32 search A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_knn_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/knn_fuzz.py 150 31 2>&1 | tail -1
for rep in 1 2; do
  TRX_ROUND_BOUND_APRIORI=1 python tools/fp32_search_ab.py 2>/dev/null | cut -c1-600
  python tools/fp32_search_ab.py 2>/dev/null | cut -c1-600
done | tee $O/fp32_search.jsonl
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-300
