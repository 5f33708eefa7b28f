#!/bin/bash
# round 5, seventh GPU call: the first read-back of the shared thresholds at tile 9 (+ a refresh there) instead of 11
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
TRX_LIB=libtrxknn_fr9.so python -m pytest tests/test_knn_gpu.py -q -x -k "not sharded_cli and not multi_rank" 2>&1 | grep -E "passed|failed" | tail -2
one() {
  local label=$1; local extra=$2; shift; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$label', 'scan %.2f ms  step %.2f ms (median %.2f)  frac %.4f uncert %s' % (r['launch_ms'], j['ms_per_step'], j['ms_per_step_median'], r['frac'], j['config']['uncertified_queries_per_step']))"
}
for rep in 1 2 3; do
  for n in 1000000 250000 125000; do
    one fr11_$n "--n-corpus $n" TRX_LIB=libtrxknn.so
    one fr9_$n "--n-corpus $n" TRX_LIB=libtrxknn_fr9.so
  done
done > $O/fr_ab.txt 2>&1
cat $O/fr_ab.txt
