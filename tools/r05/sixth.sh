#!/bin/bash
# round 5, sixth GPU call: the bootstrap's tighter bound (J = 4 union selection) against the main scan's rule (TRX_BOOT_J2=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_knn_gpu.py -q -x > $O/t_knn.log 2>&1; echo "rc=$?" >> $O/t_knn.log
grep -E "passed|failed|rc=" $O/t_knn.log | tail -2
timeout 900 python tools/knn_fuzz.py 80 11 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
one() {  # label, extra bench args, env assignments...
  local label=$1; local extra=$2; shift; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$label', 'scan %.2f ms  step %.2f ms (median %.2f)  frac %.4f uncert %s' % (r['launch_ms'], j['ms_per_step'], j['ms_per_step_median'], r['frac'], j['config']['uncertified_queries_per_step']))"
}
for rep in 1 2 3; do
  for n in 1000000 250000 125000; do
    one j2_$n "--n-corpus $n" TRX_BOOT_J2=1
    one j4_$n "--n-corpus $n" TRX_LIB=libtrxknn.so
  done
done > $O/boot_ab.txt 2>&1
cat $O/boot_ab.txt
