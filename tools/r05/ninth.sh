#!/bin/bash
# round 5, ninth GPU call: bootstrap length with the tighter bound (TRX_BOOT_TILES)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
one() {
  local label=$1; local extra=$2; shift; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$label', 'scan %.2f ms  step %.2f ms (median %.2f)  frac %.4f uncert %s' % (r['launch_ms'], j['ms_per_step'], j['ms_per_step_median'], r['frac'], j['config']['uncertified_queries_per_step']))"
}
for rep in 1 2; do
  for n in 1000000 250000 125000; do
    for bt in 16 8 12 4; do
      one boot${bt}_$n "--n-corpus $n" TRX_BOOT_TILES=$bt
    done
  done
done > $O/boot_tiles_ab.txt 2>&1
cat $O/boot_tiles_ab.txt
