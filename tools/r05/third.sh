#!/bin/bash
# round 5, third GPU call: one round of workgroups per scan launch (TRX_QUERY_BATCH) against the four-round launch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
one() {  # label, extra bench args, env assignments...
  local label=$1; local extra=$2; shift; shift
  env "$@" TRX_NO_RESCAN=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$label', 'scan/launch %.2f ms  step %.2f ms (median %.2f)  frac %.4f uncert %s q/s %.0f' % (r['launch_ms'], j['ms_per_step'], j['ms_per_step_median'], r['frac'], j['config']['uncertified_queries_per_step'], j['value']))"
}
for rep in 1 2; do
  one base "" TRX_LIB=libtrxknn.so
  one qb32768 "" TRX_LIB=libtrxknn.so TRX_QUERY_BATCH=32768
  one qb16384 "" TRX_LIB=libtrxknn.so TRX_QUERY_BATCH=16384
  one qb16384_S8 "" TRX_LIB=libtrxknn.so TRX_QUERY_BATCH=16384 TRX_NSPLITS=8
  one qb8192 "" TRX_LIB=libtrxknn.so TRX_QUERY_BATCH=8192
  one base_125k "--n-corpus 125000" TRX_LIB=libtrxknn.so
  one qb16384_125k "--n-corpus 125000" TRX_LIB=libtrxknn.so TRX_QUERY_BATCH=16384
  one qb32768_125k "--n-corpus 125000" TRX_LIB=libtrxknn.so TRX_QUERY_BATCH=32768
  one base_250k "--n-corpus 250000" TRX_LIB=libtrxknn.so
  one qb16384_250k "--n-corpus 250000" TRX_LIB=libtrxknn.so TRX_QUERY_BATCH=16384
done > $O/scan_ab3.txt 2>&1
cat $O/scan_ab3.txt
cd /tmp && export TMPDIR=/tmp
export TRX_LIB=libtrxknn.so TRX_NO_RESCAN=1
for qb in 65536 16384; do
  export TRX_QUERY_BATCH=$qb
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm_qb$qb -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_grbm_qb$qb.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_qb$qb -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_qb$qb.log 2>&1
done
echo done
