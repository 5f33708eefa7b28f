#!/bin/bash
# round 5, first GPU call: the new tests, the shard-size sweep, the scan kernel's L2 share priced (debug bits 16 / 32 of the
# dbg build), cache-policy and split-count variants under bench.py, stamps at two shard sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_knn_gpu.py -q -x -k "faiss_tie or multi_rank" > $O/t_ties.log 2>&1; echo "rc=$?" >> $O/t_ties.log
python -m pytest tests/test_bench_gpu.py -q -x -k "gpus" > $O/t_bench.log 2>&1; echo "rc=$?" >> $O/t_bench.log
tail -3 $O/t_ties.log $O/t_bench.log
python tools/r05/shard_sweep.py > $O/shard_sweep.json 2> $O/shard_sweep.err
one() {  # label, env assignments...
  local label=$1; shift
  env "$@" TRX_NO_RESCAN=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'scan %.2f ms  step %.2f ms  frac %.4f uncert %s' % (j['roofline']['launch_ms'], j['ms_per_step_median'], j['roofline']['frac'], j['config']['uncertified_queries_per_step']))"
}
for rep in 1 2; do
  one base TRX_LIB=libtrxknn.so
  one S8 TRX_LIB=libtrxknn.so TRX_NSPLITS=8
  one S2 TRX_LIB=libtrxknn.so TRX_NSPLITS=2
  one polA2B0 TRX_LIB=libtrxknn_polA2B0.so
  one polA0B16 TRX_LIB=libtrxknn_polA0B16.so
  one polA0B2 TRX_LIB=libtrxknn_polA0B2.so
  one dbg0 TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=0
  one dbg16_oneQtile TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=16
  one dbg32_oneAtile TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=32
  one dbg48_allL2 TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=48
  one dbg1_nodma TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=1
  one dbg2_nofilter TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=2
  one dbg50_nofilter_allL2 TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=50
  one dbg18_nofilter_oneQtile TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=18
  one dbg34_nofilter_oneAtile TRX_LIB=libtrxknn_dbg.so TRX_SCAN_DEBUG=34
done > $O/scan_ab.txt 2>&1
cat $O/scan_ab.txt
# stamps: rows listed per query and bookkeeping cycles at the full corpus and at the 8-way shard
for n in 1000000 125000; do
  TRX_LIB=libtrxknn_stamp.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --n-corpus $n 2>&1 | grep -a "stamp" | tail -2 | sed "s/^/n=$n /"
done > $O/stamps.txt 2>&1
cat $O/stamps.txt
# the clock and the fabric bytes with everything L2-resident against the normal run (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
export TRX_LIB=libtrxknn_dbg.so TRX_NO_RESCAN=1
for dbg in 2 50; do
  export TRX_SCAN_DEBUG=$dbg
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm_dbg$dbg -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_grbm_dbg$dbg.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_dbg$dbg -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_dbg$dbg.log 2>&1
done
find $O -name "*counter_collection.csv" | head
