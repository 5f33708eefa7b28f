#!/bin/bash
# round 5, second GPU call: the tie-rule and fp32-attention tests, fp32 attention A/B, the L2 share priced on the loop that
# never lists (abl build, debug bits 16 / 32), interval stamps, publish / refresh schedules at the full corpus and the 8-way shard
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_knn_gpu.py -q -x -k "faiss_tie or multi_rank" > $O/t_ties.log 2>&1; echo "rc=$?" >> $O/t_ties.log
python -m pytest tests/test_predictor_gpu.py tests/test_torch_ops.py -q -x > $O/t_pred.log 2>&1; echo "rc=$?" >> $O/t_pred.log
tail -3 $O/t_ties.log $O/t_pred.log
python tools/r05/attn_f32_ab.py > $O/attn_f32_ab.jsonl 2> $O/attn_f32_ab.err
TRX_NN_ATTN_VALU=1 python tools/r05/attn_f32_ab.py >> $O/attn_f32_ab.jsonl 2>> $O/attn_f32_ab.err
cat $O/attn_f32_ab.jsonl | cut -c1-400
one() {  # label, extra bench args, env assignments...
  local label=$1; local extra=$2; shift; shift
  env "$@" TRX_NO_RESCAN=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'scan %.2f ms  step %.2f ms  frac %.4f uncert %s' % (j['roofline']['launch_ms'], j['ms_per_step_median'], j['roofline']['frac'], j['config']['uncertified_queries_per_step']))"
}
for rep in 1 2; do
  one base "" TRX_LIB=libtrxknn.so
  one abl0 "" TRX_LIB=libtrxknn_abl.so TRX_SCAN_DEBUG=0
  one abl16_oneQtile "" TRX_LIB=libtrxknn_abl.so TRX_SCAN_DEBUG=16
  one abl32_oneAtile "" TRX_LIB=libtrxknn_abl.so TRX_SCAN_DEBUG=32
  one abl48_allL2 "" TRX_LIB=libtrxknn_abl.so TRX_SCAN_DEBUG=48
  one abl1_nodma "" TRX_LIB=libtrxknn_abl.so TRX_SCAN_DEBUG=1
  one pub7 "" TRX_LIB=libtrxknn_pub7.so
  one pub7r3 "" TRX_LIB=libtrxknn_pub7r3.so
  one r3 "" TRX_LIB=libtrxknn_r3.so
  one base_125k "--n-corpus 125000" TRX_LIB=libtrxknn.so
  one abl0_125k "--n-corpus 125000" TRX_LIB=libtrxknn_abl.so TRX_SCAN_DEBUG=0
  one pub7_125k "--n-corpus 125000" TRX_LIB=libtrxknn_pub7.so
  one pub7r3_125k "--n-corpus 125000" TRX_LIB=libtrxknn_pub7r3.so
  one r3_125k "--n-corpus 125000" TRX_LIB=libtrxknn_r3.so
done > $O/scan_ab2.txt 2>&1
cat $O/scan_ab2.txt
for n in 1000000 125000; do
  TRX_LIB=libtrxknn_stamp.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --n-corpus $n 2>&1 | grep -a "stamp" | tail -2 | sed "s/^/n=$n /"
done > $O/stamps2.txt 2>&1
cat $O/stamps2.txt
# the clock and the fabric bytes of the loop that never lists, normal against everything L2-resident (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
export TRX_LIB=libtrxknn_abl.so TRX_NO_RESCAN=1
for dbg in 0 48; do
  export TRX_SCAN_DEBUG=$dbg
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm_abl$dbg -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_grbm_abl$dbg.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_abl$dbg -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_abl$dbg.log 2>&1
done
echo done
