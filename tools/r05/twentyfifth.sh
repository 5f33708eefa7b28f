#!/bin/bash
# round 5, call 25: backward kernels with hidden keys / queries masked in place (one elementwise form): tests, fuzz, timing against
# the previous commit's library (two processes per library, alternating; dropout 0 and 0.1)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_predictor_gpu.py -x -q > $O/t_pred7.log 2>&1; grep -h "passed\|failed" $O/t_pred7.log
python tools/attn_fuzz.py 160 31 > $O/attn_fuzz7.log 2>&1; tail -1 $O/attn_fuzz7.log
: > $O/attn_bwd_ab.txt
for rep in 1 2; do for p in 0.0 0.1; do for lib in libtrxnn_prev.so libtrxnn.so; do
  echo "== $lib p=$p" >> $O/attn_bwd_ab.txt
  TRX_NN_LIB=$lib python tools/attn_bwd_ab.py $p 2>/dev/null >> $O/attn_bwd_ab.txt
done; done; done
cat $O/attn_bwd_ab.txt
