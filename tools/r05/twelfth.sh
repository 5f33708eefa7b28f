#!/bin/bash
# round 5, call 12: after the add+LayerNorm load reordering -- predictor tests, the train step with the old and the new
# library (two processes each, alternating), bench_predictor rows, the predictor profile
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_predictor_gpu.py -x -q > $O/t_pred4.log 2>&1; grep -h "passed\|failed" $O/t_pred4.log
: > $O/ln_step_ab.jsonl
for rep in 1 2; do
  for v in lnold ""; do
    lib=libtrxnn${v:+_$v}.so
    TRX_NN_LIB=$lib python tools/r05/ln_ab.py 2>>$O/ln_step_ab.err | sed "s/^{/{\"lib\": \"$lib\", /" >> $O/ln_step_ab.jsonl
  done
done
cut -c1-400 $O/ln_step_ab.jsonl
python bench_predictor.py > $O/predictor_bench.jsonl 2> $O/predictor_bench.err
python bench_predictor.py --live > $O/live_bench.jsonl 2> $O/live_bench.err
bash profiles/run_profile_predictor.sh r05 > $O/run_profile_predictor.log 2>&1
bash tools/prof_train.sh 512 160 > $O/prof_train_160.log 2>&1
python tools/train_soak.py > $O/train_soak.log 2>&1; tail -2 $O/train_soak.log
grep -c "^{" $O/predictor_bench.jsonl $O/live_bench.jsonl
