#!/bin/bash
# round 5, call 24: train step with the library of commit 3d968cf against this tree's (two processes each, alternating); predictor rows
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
: > $O/attn_step_ab.jsonl
for rep in 1 2; do
  for lib in $R/tools/ab/libtrxnn_head.so libtrxnn.so; do
    TRX_NN_LIB=$lib python tools/r05/ln_ab.py 2>>$O/attn_step_ab.err | sed "s#^{#{\"lib\": \"$(basename $lib)\", #" >> $O/attn_step_ab.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r05/attn_step_ab.jsonl"):
    r = json.loads(l); print(r["lib"], r["deferred"]["median"], r["deferred"]["min"])
PY
python bench_predictor.py > $O/predictor_bench.jsonl 2> $O/predictor_bench.err
bash profiles/run_profile_predictor.sh r05 > $O/run_profile_predictor.log 2>&1
bash tools/prof_train.sh 512 160 > $O/prof_train_160.log 2>&1
python tools/train_soak.py > $O/train_soak.log 2>&1; tail -2 $O/train_soak.log
