#!/bin/bash
# round 5, fourth GPU call: fp32 attention backward on the matrix cores (tests + A/B), the training step with the grouped
# weight gradients flushed in groups
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_predictor_gpu.py tests/test_torch_ops.py -q -x > $O/t_pred2.log 2>&1; echo "rc=$?" >> $O/t_pred2.log
grep -E "passed|failed|rc=" $O/t_pred2.log | tail -3
python tools/r05/attn_f32_ab.py > $O/attn_f32_ab2.jsonl 2> $O/attn_f32_ab2.err
TRX_NN_ATTN_VALU=1 python tools/r05/attn_f32_ab.py >> $O/attn_f32_ab2.jsonl 2>> $O/attn_f32_ab2.err
cut -c1-330 $O/attn_f32_ab2.jsonl; tail -3 $O/attn_f32_ab2.err
python - <<'PY' > $O/train_step.jsonl 2> $O/train_step.err
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench_predictor as bp
for T in (160, 7):
    for o in bp.train_step_bench("cuda", T=T):
        print(json.dumps(o), flush=True)
PY
cut -c1-300 $O/train_step.jsonl; tail -3 $O/train_step.err
