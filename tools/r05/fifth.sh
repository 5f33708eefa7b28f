#!/bin/bash
# round 5, fifth GPU call: the training step with the weight gradients flushed in groups against all at the end
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for rep in 1 2; do
for mb in 2048 1000000 4096; do
TRX_NN_WGRAD_BUDGET_MB=$mb python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench_predictor as bp
rows = [o for o in bp.train_step_bench("cuda", T=160) if o["kernel"] == "train_step" and o["ops"] == "hip"]
print("budget_mb", os.environ["TRX_NN_WGRAD_BUDGET_MB"], "train step ms", [round(o["ms"], 2) for o in rows], "peak GiB %.2f" % (torch.cuda.max_memory_allocated() / 2**30), flush=True)
PY
done; done 2>/dev/null | tee $O/wgrad_budget.txt
