#!/bin/bash
# round 5, eighth GPU call: the deferred add+LayerNorm column sums -- tests, the training step against TRX_NN_WGRAD=percall
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_predictor_gpu.py tests/test_torch_ops.py tests/test_main_cli.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for rep in 1 2; do
for mode in deferred noln; do
TRX_LN_MODE=$mode python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from textreact_amd.predictor import ops
if os.environ["TRX_LN_MODE"] == "noln":
    ops._ln_deferrable = lambda params, needs: False       # the per-call second stage (round 4's form), weight gradients still grouped
import bench_predictor as bp
for T in (160, 7):
    rows = [o for o in bp.train_step_bench("cuda", T=T) if o["kernel"] == "train_step" and o["ops"] == "hip"]
    print("ln second stage:", os.environ["TRX_LN_MODE"], "T", T, "train step ms", [round(o["ms"], 2) for o in rows], flush=True)
PY
done; done 2>/dev/null | tee $O/ln_deferred.txt
