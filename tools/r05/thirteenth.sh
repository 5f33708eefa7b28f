#!/bin/bash
# round 5, call 13: the attention forward prologue (mask, Q and the first two tiles requested together, one wait): tests, fuzz,
# shapes A/B against the library of the commit before
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_predictor_gpu.py -x -q > $O/t_pred5.log 2>&1; grep -h "passed\|failed" $O/t_pred5.log
python tools/attn_fuzz.py 120 > $O/attn_fuzz5.log 2>&1; tail -2 $O/attn_fuzz5.log
python tools/attn_shapes_ab.py head=tools/ab/libtrxnn_head.so new=textreact_amd/csrc/libtrxnn.so > $O/attention_ab_prologue.json 2> $O/attention_ab_prologue.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/attention_ab_prologue.json"))
for s in d["shapes"]:
    print(s["what"], {n: (round(v["fwd_us_median"], 2), round(v["bwd_us_median"], 2), v["max_abs_diff_vs_first [out, lse, dq, dk, dv]"]) for n, v in s["variants"].items()})
PY
