mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_predictor_gpu.py -x -q -m gpu -k "attention" 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 900 python tools/attn_fuzz.py 120 7 2>&1 | tail -1
timeout 600 python tools/attn_ab.py old=tools/ab/libtrxnn_nointerleave.so new=textreact_amd/csrc/libtrxnn.so > gpurun_out/r06/attention_ab_interleave.json 2> gpurun_out/r06/attention_ab_interleave.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06/attention_ab_interleave.json"))
for r in j["shapes"]:
    print(r["what"], {n:(round(v["us_median"],2), round(v["us_min"],2), v["max_abs_diff_vs_first"]) for n,v in r["variants"].items()})
PY
