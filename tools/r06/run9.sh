mkdir -p gpurun_out/r06
timeout 600 python tools/attn_ab.py old=tools/ab/libtrxnn_nointerleave.so nop0=tools/ab/libtrxnn_ilnop0.so nop1=textreact_amd/csrc/libtrxnn.so nop4=tools/ab/libtrxnn_ilnop4.so > gpurun_out/r06/attention_ab_interleave.json 2> gpurun_out/r06/attention_ab_interleave.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06/attention_ab_interleave.json"))
for r in j["shapes"]:
    print(r["what"], {n:(round(v["us_median"],2), round(v["us_min"],2), v["max_abs_diff_vs_first"]) for n,v in r["variants"].items()})
PY
