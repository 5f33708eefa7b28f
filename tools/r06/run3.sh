mkdir -p gpurun_out/r06
timeout 900 tools/attn_ablate.sh > gpurun_out/r06/attention_ablation.json 2> gpurun_out/r06/attention_ablation.err; tail -3 gpurun_out/r06/attention_ablation.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06/attention_ablation.json"))
for r in j["shapes"]:
    print(r["what"]); 
    for n,v in r["variants"].items(): print("   %-8s %7.2f %7.2f"%(n, v["us_median"], v["us_min"]))
PY
timeout 600 python tools/attn_ab.py noaug=tools/ab/libtrxnn_noaug.so aug=tools/ab/libtrxnn_aug.so head=textreact_amd/csrc/libtrxnn.so > gpurun_out/r06/attention_ab_aug.json 2> gpurun_out/r06/attention_ab_aug.err
