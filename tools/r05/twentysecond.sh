#!/bin/bash
# round 5, call 22: add+LayerNorm forward as a row loop over a grid of whole rounds (slots 1024 / 1280 / 1536) against one row per wave
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
: > $O/lnf_ab.jsonl
for rep in 1 2; do
  for v in lnold lnf1024 lnf1280 lnf1536; do
    lib=libtrxnn_$v.so
    TRX_NN_LIB=$lib python tools/ln_bench.py 2>>$O/lnf_ab.err | sed "s/^{/{\"lib\": \"$lib\", /" >> $O/lnf_ab.jsonl
  done
done
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r05/lnf_ab.jsonl"):
    r = json.loads(l)
    if "first stage" in r["variant"]: continue
    d[(r["rows"], r["variant"][:6], r["lib"])].append(r["fwd_us"])
for k in sorted(d): print(k, d[k])
PY
