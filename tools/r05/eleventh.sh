#!/bin/bash
# round 5, call 11: add+LayerNorm kernels with every load of a row issued before its arithmetic (old = HEAD~'s library,
# new = this tree's, pf = new + next-row prefetch at 2 waves per SIMD), tools/ln_bench.py interleaved; then the predictor tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_predictor_gpu.py -x -q > $O/t_pred3.log 2>&1; tail -2 $O/t_pred3.log
python -m pytest tests/test_knn_gpu.py -x -q -k "cli" > $O/t_cli.log 2>&1; tail -2 $O/t_cli.log
: > $O/ln_ab.jsonl
for rep in 1 2; do
  for v in lnold "" lnpf; do
    lib=libtrxnn${v:+_$v}.so
    TRX_NN_LIB=$lib python tools/ln_bench.py 2>>$O/ln_ab.err | sed "s/^{/{\"lib\": \"$lib\", /" >> $O/ln_ab.jsonl
  done
done
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r05/ln_ab.jsonl"):
    r = json.loads(l)
    d[(r["rows"], r["variant"][:24], r["lib"])].append((r["fwd_us"], r["bwd_us"]))
for k in sorted(d): print(k, d[k])
PY
