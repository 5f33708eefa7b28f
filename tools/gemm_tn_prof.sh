#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tnp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tnp -- python3 $R/tools/gemm_tn_one.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
for r in list(csv.DictReader(open(glob.glob("/tmp/tnp/*/*kernel_stats.csv")[0])))[:4]:
    print(r["Name"][:70].ljust(70), "calls", r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1))
PY
