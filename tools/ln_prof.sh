#!/bin/bash
# kernel-level times of the mixed-storage add+LayerNorm backward at the training shape
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lnp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lnp -- python3 $R/tools/ln_bwd_time.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/lnp/*/*kernel_stats.csv")[0])):
    if 'add_ln' in r['Name']: print(r['Name'][:70], 'avg_us', round(float(r['AverageNs'])/1e3,1))
PY
