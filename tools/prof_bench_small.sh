#!/bin/bash
# usage: tools/prof_bench_small.sh <n_corpus>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_small_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 --n-corpus $1 > $OUT/log.txt 2>&1
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print(r['Name'][:80].ljust(80), r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'tot_ms', round(float(r['TotalDurationNs'])/1e6,2))
PY
