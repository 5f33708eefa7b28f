#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/train_step_one.py ${3:-8} hip ${1:-512} ${2:-160} > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/train_step_kernel_stats_${1:-512}.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/1e6/${3:-8})
for r in rows[:28]:
    print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['TotalDurationNs'])/1e6/${3:-8},3), r['Percentage'])
PY
