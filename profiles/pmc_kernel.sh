#!/bin/bash
# usage: profiles/pmc_kernel.sh <tag> <kernel-name-substring> <counters...>   one --pmc pass over bench.py, counters of one kernel
TAG=$1; KN=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list)
dur=None
for r in csv.DictReader(open(f)):
    if "$KN" in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
        dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
print("$TAG", {k: sum(v)/len(v) for k,v in agg.items()}, "last_ms", dur)
PY
