#!/bin/bash
# usage: tools/attn_pmc.sh <counters...>   one --pmc pass over tools/predictor_shapes.py (the five attention shapes, 10 calls each;
# the program itself directly behind "--": no env / bash -c hop under rocprofv3), counters of the forward MFMA kernel averaged
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/attn_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/predictor_shapes.py 10 > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob("$OUT/*/*_counter_collection.csv")
if not fs:
    print(open("$OUT/log.txt").read()[-2000:]); raise SystemExit
agg=collections.defaultdict(list); dur=[]
for r in csv.DictReader(open(fs[0])):
    if 'attention_fwd_mfma' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
        dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print({k: sum(v[5:])/len(v[5:]) for k,v in agg.items()}, "us", sorted(dur)[len(dur)//2] if dur else None)
PY
