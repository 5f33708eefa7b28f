#!/bin/bash
# GPU busy fraction of the training step: sum of kernel durations / span of the kernel timeline (last 3 of 5 steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/prof_busy; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/train_step_one.py 5 > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(glob.glob("$OUT/*/*kernel_trace.csv")[0]))]
rows.sort()
# find optimizer step markers: the fused adam kernel name
marks=[i for i,(s,e,n) in enumerate(rows) if 'adam' in n.lower()]
# group consecutive marks into steps: take the last kernel index of each group
ends=[]
for i in marks:
    if not ends or i-ends[-1]>50: ends.append(i)
    else: ends[-1]=i
print("optimizer groups", len(ends))
if len(ends)>=3:
    a,b=ends[-3],ends[-1]
    seg=rows[a+1:b+1]
    span=(seg[-1][1]-seg[0][0])/1e6; busy=sum(e-s for s,e,_ in seg)/1e6
    print("last 2 steps: span %.2f ms, kernel time %.2f ms, busy %.1f %%, kernels %d" % (span, busy, 100*busy/span, len(seg)))
PY
