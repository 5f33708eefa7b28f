#!/bin/bash
# sweep splits x cache policy on the headline workload with the experiment build (TRX_LIB=libtrxknn_pol.so)
export TRX_LIB=libtrxknn_pol.so
for dbg in 0 2; do
for s in 1 4 8 16; do
for pol in 0 2 1; do
TRX_SCAN_DEBUG=$dbg TRX_NSPLITS=$s TRX_POLICY=$pol python bench.py --no-cpu-baseline --steps 3 --warmup 1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $dbg S $s pol $pol', round(j['ms_per_step'],2), 'ms/step scan', round(j['roofline']['launch_ms'],2), 'uncert', j['config']['uncertified_queries_per_step'])
"
done; done; done
