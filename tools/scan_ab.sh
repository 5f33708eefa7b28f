#!/bin/bash
# same-box A/B of two builds of libtrxknn.so under bench.py (TRX_LIB names the library relative to textreact_amd/csrc):
#   tools/scan_ab.sh libtrxknn.so ../../tools/ab/libtrxknn_prevscan.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in $A $B; do
    TRX_NO_RESCAN=1 TRX_LIB=$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'scan %.2f ms  step %.2f ms  frac %.4f' % (j['roofline']['launch_ms'], j['ms_per_step_median'], j['roofline']['frac']))"
  done
done
