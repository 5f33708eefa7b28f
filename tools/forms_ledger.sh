#!/bin/bash
# ledger of the int8 and fp4 forms of the scan on the reference's workloads (round 6): the production loop, the loop without its
# LDS-DMA (TRX_SCAN_DEBUG=1: MFMAs + fragment reads + selection), with everything L2-resident (=48), and the same three with a
# listing condition that never holds (make abl).  Timing only: every variant but the first returns wrong results.
#   make -C textreact_amd/csrc dbg abl && tools/forms_ledger.sh > gpurun_out/r06/forms_ledger.jsonl
for wl in morgan fingerprint; do
  N=680000; if [ $wl = morgan ]; then N=800000; fi
  for v in "libtrxknn.so 0" "libtrxknn_dbg.so 0" "libtrxknn_dbg.so 1" "libtrxknn_dbg.so 48" "libtrxknn_abl.so 0" "libtrxknn_abl.so 1" "libtrxknn_abl.so 48"; do
    set -- $v
    TRX_NO_RESCAN=1 TRX_LIB=$1 TRX_SCAN_DEBUG=$2 python bench.py --workload $wl --n-corpus $N --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print(json.dumps({'workload':'$wl','lib':'$1','TRX_SCAN_DEBUG':$2,'launch_ms':round(r['launch_ms'],3),'mfma_frac':round(r['frac'],4),'fill_TBps':round(r['fill']['achieved'],2),'form':j['dtype']}))"
  done
done
