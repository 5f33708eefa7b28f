#!/bin/bash
# Clock evidence for the "power-limited" reading of the scan kernel.  Usage: profiles/run_clock.sh <tag>
#  1. production kernel under bench.py: effective clock = GRBM_GUI_ACTIVE / 8 / dispatch duration
#     (MI355X_MICROARCH.md, DVFS give-back: the counter is summed over the 8 XCDs)
#  2. tools/scan_lab (the contraction alone, same tile / loop structure, random data, >= 2 s of back-to-back launches):
#     in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz, for {MFMA on register operands, + LDS fragment reads,
#     + LDS-DMA fill, everything}
# then: python3 profiles/summarize_clock.py <tag>   (here, after gpurun merged gpurun_out/ back)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/clock_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_grbm -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc_grbm.log 2>&1
# lab: variant 4 = whole-K-step phases (flag 8 exists there); variant 2 = the loop the production kernel uses
for f in 9 1 8 0; do timeout 300 $R/tools/scan_lab 4 4 $f 3907 30 > $OUT/lab_v4_f$f.log 2>&1; done
for f in 1 0; do timeout 300 $R/tools/scan_lab 2 4 $f 3907 30 > $OUT/lab_v2_f$f.log 2>&1; done
tail -n 3 $OUT/lab_*.log
