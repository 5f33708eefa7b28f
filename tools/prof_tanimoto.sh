#!/bin/bash
# kernel-level summary of bench_tanimoto.py (rocprofv3 --kernel-trace --stats) -> gpurun_out/r01_tanimoto_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_tani
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench_tanimoto.py --no-cpu-baseline > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt > $R/gpurun_out/r01_tanimoto_bench.jsonl
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r01_tanimoto_kernel_stats.csv
head -12 $f | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete
