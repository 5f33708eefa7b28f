#!/bin/bash
# rocprofv3 kernel trace of bench_predictor.py (run on the GPU box through gpurun)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_pred_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench_predictor.py > $OUT/bench.log 2>&1
head -14 $OUT/trace/*/*kernel_stats.csv | cut -c1-170
grep '"kernel"' $OUT/bench.log | cut -c1-400
