#!/bin/bash
# round 5, call 21: which kernels the live refresh's search spends its time in (real encoder embeddings, mean cosine 0.99)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05/live_prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench_predictor.py --live > $O/log.txt 2>&1
f=$(ls $O/trace/*/*kernel_stats.csv | head -1)
grep "trx::" $f | cut -c1-200 | head -20
