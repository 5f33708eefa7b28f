#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05/clustered
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/r05/clustered_search.py > $O/log.txt 2>&1
tail -2 $O/log.txt | cut -c1-600
f=$(ls $O/trace/*/*kernel_stats.csv | head -1); head -16 $f | cut -c1-170
