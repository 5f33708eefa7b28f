#!/bin/bash
# kernel-level view of the graph-replayed beam search (tools/gen_profile.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/genprof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/genprof -o gen --output-format csv -- python3 $R/tools/gen_profile.py > $R/gpurun_out/genprof.log 2>&1
f=$(find $R/gpurun_out/genprof -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-230
find $R/gpurun_out/genprof -name "*kernel_trace.csv" -delete
