#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun).  Usage: profiles/run_profile.sh <tag>
# Pass 1: kernel trace + stats.  Pass 2/3: HBM traffic counters, one --pmc pass each (TCC slots).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# what was measured: the text of the sources of this snapshot (the box has no .git); profiles/summarize.py refuses to write
# the committed summaries unless these are the files HEAD holds
(cd $R && sha256sum textreact_amd/csrc/knn_scan.hip textreact_amd/csrc/knn_common.h textreact_amd/csrc/knn_api.hip \
   textreact_amd/csrc/knn_select.hip textreact_amd/csrc/knn_prep.hip bench.py) > $OUT/source.sha256
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -40
tail -2 $OUT/trace.log
