#!/bin/bash
# rocprofv3 evidence for the predictor kernels ("K4/K5 achieved GB/s and MFMA util from rocprofv3").
# Usage (GPU box, through gpurun): profiles/run_profile_predictor.sh <tag>; then profiles/summarize_predictor.py <tag> here.
# One pass for the kernel trace, one --pmc pass per counter group (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share one).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_${TAG}_predictor
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/predictor_shapes.py 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
grep -h PLAN $OUT/trace.log > $OUT/plan.json
tail -3 $OUT/pmc_sq.log
