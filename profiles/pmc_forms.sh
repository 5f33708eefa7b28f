#!/bin/bash
# HBM-side traffic of the scan kernel's int8 and fp4 forms on the reference's own workloads (bench.py --workload fingerprint:
# 680,000 x 2048 count fingerprints; --workload morgan: 800,000 x 1024 bit vectors), collected as the microarchitecture guide
# prescribes: one rocprofv3 --pmc pass per counter (FETCH_SIZE, WRITE_SIZE), a kernel-trace pass for the duration.
#   gpurun -- bash profiles/pmc_forms.sh <tag>      then here: python3 profiles/summarize_forms.py <tag>
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/forms_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd $R && sha256sum textreact_amd/csrc/knn_scan.hip textreact_amd/csrc/knn_common.h) > $OUT/source.sha256
cd /tmp && export TMPDIR=/tmp
for wl in fingerprint morgan; do
  N=680000; if [ $wl = morgan ]; then N=800000; fi
  CMD="python3 $R/bench.py --workload $wl --n-corpus $N --steps 1 --warmup 1 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${wl}_trace -- $CMD > $OUT/${wl}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${wl}_fetch -- $CMD > $OUT/${wl}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${wl}_write -- $CMD > $OUT/${wl}_write.log 2>&1
  tail -1 $OUT/${wl}_trace.log | cut -c1-300
done
find $OUT -name "*.csv" | wc -l
