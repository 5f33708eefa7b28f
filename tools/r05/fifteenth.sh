#!/bin/bash
# round 5, call 15: row_stats with atomics only where they change a word (old library = the commit before)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in libtrxknn_head.so libtrxknn.so; do
  export TRX_LIB=$lib
  rm -rf $O/small_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_$lib -- python3 $R/tools/r05/small_search.py > $O/small_$lib.log 2>&1
  f=$(ls $O/small_$lib/*/*kernel_stats.csv | head -1)
  echo "== $lib"; head -12 $f | cut -c1-160
done
unset TRX_LIB
cd $R
python -m pytest tests/test_knn_gpu.py -x -q > $O/t_knn7.log 2>&1; grep -h "passed\|failed" $O/t_knn7.log
