#!/bin/bash
# round 5, call 17: the whole GPU suite, smoke() and the fuzzers at HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/gputest_final.log 2>&1; grep -h "passed\|failed" $O/gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.log 2>&1; tail -4 $O/smoke_final.log
python tools/knn_fuzz.py 150 5 > $O/knn_fuzz_final.log 2>&1; tail -1 $O/knn_fuzz_final.log
python tools/attn_fuzz.py 120 7 > $O/attn_fuzz_final.log 2>&1; tail -1 $O/attn_fuzz_final.log
python tools/attn_fuzz.py 80 7 fp32 > $O/attn_fuzz32_final.log 2>&1; tail -1 $O/attn_fuzz32_final.log
python tools/tani_fuzz.py 40 3 > $O/tani_fuzz_final.log 2>&1; tail -1 $O/tani_fuzz_final.log
