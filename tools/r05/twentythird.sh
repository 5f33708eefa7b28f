#!/bin/bash
# round 5, call 23: hidden keys masked in place (one tile form, three waves per SIMD with dropout too): tests, fuzz, timing against
# the library of an earlier commit (tools/ab/libtrxnn_head.so), outputs compared
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_predictor_gpu.py -x -q > $O/t_pred6.log 2>&1; grep -h "passed\|failed" $O/t_pred6.log
python tools/attn_fuzz.py 160 21 > $O/attn_fuzz6.log 2>&1; tail -1 $O/attn_fuzz6.log
python tools/r05/persist_check.py > $O/persist_check.log 2>&1; tail -1 $O/persist_check.log
python tools/r05/persist_time.py head=tools/ab/libtrxnn_head.so new=textreact_amd/csrc/libtrxnn.so persist=tools/ab/libtrxnn_persist.so > $O/attention_dropout_ab.json 2> $O/attention_dropout_ab.err
cat $O/attention_dropout_ab.json; tail -2 $O/attention_dropout_ab.err
