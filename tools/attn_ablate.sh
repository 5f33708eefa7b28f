#!/bin/bash
# timing-only ablations of attention_fwd_mfma_kernel (TRX_ATT_ABL bits, textreact_amd/csrc/nn_ops.hip): what each part of a key
# tile costs when it is taken out -- the results of every variant but `head` are wrong by construction.  Built by
#   for a in 1 2 3 4 8 16 32 48 64 72 127 128 255; do make -C textreact_amd/csrc nnvar NAME=abl$a D=-DTRX_ATT_ABL=$a; done
# usage: tools/attn_ablate.sh > gpurun_out/r06/attention_ablation.json
args="head=textreact_amd/csrc/libtrxnn.so"
for a in 1 2 3 4 8 16 32 48 64 72 127 128 255; do args="$args abl$a=tools/ab/libtrxnn_abl$a.so"; done
python3 tools/attn_ab.py $args
