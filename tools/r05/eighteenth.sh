#!/bin/bash
# round 5, call 18: LDS and wait counters of the attention kernels at the encoder shape (separate --pmc passes, kernel trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05/attn_lds
rm -rf $O; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/r05/attn_enc_only.py 10"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- $CMD > $O/p$i.log 2>&1
  echo "pass $i rc=$? : $set"; tail -1 $O/p$i.log | cut -c1-200
done
ls $O
