#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel-substring> <command...>  -- separate rocprofv3 --pmc passes (counters only)
tag=$1; filt=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
: > $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --pmc $set -d /tmp/pmc$i -- "$@" > /tmp/pmc$i.log 2>&1
  db=$(find /tmp/pmc$i -name "*results.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db "$filt" 2>/dev/null | grep -v "^kernel" >> $out
done
cat $out
