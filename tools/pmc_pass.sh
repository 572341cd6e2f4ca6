#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <bench args...> -- runs separate rocprofv3 --pmc passes (counters only, no tracing)
# and writes gpurun_out/<tag>_pmc.txt with per-kernel means.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
: > $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --pmc $set -d /tmp/pmc$i -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 "$@" > /tmp/pmc$i.log 2>&1
  db=$(find /tmp/pmc$i -name "*results.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db sgs:: 2>/dev/null | grep -v "^kernel" >> $out
done
grep -E "blend_|^ " $out | head -80
