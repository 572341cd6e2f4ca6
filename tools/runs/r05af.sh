#!/bin/bash
# round 5, GPU call AF: bwd_geom_kernel with / without the per-wave hit mask -- kernel times (rocprofv3 --kernel-trace) and dynamic instruction counts (--pmc, own passes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$GRAFT_REPO_ROOT/gpurun_out/r05af; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for g in 1 0; do
rm -rf /tmp/kb
SGS_BWD_GEOM_NOFILTER=$g rocprofv3 --kernel-trace --stats -d /tmp/kb -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > /dev/null 2>&1
db=$(find /tmp/kb -name "*results.db" | head -1)
echo "NOFILTER=$g: $(python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db sgs:: | grep bwd_geom)" | tee -a $O/times.txt
done; done
for g in 1 0; do
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE"; do
rm -rf /tmp/pm
SGS_BWD_GEOM_NOFILTER=$g rocprofv3 --pmc $set -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > /dev/null 2>&1
db=$(find /tmp/pm -name "*results.db" | head -1)
echo "NOFILTER=$g" >> $O/pmc.txt
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db sgs:: 2>/dev/null | grep -A6 "== sgs::(anonymous namespace)::bwd_geom" >> $O/pmc.txt
done; done
cat $O/pmc.txt
