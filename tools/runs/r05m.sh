#!/bin/bash
# round 5, GPU call M: cfg3 backward against the oracle on sampled tiles; kernel stats + PMC of the fused backward
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05m; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "oracle_on_sampled or worklist_path" --timeout=800 -s ) > $O/pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "cfg3 backward|passed|failed|real" $O/pytest.txt | tail -24
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bwd -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 4 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py /tmp/prof/bwd_results.db 2>&1 | head -12 | cut -c1-150 | tee $O/kernel_stats.txt
timeout 600 bash tools/pmc_kernel.sh r05m/backward "bwd_" python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > $O/pmc.log 2>&1
grep -A12 "bwd_fused" gpurun_out/r05m/backward_pmc.txt | grep -E "==|FETCH|WRITE|MFMA|WAIT|BUSY|LDS" | head -40
