#!/bin/bash
# round 5, GPU call D: dL/dF through slot rows + a gather kernel instead of 233 M atomics
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_training_loop.py -q -m gpu -k "backward or training or grad" --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 300 python tools/bench_bwd_modes.py 4 6 0 3 2>&1 | tee $O/modes.txt
for d in 1 2 4 8 15; do echo "SGS_BWD_DBG=$d"; SGS_BWD_DBG=$d timeout 120 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode; done | tee $O/ablations.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bwd -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 6 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py /tmp/prof/bwd_results.db 2>&1 | head -12 | cut -c1-150 | tee $O/kernel_stats.txt
