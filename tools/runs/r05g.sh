#!/bin/bash
# round 5, GPU call G: fused backward with the finishing split between the K halves and a one-round-trip chunk prologue
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_training_loop.py -q -m gpu -k "backward or training or grad" --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 300 python tools/bench_bwd_modes.py 4 0 5 3 2>&1 | tee $O/modes.txt
SGS_BWD_DBG=16 timeout 200 python tools/bwd_phases.py 0 2>&1 | tee $O/phases0.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bwd -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py /tmp/prof/bwd_results.db 2>&1 | head -9 | cut -c1-150 | tee $O/kernel_stats.txt
