#!/bin/bash
# round 5, GPU call Z: the backward's pre-pass with the super-batch walk (fp32 rows) against round 3's 16-entry-batch kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05z; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_training_loop.py tests/test_ref_splat.py -q -m gpu -k "backward or training or grad" --timeout=600 2>&1 | tail -3
for r in 1 2 3; do
echo "super-batch"; timeout 100 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
echo "batch16"; SGS_BWD_PREPASS=3 timeout 100 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
done | tee $O/ab.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bwd -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py /tmp/prof/bwd_results.db 2>&1 | head -7 | cut -c1-150 | tee $O/kernel_stats.txt
