#!/bin/bash
# round 5, GPU call W: fused backward on the double-rate MFMA -- parity (incl. cfg3 vs the oracle), timing, kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05w; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_training_loop.py tests/test_ref_splat.py -q -m gpu -k "backward or training or grad" --timeout=600 -s 2>&1 | grep -E "cfg3 backward vs oracle, mode 0|passed|failed" | tail -12
for r in 1 2; do timeout 100 python tools/bench_bwd_modes.py 0 3 2>&1 | grep backward_mode; done | tee $O/modes.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bwd -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py /tmp/prof/bwd_results.db 2>&1 | head -8 | cut -c1-150 | tee $O/kernel_stats.txt
