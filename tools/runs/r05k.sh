#!/bin/bash
# round 5, GPU call K: fused backward -- overflow word off the critical path; s_setprio around the matrix phases (A/B); phases
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05k; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "backward" --timeout=600 2>&1 | tail -2
for r in 1 2 3; do
echo "default"; timeout 100 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
echo "setprio"; SGS_BWD_DBG=32 timeout 100 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
done | tee $O/setprio.txt
timeout 100 python tools/bench_bwd_modes.py 4 2>&1 | grep backward_mode
SGS_BWD_DBG=16 timeout 200 python tools/bwd_phases.py 0 2>&1 | tee $O/phases0.txt | grep -i "prologue\|per iteration"
