#!/bin/bash
# round 5, GPU call I: fused backward, tiles longest-first (the forward's order) against XCD bands; 3 repeats each
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05i; mkdir -p $O
for r in 1 2 3; do
echo "bands"; timeout 100 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
echo "longest first"; SGS_BWD_ORDER=1 timeout 100 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
done | tee $O/order.txt
