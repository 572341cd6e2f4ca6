#!/bin/bash
# round 5, GPU call B: why the fused backward takes 13 ms -- LDS reduction microbenchmark + ablations of the kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout 60 tools/ubench_ldsadd 2>&1 | tee $O/ubench_ldsadd.txt
for d in 1 2 4 8 15; do echo "SGS_BWD_DBG=$d"; SGS_BWD_DBG=$d timeout 120 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode; done | tee $O/ablations.txt
