#!/bin/bash
# round 5, GPU call N: upper half's W g^T and slab take-over as ONE basic block (VALU / LDS issued between the products)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "backward" --timeout=600 2>&1 | tail -2
for r in 1 2 3; do timeout 100 python tools/bench_bwd_modes.py 0 4 2>&1 | grep backward_mode; done | tee $O/modes.txt
SGS_BWD_DBG=16 timeout 200 python tools/bwd_phases.py 0 2>&1 | tee $O/phases0.txt | head -26
