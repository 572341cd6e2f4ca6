#!/bin/bash
# round 5, GPU call O: merged (W g^T + slab take-over in one block) against the two-phase form, same box, alternating; PMC bytes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05o; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2 3 4; do
echo merged; timeout 100 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
echo two-phase; SGS_BWD_DBG=32 timeout 100 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
done | tee $O/ab.txt
echo legacy; timeout 100 python tools/bench_bwd_modes.py 4 2>&1 | grep backward_mode
timeout 600 bash tools/pmc_kernel.sh r05o/backward "bwd_" python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 4 > $O/pmc.log 2>&1
grep -E "==|FETCH_SIZE|WRITE_SIZE" gpurun_out/r05o/backward_pmc.txt | grep -B1 -E "FETCH|WRITE" | cut -c1-110
