#!/bin/bash
# round 4, GPU call U: the backward's two products + the geometry kernel walking the tile grid in 8 x 8 blocks (A/B against SGS_BWD_TILE_PERM=0)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04u; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_training_loop.py tests/test_configs_gpu.py -q -m gpu -x -k "backward or train or grad" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
for i in 1 2; do
  timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | sed 's/^/perm on  /' | tee -a $O/timing.txt
  SGS_BWD_TILE_PERM=0 timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | sed 's/^/perm off /' | tee -a $O/timing.txt
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kb -o runs -- python $GRAFT_REPO_ROOT/tools/bench_bwd_cfg3.py 8 > /tmp/kb.log 2>&1
cd "$GRAFT_REPO_ROOT"; db=$(find /tmp/kb -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/kernel_stats.txt 2>&1; head -8 $O/kernel_stats.txt
