#!/bin/bash
# round 4, GPU call S: the backward's kernels take their tiles longest-first by the forward's own work-list lengths (tile_order) -- A/B + the backward tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_training_loop.py tests/test_configs_gpu.py -q -m gpu -x -k "backward or train or grad" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
for i in 1 2; do
  timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | tee -a $O/timing.txt
  SGS_NO_TILE_ORDER=1 timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | tee -a $O/timing.txt
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kb -o runs -- python $GRAFT_REPO_ROOT/tools/bench_bwd_cfg3.py 8 > /tmp/kb.log 2>&1
cd "$GRAFT_REPO_ROOT"; db=$(find /tmp/kb -name "*results.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt
