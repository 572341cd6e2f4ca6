#!/bin/bash
# round 4, GPU call T: backward with the tile order on its PRE-PASS only (A/B against SGS_NO_TILE_ORDER=1) + the backward tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04t; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_training_loop.py tests/test_configs_gpu.py -q -m gpu -x -k "backward or train or grad" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
for i in 1 2; do
  timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | tee -a $O/timing.txt
  SGS_NO_TILE_ORDER=1 timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | tee -a $O/timing.txt
done
