#!/bin/bash
# round 4, GPU call P: the weights pre-pass with 256-entry super-batches (blend_weights_sb_kernel) against the 16-entry-batch kernel (bit 15)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04p; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sweep2_gpu.py -q -m gpu -x -k "superbatch or shapes or background or ping_pong" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 300 python tools/exp_r03_sweep2.py 0x36 0x8036 0x36 0x8036 0x36 0x8036 0x3B 0x803B > $O/timing.txt 2>&1; grep frame $O/timing.txt
