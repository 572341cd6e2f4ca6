#!/bin/bash
# round 4, GPU call C: phase clocks of the ping-pong sweep; tile-order feedback A/B for the weights pre-pass; the fixed gloo test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/sweep_phases.py 0x466 > $O/phases.txt 2>&1; cat $O/phases.txt | grep -v amdgpu.ids
timeout 200 python tools/exp_r03_sweep2.py 0x6E 0x66 0x6E 0x66 > $O/timing_order.txt 2>&1; grep frame $O/timing_order.txt
SGS_NO_TILE_ORDER=1 timeout 200 python tools/exp_r03_sweep2.py 0x6E 0x66 0x6E 0x66 > $O/timing_noorder.txt 2>&1; echo "-- SGS_NO_TILE_ORDER=1"; grep frame $O/timing_noorder.txt
timeout 600 python -m pytest tests/test_multigpu.py tests/test_parity_gpu.py -q -m gpu -x --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
