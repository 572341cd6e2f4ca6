#!/bin/bash
# round 4, GPU call F: ping-pong sweep with the DMA source addresses computed in PREP (only m0 + the load between the MFMA pairs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/exp_r03_sweep2.py 0x6E 0x36 0x6E 0x36 0x26 0x46 0x66 0x136 0x236 0x36 0x6E > $O/timing.txt 2>&1; grep frame $O/timing.txt
timeout 100 python tools/sweep_phases.py 0x436 > $O/phases.txt 2>&1; grep -v amdgpu.ids $O/phases.txt
timeout 600 python -m pytest tests/test_sweep2_gpu.py -q -m gpu -x --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.txt
