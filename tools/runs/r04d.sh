#!/bin/bash
# round 4, GPU call D: ping-pong sweep with its DMA pieces inside the MFMA phase and the arrival check's reads hidden
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/exp_r03_sweep2.py 0x6E 0x66 0x6E 0x66 0x36 0x26 0x46 0x166 0x266 0x66 0x6E > $O/timing.txt 2>&1; grep frame $O/timing.txt
timeout 200 python tools/sweep_phases.py 0x466 > $O/phases.txt 2>&1; cat $O/phases.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_sweep2_gpu.py tests/test_parity_gpu.py tests/test_multigpu.py -q -m gpu -x --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
