#!/bin/bash
# round 4, GPU call G: ping-pong sweep with transposed 16-byte stores (a tile pair = 32 store instructions instead of 128)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/exp_r03_sweep2.py 0x6E 0x36 0x836 0x6E 0x36 0x836 0x26 0x46 0x66 0x136 0x236 0x36 0x6E > $O/timing.txt 2>&1; grep frame $O/timing.txt
timeout 100 python tools/sweep_phases.py 0x436 > $O/phases.txt 2>&1; grep -v amdgpu.ids $O/phases.txt
timeout 900 python -m pytest tests/test_sweep2_gpu.py tests/test_parity_gpu.py tests/test_stress_gpu.py -q -m gpu -x --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
