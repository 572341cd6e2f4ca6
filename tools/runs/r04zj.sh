#!/bin/bash
# round 4, GPU call ZJ: the new default pre-pass (two pixels per lane, super-batches): sweep2 / parity / stress / configs tests + the kernel matrix
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zj; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sweep2_gpu.py tests/test_parity_gpu.py tests/test_stress_gpu.py tests/test_configs_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x --timeout=900 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python tools/exp_r03_sweep2.py 0x36 0x4036 0x8036 0xC036 0x36 0x4036 0x8036 0xC036 > $O/timing.txt 2>&1; grep frame $O/timing.txt
