#!/bin/bash
# round 4, GPU call ZB: nt hint on every work-list weight store (all three pre-pass kernels): parity tests, the formats' timing, the backward
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zb; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sweep2_gpu.py tests/test_parity_gpu.py tests/test_training_loop.py -q -m gpu -x --timeout=900 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python tools/exp_r03_sweep2.py 0x36 0x8036 0xC036 0x38 0x3B 0x403B 0x36 0x8036 0xC036 0x38 0x3B 0x403B > $O/timing.txt 2>&1; grep frame $O/timing.txt
for i in 1 2; do timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | tee -a $O/bwd.txt; done
