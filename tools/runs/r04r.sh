#!/bin/bash
# round 4, GPU call R: which weights pre-pass kernel / hand-over format costs what (cfg3, one view in flight)
#   0x36 super-batch kernel, three terms (default) | 0x8036 16-entry-batch kernel, three terms | 0xC036 two-pixels-per-lane kernel, three terms
#   0x38 16-entry-batch kernel, two terms (1 KB per entry) | 0x3B two-pixels-per-lane kernel, fp32 rows | 0x403B 16-entry-batch kernel, fp32 rows
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04r; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sweep2_gpu.py -q -m gpu -x -k "superbatch" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 400 python tools/exp_r03_sweep2.py 0x36 0x8036 0xC036 0x38 0x3B 0x403B 0x36 0x8036 0xC036 0x38 0x3B 0x403B > $O/timing.txt 2>&1; grep frame $O/timing.txt
