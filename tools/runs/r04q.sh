#!/bin/bash
# round 4, GPU call Q: split-tail sweep plan (bits [22:20]: eighths of the segments handed out as halves; 7 = none) and the super-batch weights pre-pass
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sweep2_gpu.py -q -m gpu -x -k "superbatch or split_tail or shapes or ping_pong" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 400 python tools/exp_r03_sweep2.py 0x700036 0x100036 0x200036 0x400036 0x600036 0x700036 0x100036 0x200036 0x400036 0x600036 0x700036 0x200036 0x400036 > $O/timing.txt 2>&1; grep frame $O/timing.txt
