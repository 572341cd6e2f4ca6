#!/bin/bash
# round 4, GPU call B: first run of the ping-pong sweep (variant nibble 6, now the default): timing, then the whole -m gpu suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
timeout 240 python tools/exp_r03_sweep2.py 0x6E 0x66 0x6E 0x66 0x36 0x26 0x46 0x166 0x266 0x366 0x66 0x6E > $O/timing.txt 2>&1
echo "timing rc=$?"; cat $O/timing.txt | grep -v amdgpu.ids
timeout 1200 python -m pytest tests -q -m gpu --maxfail=12 --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -40 $O/pytest.txt
