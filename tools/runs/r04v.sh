#!/bin/bash
# round 4, GPU call V: ping-pong sweep -- segment length (nibble [7:4] x 8 tiles) and workgroup order (bits [13:12]: 1 row-major bands, 2 dealt unsorted, 0 / 3 sorted + dealt)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04v; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python tools/exp_r03_sweep2.py 0x36 0x26 0x46 0x1036 0x2036 0x16 0x36 0x26 0x46 0x1036 0x2036 0x16 > $O/timing.txt 2>&1; grep frame $O/timing.txt
