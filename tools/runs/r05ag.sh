#!/bin/bash
# round 5, GPU call AG: the free-running x16 sweep's segment length / workgroup order (the defaults were tuned on x8)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05ag; mkdir -p $O
timeout 600 python tools/exp_r03_sweep2.py 0x10004 0x10004 0x10024 0x10044 0x10064 0x10004 0x11004 0x12004 0x13004 0x10024 0x10044 0x10064 0x10004 0x10014 0x10054 2>&1 | grep -v amdgpu.ids | tee $O/seg.txt
