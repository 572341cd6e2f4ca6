#!/bin/bash
# round 4, GPU call E: where do the ping-pong sweep's DMA pieces go (n at the start of PREP, 5 - n between the MFMA pairs), static vs per-phase priority
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/exp_r03_sweep2.py 0x6E 0x36 0x20036 0x30036 0x50036 0x80036 0xA0036 0xB0036 0xD0036 0x36 0x20036 0x30036 0x50036 0x80036 0xA0036 0xB0036 0xD0036 0x6E > $O/timing.txt 2>&1; grep frame $O/timing.txt
timeout 100 python tools/sweep_phases.py 0x30436 > $O/phases_n3.txt 2>&1; grep -v amdgpu.ids $O/phases_n3.txt
timeout 100 python tools/sweep_phases.py 0xB0436 > $O/phases_n3_flip.txt 2>&1; grep -v amdgpu.ids $O/phases_n3_flip.txt
