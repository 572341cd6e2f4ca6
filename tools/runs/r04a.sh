#!/bin/bash
# round 4, GPU call A: the x16 question (library GEMM as the aggressor) + the sweep at one / two workgroups per CU
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/x16_gemm_aggressor.py 20000 4000 > $O/x16_gemm.txt 2>&1
tail -8 $O/x16_gemm.txt
V="0x6E 0x6E 0x16E 0x26E 0x36E 0x6F 0x16F 0x6B 0x67 0x6E"
timeout 300 python tools/exp_r03_sweep2.py $V > $O/occ2.txt 2>&1
SGS_DEBUG_SWEEP_DYNLDS=81920 timeout 300 python tools/exp_r03_sweep2.py $V > $O/occ1.txt 2>&1
echo "== two workgroups per CU"; cat $O/occ2.txt
echo "== one workgroup per CU"; cat $O/occ1.txt
