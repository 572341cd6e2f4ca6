#!/bin/bash
# round 4, GPU call ZN: the default pre-pass at 4 instead of 5 waves per SIMD (128 VGPRs) -- variant build vs the tree's, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zn; mkdir -p $O
export TMPDIR=/tmp
L=semantic-gaussians_amd/sgs_hip/libsgs_hip.so
cp $L /tmp/lib_def.so
for v in def w4 def w4; do
  if [ $v = def ]; then cp /tmp/lib_def.so $L; else cp gpurun_in/libsgs_hip_$v.so $L; fi
  echo "== $v" | tee -a $O/timing.txt
  timeout 200 python tools/exp_r03_sweep2.py 0x36 0x36 0x36 2>&1 | grep frame | tee -a $O/timing.txt
done
cp /tmp/lib_def.so $L
