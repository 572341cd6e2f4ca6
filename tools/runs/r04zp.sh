#!/bin/bash
# round 4, GPU call ZP: sweep3, first half's arrival check issued at the START of its matrix phase instead of under the last four MFMAs (variant build 'ep' vs the tree 'def')
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zp; mkdir -p $O
export TMPDIR=/tmp
L=semantic-gaussians_amd/sgs_hip/libsgs_hip.so
cp $L /tmp/lib_def.so
for v in def ep def ep; do
  if [ $v = def ]; then cp /tmp/lib_def.so $L; else cp gpurun_in/libsgs_hip_$v.so $L; fi
  echo "== $v" | tee -a $O/timing.txt
  timeout 200 python tools/exp_r03_sweep2.py 0x36 0x36 0x36 0x6E 2>&1 | grep frame | tee -a $O/timing.txt
  if [ $v = ep ]; then timeout 300 python -m pytest tests/test_sweep2_gpu.py -q -m gpu -x -k "ping_pong" 2>&1 | tail -1; fi
done
cp /tmp/lib_def.so $L
