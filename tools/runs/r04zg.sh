#!/bin/bash
# round 4, GPU call ZG: the pre-pass reading point_list with the nt hint (+ nt on final_T / n_contrib) -- variant build vs the tree's, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zg; mkdir -p $O
export TMPDIR=/tmp
L=semantic-gaussians_amd/sgs_hip/libsgs_hip.so
cp $L /tmp/lib_def.so
for v in def pl def pl; do
  if [ $v = def ]; then cp /tmp/lib_def.so $L; else cp gpurun_in/libsgs_hip_$v.so $L; fi
  echo "== $v" | tee -a $O/timing.txt
  timeout 200 python tools/exp_r03_sweep2.py 0x36 0x36 0x36 2>&1 | grep frame | tee -a $O/timing.txt
done
cp /tmp/lib_def.so $L
