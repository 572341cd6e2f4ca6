#!/bin/bash
# round 4, GPU call ZM: backward -- nt hint on the gradient loads of the two products (dot: the dcolor kernel only, VEC path): variant build vs the tree's, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zm; mkdir -p $O
export TMPDIR=/tmp
L=semantic-gaussians_amd/sgs_hip/libsgs_hip.so
cp $L /tmp/lib_def.so
for v in def gnt def gnt; do
  if [ $v = def ]; then cp /tmp/lib_def.so $L; else cp gpurun_in/libsgs_hip_$v.so $L; fi
  timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | sed "s/^/$v  /" | tee -a $O/timing.txt
done
cp /tmp/lib_def.so $L
