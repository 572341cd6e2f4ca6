#!/bin/bash
# round 4, GPU call ZC: backward -- nt hint on the D rows (dot kernel -> geometry kernel) and on the folded clear of dL/dcolor: variant build vs the tree's, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zc; mkdir -p $O
export TMPDIR=/tmp
L=semantic-gaussians_amd/sgs_hip/libsgs_hip.so
cp $L /tmp/lib_def.so
for v in def bnt def bnt; do
  if [ $v = def ]; then cp /tmp/lib_def.so $L; else cp gpurun_in/libsgs_hip_$v.so $L; fi
  timeout 200 python tools/bench_bwd_cfg3.py 20 2>&1 | grep cfg3 | sed "s/^/$v  /" | tee -a $O/timing.txt
done
cp /tmp/lib_def.so $L
timeout 300 python tools/exp_r03_sweep2.py 0x36 0xC036 0x36 0xC036 2>&1 | grep frame | tee -a $O/timing.txt
