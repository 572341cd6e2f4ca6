#!/bin/bash
# round 4, GPU call ZQ: the sweep output stores with nt (default) / sc1 nt / sc0 nt -- three builds of the library, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zq; mkdir -p $O
export TMPDIR=/tmp
L=semantic-gaussians_amd/sgs_hip/libsgs_hip.so
cp $L /tmp/lib_nt.so
for v in nt sc1nt sc0nt nt sc1nt sc0nt; do
  if [ $v = nt ]; then cp /tmp/lib_nt.so $L; else cp gpurun_in/libsgs_hip_$v.so $L; fi
  echo "== stores: $v" | tee -a $O/timing.txt
  timeout 200 python tools/exp_r03_sweep2.py 0x36 0x36 0x36 0x6E 2>&1 | grep frame | tee -a $O/timing.txt
done
cp /tmp/lib_nt.so $L
