#!/bin/bash
# round 5, GPU call V (X16=1 EXPERIMENTS=1 build): the soak on a box whose class is established in the same call -- 0x6F control, 300 000 forwards of the
# default (x16 ping-pong sweep, 256-register waves) on SHARED compute units, 0x6F control again
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05v; mkdir -p $O
python - <<'PY' > /dev/null
PY
sed -e 's/for rep in range(2):/for rep in range(1):/' -e 's/R \/\/ 2, serial/R, serial/g' tools/x16_cu_mask.py > /tmp/x16_one.py; cp /tmp/x16_one.py tools/_x16_one.py
timeout 200 python tools/_x16_one.py 3000 0x6F 4 0 2>&1 | grep -v amdgpu.ids | grep -v "^  \[" | tee $O/control_before.txt
timeout 900 python tools/_x16_one.py 50000 0 4 0 2>&1 | grep -v amdgpu.ids | tee $O/soak_default.txt
timeout 200 python tools/_x16_one.py 3000 0x6F 4 0 2>&1 | grep -v amdgpu.ids | grep -v "^  \[" | tee $O/control_after.txt
rm -f tools/_x16_one.py
