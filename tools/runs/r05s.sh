#!/bin/bash
# round 5, GPU call S (X16=1 EXPERIMENTS=1 build): box class by the 0x6F control, then the x16 ping-pong sweep (whole-CU workgroups) as the aggressor; its speed
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05s; mkdir -p $O
timeout 300 python tools/x16_cu_mask.py 6000 0x6F 4 0 2>&1 | grep -v amdgpu.ids | grep -v "^  \[" | tee $O/control_6f.txt
timeout 600 python tools/x16_cu_mask.py 24000 0x10066 4 0 2>&1 | grep -v amdgpu.ids | tee $O/pingpong_x16.txt
for v in 0 65638 0 65638; do python bench.py --variant $v --no-cpu-baseline --no-extras --steps 120 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant', $v, 'value', round(d['value'],1), 'single', round(d['single_view']['ms_median'],4), 'roofline', round(d['roofline']['frac'],4), d['roofline']['kernels_ms'], 'mismatches', d['integrity'])"; done | tee $O/speed.txt
