#!/bin/bash
# round 5, GPU call U: the soak the verdict prescribes for the x16 default -- box class by the 0x6F control needs the X16 build, so here: the PRODUCT build,
# 300 000 pipelined small forwards (4 in flight, default = x16 ping-pong sweep) against the serial render + 36 000 cfg3 forwards with num_rendered integrity
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05u; mkdir -p $O
timeout 300 python -m pytest tests/test_stress_gpu.py tests/test_sweep2_gpu.py -q -m gpu --timeout=600 2>&1 | tail -3
timeout 900 python tools/x16_cu_mask.py 50000 0 4 0 2>&1 | grep -v amdgpu.ids | tee $O/soak_small.txt
timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 9000 > $O/soak.json 2> $O/soak.err; tail -1 $O/soak.err; python -c "
import json
d=json.loads(open('$O/soak.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d.get('integrity'))" | tee $O/soak_cfg3.txt
