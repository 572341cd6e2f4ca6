#!/bin/bash
# round 5, GPU call AL: soak of the final binary (free-running x16 sweep at 256 registers per wave) -- pipelined small forwards on shared CUs, cfg3 four in flight
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05al; mkdir -p $O
timeout 400 python tools/x16_cu_mask.py 25000 0 4 0 2>&1 | grep -v amdgpu.ids | tee $O/soak_small.txt
timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 9000 > $O/soak.json 2> $O/soak.err; python -c "
import json
d=json.loads(open('$O/soak.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d.get('integrity'))" | tee $O/soak_cfg3.txt
