#!/bin/bash
# round 4, GPU call N: per-workgroup timeline of the ping-pong sweep and of the (tile-ordered) weights pre-pass; a 6 000-forward soak of the headline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/sweep_trace.py 0 sweep > $O/trace_sweep.txt 2>&1; grep -v amdgpu.ids $O/trace_sweep.txt
timeout 200 python tools/sweep_trace.py 0 weights > $O/trace_weights.txt 2>&1; grep -v amdgpu.ids $O/trace_weights.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 1500 > $O/soak.json 2> $O/soak.err
tail -3 $O/soak.err; python -c "
import json
d=json.loads(open('$O/soak.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median')}, d['integrity'])"
