#!/bin/bash
# round 4, GPU call ZF: does the tile-order feedback help the HEADLINE (each view slot cycles through 8 cameras, so a stream's previous frame is another view)?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zf; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order on ', round(d['value'],1), round(d['ms_per_step'],3), d['stage_ms_timed_region'])" | tee -a $O/timing.txt
  SGS_NO_TILE_ORDER=1 python bench.py --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order off', round(d['value'],1), round(d['ms_per_step'],3), d['stage_ms_timed_region'])" | tee -a $O/timing.txt
done
