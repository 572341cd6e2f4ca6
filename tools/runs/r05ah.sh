#!/bin/bash
# round 5, GPU call AH: views in flight (the headline keeps 4)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05ah; mkdir -p $O
for v in 4 2 3 6 8 4; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --views $v > $O/v$v.json 2> $O/v$v.err
python - <<PY
import json
d=json.loads(open('$O/v$v.json').read().strip().splitlines()[-1])
print('views=$v', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'per view', round(d['ms_per_step']/$v,4), d.get('memory',{}).get('max_allocated_GB'), d.get('integrity'))
PY
done
