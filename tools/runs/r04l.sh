#!/bin/bash
# round 4, GPU call L: the whole -m gpu suite + smoke on the final tree; views in flight 2 / 4 / 6 / 8
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for v in 2 4 6 8; do
  timeout 200 python bench.py --views $v --no-cpu-baseline --no-extras --steps 150 > $O/bench_v$v.json 2> $O/bench_v$v.err
  python -c "
import json
d=json.loads(open('$O/bench_v$v.json').read().strip().splitlines()[-1])
print('views', $v, 'value', round(d['value'],1), 'ms/view', round(d['ms_per_view'],4), 'mem', d['memory']['max_allocated_GB'])"
done
