#!/bin/bash
# round 5, GPU call J: whole -m gpu suite + smoke + default bench on the fused backward
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05j; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 -x > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05j/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}); print('single', d['single_view']); print('roofline', {k:d['roofline'][k] for k in ('frac','kernels_ms')})
print('backward', d['backward'])
PY
