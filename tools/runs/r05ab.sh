#!/bin/bash
# round 5, GPU call AB: product build with the free-running x16 sweep as the default -- whole -m gpu suite, smoke, bench, soak
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05ab; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05ab/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('integrity'))
for k in ('single_view','single_view_inference'): print(k, {q:d[k][q] for q in ('value','ms_median','ms_min')})
print('api', {k:d['api_path'][k] for k in ('ms_median','ratio_to_single_view','ms_median_with_debug_false')})
print('roofline', {k:d['roofline'][k] for k in ('frac','kernels_ms')}); print('backward', {k:d['backward'][k] for k in ('fwd_bwd_ms_median','backward_ms_median')}, d['backward']['roofline']['frac'])
PY
timeout 600 python tools/x16_cu_mask.py 25000 0 4 0 2>&1 | grep -v amdgpu.ids | tee $O/soak_small.txt
timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 9000 > $O/soak.json 2> $O/soak.err; python -c "
import json
d=json.loads(open('$O/soak.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d.get('integrity'))" | tee $O/soak_cfg3.txt
