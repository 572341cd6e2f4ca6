#!/bin/bash
# round 5, GPU call T: product build with the x16 ping-pong sweep as the default -- whole -m gpu suite, smoke, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05t; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05t/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('integrity'))
for k in ('single_view','single_view_inference'): print(k, {q:d[k][q] for q in ('value','ms_median','ms_min')})
print('api', {k:d['api_path'][k] for k in ('ms_median','ratio_to_single_view','ratio_to_single_view_inference','ms_median_with_debug_false')})
for k in ('classic_count','deferred_count','exact_f32','two_term'): print(k, {q:d[k][q] for q in ('value','ms_per_step','num_rendered_mismatches_vs_serial')})
print('roofline', {k:d['roofline'][k] for k in ('frac','kernels_ms')}); print('backward', {k:d['backward'][k] for k in ('fwd_bwd_ms_median','backward_ms_median')}, d['backward']['roofline']['frac'])
PY
