#!/bin/bash
# round 4, GPU call X3: the final tree (two-pixels-per-lane super-batch pre-pass, nt work-list stores, front-end merges): -m gpu suite, smoke, the evidence bundle, a 36 000-forward soak
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04x3; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 bash tools/profile_round.sh r04x3 > $O/profile_round.log 2>&1
tail -3 $O/bench_default.err; cat $O/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_ranks','memory')})
print('single', d['single_view']); print('api', {k:d['api_path'][k] for k in ('ms_median','ratio_to_single_view','ms_median_with_debug_false')}); print('roofline', {k:d['roofline'][k] for k in ('frac','kernel_ms','kernels_ms','traffic')})
print('backward', {k:d['backward'][k] for k in ('fwd_bwd_ms_median','backward_ms_median')})"
head -8 $O/kernel_stats_views1.txt
timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 9000 > $O/soak.json 2> $O/soak.err; tail -2 $O/soak.err; python -c "
import json
d=json.loads(open('$O/soak.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d.get('integrity'))"
