#!/bin/bash
# round 4, GPU call X: the whole -m gpu suite on the final tree, smoke, the round's evidence (tools/profile_round.sh), a 36 000-forward pipelined soak
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04x; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 bash tools/profile_round.sh r04x > $O/profile_round.log 2>&1
tail -3 $O/bench_default.err; cat $O/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_ranks','memory')})
print('single', d['single_view']); print('api', d['api_path']); print('roofline', {k:d['roofline'][k] for k in ('frac','kernel_ms','kernels_ms','traffic')})
print('backward', d['backward'])"
head -12 $O/kernel_stats_views1.txt
timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 9000 > $O/soak.json 2> $O/soak.err; tail -2 $O/soak.err; python -c "
import json
d=json.loads(open('$O/soak.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d.get('integrity'))"
