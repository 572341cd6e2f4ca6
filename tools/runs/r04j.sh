#!/bin/bash
# round 4, GPU call J: the round's evidence on the final code (tools/profile_round.sh) + a pipelined soak
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 bash tools/profile_round.sh r04j > $O/profile_round.log 2>&1
tail -3 $O/bench_default.err; python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','memory')})
print('single', d['single_view']['ms_median'], 'api', d['api_path']['ms_median'], d['api_path']['ratio_to_single_view'])
print('roofline', {k:d['roofline'][k] for k in ('frac','kernel_ms','kernels_ms','traffic')})
print('backward', d['backward']['fwd_bwd_ms_median'], d['backward']['backward_ms_median'], d['backward']['roofline']['frac'])
print('exact', d['exact_f32']); print('two_term', d['two_term'])"
head -6 $O/kernel_stats_views1.txt
timeout 400 python tools/stress_pipelined.py 2000 > $O/stress.txt 2>&1; tail -6 $O/stress.txt
