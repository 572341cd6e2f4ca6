#!/bin/bash
# round 4, GPU call ZK: the default bench line once more on another box (range of the headline on the final tree)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zk; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'single', d['single_view']['ms_median'], 'api', d['api_path']['ms_median'], 'roofline', d['roofline']['frac'], d['roofline']['kernels_ms'], 'bwd', d['backward']['backward_ms_median'], 'exact', d['exact_f32']['value'], 'two_term', d['two_term']['value'], 'deferred', d['deferred_count']['value'])"
