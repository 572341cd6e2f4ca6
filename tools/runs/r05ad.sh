#!/bin/bash
# round 5, GPU call AD: events without the system-scope fence (SGS_EVENT_NOFENCE=1) -- A/B on the single view and the headline, the frame's launch timeline;
# the empty-shard band test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05ad; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_multigpu.py -q -m gpu -k "band_major" 2>&1 | tail -2
for rep in 1 2; do for nf in 0 1; do
SGS_EVENT_NOFENCE=$nf timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/b_${nf}_$rep.json 2> $O/b_${nf}_$rep.err
python - <<PY
import json
d=json.loads(open('$O/b_${nf}_$rep.json').read().strip().splitlines()[-1])
print('nofence=$nf', round(d['value'],1), round(d['ms_per_step'],3), 'single', round(d['single_view']['ms_median'],4), round(d['single_view']['ms_min'],4), 'inf', round(d.get('single_view_inference',{}).get('ms_median',0),4), 'kern', d['roofline']['kernels_ms'], d.get('integrity'))
PY
done; done
cd /tmp
for nf in 0 1; do
rm -rf /tmp/kt
SGS_EVENT_NOFENCE=$nf rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --views 1 --fixed-camera --no-cpu-baseline --no-extras --steps 60 --warmup 4 > /dev/null 2>&1
db=$(find /tmp/kt -name "*results.db" | head -1)
echo "timeline nofence=$nf"; python $GRAFT_REPO_ROOT/tools/frame_timeline.py $db 3 | tee $GRAFT_REPO_ROOT/$O/timeline_$nf.txt | awk '$NF>0.5 || /kernels/'
done
