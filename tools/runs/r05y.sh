#!/bin/bash
# round 5, GPU call Y: the evidence bundle on the round's tree (x16 ping-pong sweep + x16 fused backward as defaults): profile_round + backward PMC
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05y; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 bash tools/profile_round.sh r05y > $O/profile_round.log 2>&1
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05y/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','memory')}); print('single', {q:d['single_view'][q] for q in ('ms_median','stage_ms')}, 'inference', d['single_view_inference']['ms_median'])
print('api', {k:d['api_path'][k] for k in ('ms_median','ratio_to_single_view','ms_median_with_debug_false')}); print('roofline', {k:d['roofline'][k] for k in ('frac','kernels_ms','traffic')})
print('backward', {k:d['backward'][k] for k in ('fwd_bwd_ms_median','backward_ms_median')}, d['backward']['roofline']['frac']); print('cpu', d.get('cpu_baseline'))
PY
head -12 $O/kernel_stats_views1.txt | cut -c1-140
cat $O/frame_timeline.txt | tail -26
head -14 $O/backward_kernel_stats.txt | cut -c1-140
timeout 500 bash tools/pmc_kernel.sh r05y/backward "bwd_" python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > $O/pmc.log 2>&1
grep -A3 "bwd_fused" $O/backward_pmc.txt | grep -E "FETCH|WRITE|MFMA" | head
