#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# The round's evidence in one go, everything into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   bench_default.json/.err            the default bench line (what the driver runs)
#   bench_views1_under_rocprof.json    bench.py --views 1 --fixed-camera under rocprofv3 --kernel-trace --stats
#   kernel_stats_views1.txt            its per-kernel summary (avg duration of the dominant kernels = roofline.kernels_ms)
#   frame_timeline.txt                 one forward of that run as its launch sequence with gaps
#   backward_kernel_stats.txt          forward + backward (tools/bench_backward.py) per-kernel summary
#   blend_pmc.txt                      PMC passes (counters only, separate runs) of the forward
#   calib.txt                          the known-byte calibration kernels under the same counters
#   blend_traffic.json                 calibrated fabric bytes of the two blend kernels (-> profiles/blend_traffic.json)
tag=${1:-prof}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python bench.py 2> $O/bench_default.err > $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --views 1 --fixed-camera --no-cpu-baseline --no-extras --steps 600 --warmup 4 \
	> $O/bench_views1_under_rocprof.json 2> $O/bench_views1_under_rocprof.err
db=$(find /tmp/kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $db > $O/kernel_stats_views1.txt
python $R/tools/frame_timeline.py $db 3 > $O/frame_timeline.txt
rm -rf /tmp/kb
rocprofv3 --kernel-trace --stats -d /tmp/kb -- python $R/tools/bench_backward.py > $O/bench_backward.log 2>&1
db=$(find /tmp/kb -name "*results.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db sgs:: > $O/backward_kernel_stats.txt
bash $R/tools/pmc_pass.sh $tag/blend --views 1 --fixed-camera --no-extras > /dev/null 2>&1
bash $R/tools/calib/run_calib.sh $O/calib.txt > /dev/null 2>&1
python $R/tools/make_blend_traffic.py $O/blend_pmc.txt $O/calib.txt cfg3 > $O/blend_traffic.json 2> $O/blend_traffic.err
ls -la $O
