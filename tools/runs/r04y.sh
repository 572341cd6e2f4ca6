#!/bin/bash
# round 4, GPU call Y: front end with stage A's tables written by the sort's last pass and stage A's three single-workgroup steps in one kernel (20 -> 17 launches)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04y; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py tests/test_stress_gpu.py tests/test_configs_gpu.py tests/test_abi.py -q -m gpu -x --timeout=900 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 300 python tools/exp_r03_sweep2.py 0x36 0x36 0x36 0x36 > $O/timing.txt 2>&1; grep frame $O/timing.txt
cd /tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --views 1 --fixed-camera --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $GRAFT_REPO_ROOT/$O/bench_views1.json 2> $GRAFT_REPO_ROOT/$O/bench_views1.err
cd $GRAFT_REPO_ROOT; db=$(find /tmp/kt -name "*results.db" | head -1); python tools/frame_timeline.py $db 3 > $O/frame_timeline.txt; cat $O/frame_timeline.txt
