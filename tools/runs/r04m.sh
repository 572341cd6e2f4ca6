#!/bin/bash
# round 4, GPU call M: the backward's two products on two streams (D = F G on its own rows): A/B against SGS_BWD_SERIAL=1, gradient tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/bench_bwd_modes.py 0 0 3 > $O/two_streams.txt 2>&1; grep backward_mode $O/two_streams.txt
SGS_BWD_SERIAL=1 timeout 200 python tools/bench_bwd_modes.py 0 0 3 > $O/serial.txt 2>&1; echo "-- SGS_BWD_SERIAL=1"; grep backward_mode $O/serial.txt
timeout 200 python tools/bench_bwd_modes.py 0 0 > $O/two_streams2.txt 2>&1; grep backward_mode $O/two_streams2.txt
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_ref_splat.py tests/test_training_loop.py tests/test_configs_gpu.py -q -m gpu -x --timeout=600 -k "backward or grad or train or bwd or densif or worklist" > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.txt
cd /tmp && rm -rf /tmp/kb && rocprofv3 --kernel-trace --stats -d /tmp/kb -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > /dev/null 2>&1
db=$(find /tmp/kb -name "*results.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db sgs:: > $GRAFT_REPO_ROOT/$O/bwd_kernel_stats.txt 2>&1; head -12 $GRAFT_REPO_ROOT/$O/bwd_kernel_stats.txt
