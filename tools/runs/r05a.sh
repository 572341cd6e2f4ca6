#!/bin/bash
# round 5, GPU call A: the fused backward (one read of the gradient) -- parity tests, then fwd+bwd per mode
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_training_loop.py -q -m gpu -k "backward or training or grad" --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.txt
timeout 300 python tools/bench_bwd_modes.py 4 0 5 3 2>&1 | tee $O/modes.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bwd -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 4 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
