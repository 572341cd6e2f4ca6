#!/bin/bash
# round 5, GPU calls AE / AF: the backward's geometry walk -- (AE) two pixels per lane (bwd_geom2_kernel, removed), (AF) per-wave hit mask; backward parity tests, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05ae; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_training_loop.py tests/test_ref_splat.py -q -m gpu -k "backward or grad or train or densif" --timeout=600 2>&1 | tail -4
for rep in 1 2; do for g in 1 0; do
echo "SGS_BWD_GEOM_NOFILTER=$g"; SGS_BWD_GEOM_NOFILTER=$g timeout 200 python tools/bench_bwd_modes.py 0 2>&1 | grep backward_mode
done; done
cd /tmp; rm -rf /tmp/kb
rocprofv3 --kernel-trace --stats -d /tmp/kb -- python $GRAFT_REPO_ROOT/tools/bench_bwd_modes.py 0 > /dev/null 2>&1
db=$(find /tmp/kb -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db sgs:: | grep -i "bwd\|weights2" | tee $GRAFT_REPO_ROOT/$O/kernels.txt
