#!/bin/bash
# round 4, GPU call ZL: the library rebuilt from scratch (make clean; __graft_entry__.build()): smoke + the sweep / parity files + one bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zl; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_sweep2_gpu.py tests/test_parity_gpu.py tests/test_abi.py -q -m gpu -x --timeout=600 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['frac'], d['roofline']['kernels_ms'], d['single_view']['ms_median'])"
