#!/bin/bash
# round 4, GPU call ZO: the whole -m gpu suite + smoke on the last commit of the round
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zo; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['frac'], d['roofline']['kernels_ms'], d['single_view']['ms_median'])"
