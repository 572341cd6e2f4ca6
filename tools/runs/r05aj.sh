#!/bin/bash
# round 5, GPU call AJ: the final tree once more (geometry-walk hit mask, fence-less timer events, band guard) -- whole -m gpu suite, smoke, the evidence bundle
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05aj; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 bash tools/profile_round.sh r05aj/prof | tail -3
