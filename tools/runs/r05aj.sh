#!/bin/bash
# round 5, GPU calls AJ / AK: the final tree (AK: after the free-running sweep got its 256th register pinned) -- whole -m gpu suite, smoke, the evidence bundle
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05ak; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 bash tools/profile_round.sh r05ak/prof | tail -3
