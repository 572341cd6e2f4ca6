#!/bin/bash
# round 5, GPU call AI: the whole -m gpu suite on the `make EXPERIMENTS=1` build of the final tree (the development forms the product build skips)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05ai; mkdir -p $O
export TMPDIR=/tmp
python -c "
import sys; sys.path.insert(0,'semantic-gaussians_amd')
from sgs_hip import raster; print('build flags', raster.build_flags())"
timeout 1500 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.txt
