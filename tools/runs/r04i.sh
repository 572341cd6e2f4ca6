#!/bin/bash
# round 4, GPU call I: SGPR-base LDS-DMA for the weight / id pieces; memory footprint after dropping the unused sort room
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/exp_r03_sweep2.py 0x6E 0x36 0x6E 0x36 0x6E 0x36 0x136 0x236 0x36 0x6E > $O/timing.txt 2>&1; grep frame $O/timing.txt
timeout 100 python tools/sweep_phases.py 0x436 > $O/phases.txt 2>&1; grep -v amdgpu.ids $O/phases.txt
timeout 600 python -m pytest tests/test_sweep2_gpu.py tests/test_parity_gpu.py tests/test_multigpu.py -q -m gpu -x --timeout=600 > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','memory')}); print(d['single_view']); print(d['roofline']['frac'], d['roofline']['kernels_ms'])"
