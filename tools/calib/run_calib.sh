#!/bin/bash
# usage: tools/calib/run_calib.sh <outfile>  -- one rocprofv3 --pmc pass per (kernel, counter set); counters only, no tracing
out=${1:-$GRAFT_REPO_ROOT/gpurun_out/calib.txt}
cd /tmp && export TMPDIR=/tmp
: > $out
for k in 0 1 2 3 4; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
    rm -rf /tmp/cal
    timeout 120 rocprofv3 --pmc $set -d /tmp/cal -- $GRAFT_REPO_ROOT/tools/calib/calib_counters $k > /tmp/cal.log 2>&1
    db=$(find /tmp/cal -name "*results.db" | head -1)
    echo "## kernel $k  set: $set   ($(grep 'per launch' /tmp/cal.log))" >> $out
    [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db k_ 2>/dev/null | grep -v "^kernel\|^$" >> $out
  done
done
cat $out
