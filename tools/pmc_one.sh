#!/bin/bash
# one rocprofv3 PMC pass (counters only) over the SpMV probe: tools/pmc_one.sh "<counters>" [probe args]
set -u
R=/root/repo; CNT=$1; shift
export TMPDIR=/tmp; rm -rf /tmp/pmc1; mkdir -p /tmp/pmc1; cd /tmp
timeout 300 rocprofv3 --pmc $CNT -d /tmp/pmc1/g1 -o p -- python3 $R/tools/spmv_probe.py --reps 3 "$@" > /tmp/pmc1/log 2>&1
grep '"ms"' /tmp/pmc1/log | tail -1 | cut -c1-200
python3 $R/tools/pmc_summary.py /tmp/pmc1
