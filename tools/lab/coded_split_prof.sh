#!/bin/bash
# per-launch time and HBM fetch of the coded Kronecker split (row kernel) on the headline operator's fast path
set -u
R=/root/repo
export TMPDIR=/tmp
cd /tmp
ARGS="--format fast --steps 10 --warmup 2 --no-cpu-baseline --no-matrix-free --no-converge"
for m in 0 1; do
  export QBH_KRON_CODED=$m
  rm -rf /tmp/cs$m; mkdir -p /tmp/cs$m
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cs$m/stats -o s -- python3 $R/bench.py $ARGS > /tmp/cs$m/log 2>&1
  python3 $R/tools/stats_summary.py /tmp/cs$m/stats "coded split $m" | grep "k_spmv_rows\|k_kron_tile\|k_axpy"
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/cs$m/g1 -o p -- python3 $R/bench.py $ARGS > /tmp/cs$m/g1.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/cs$m "%k_spmv_rows%"
done
