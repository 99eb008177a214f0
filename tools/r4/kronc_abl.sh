#!/bin/bash
# QBH_KRON_CODED=2 on C3: ablations of the two passes (wrong results by design): 1 no dictionary lookups, 2 no gathers, 4 no row sums
mkdir -p gpurun_out/r4kronc
O=$GRAFT_REPO_ROOT/gpurun_out/r4kronc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for abl in 0 32; do
  export QBH_KRON_CODED=2 QBH_KRONC_ABL=$abl
  rm -rf /tmp/kp; mkdir -p /tmp/kp
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kp -o k -- python3 $R/bench.py --format fast --steps 10 --warmup 2 --no-converge --no-cpu-baseline --no-matrix-free > /tmp/kp/log 2>&1
  echo "== abl $abl" | tee -a $O/abl.txt
  python3 $R/tools/stats_summary.py /tmp/kp "abl" | grep -E "kronc_far|kronc_near" | tee -a $O/abl.txt
done
