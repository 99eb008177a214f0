#!/bin/bash
# QBH_KRON_CODED=2 on C3: chunk of the far pass (groups per wavefront turn)
mkdir -p gpurun_out/r4kronc
O=$GRAFT_REPO_ROOT/gpurun_out/r4kronc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "1 0" "1 4" "1 16" "1 32" "2 0" "2 8"; do
  set -- $cfg
  export QBH_KRON_CODED=2 QBH_KRONC_FAR_NG=$1 QBH_KRONC_FAR_CHUNK=$2
  rm -rf /tmp/kp; mkdir -p /tmp/kp
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kp -o k -- python3 $R/bench.py --format fast --steps 20 --warmup 3 --no-cpu-baseline --no-matrix-free > /tmp/kp/log 2>&1
  echo "== far_ng $1 far_chunk $2" | tee -a $O/tune2.txt
  python3 $R/tools/stats_summary.py /tmp/kp "tune" | grep -E "kronc_far|kronc_near" | tee -a $O/tune2.txt
  grep '"metric"' /tmp/kp/log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s', d['value'], 'spmv', d['roofline']['ms_per_launch'], 'e0', d['e0'])" | tee -a $O/tune2.txt
done
