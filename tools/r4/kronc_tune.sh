#!/bin/bash
# QBH_KRON_CODED=2 on C3, fast format: groups in flight per pass, per-kernel times from one rocprofv3 run each
mkdir -p gpurun_out/r4kronc
O=$GRAFT_REPO_ROOT/gpurun_out/r4kronc
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_kron.py -x -q -m gpu -k "coded_real_form" 2>&1 | tail -30
cd /tmp && export TMPDIR=/tmp
for cfg in "2 3" "1 2" "3 4" "4 3" "2 1"; do
  set -- $cfg
  export QBH_KRON_CODED=2 QBH_KRONC_FAR_NG=$1 QBH_KRONC_NEAR_NG=$2
  rm -rf /tmp/kp; mkdir -p /tmp/kp
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kp -o k -- python3 $R/bench.py --format fast --steps 20 --warmup 3 --no-cpu-baseline --no-matrix-free > /tmp/kp/log 2>&1
  echo "== far_ng $1 near_ng $2" | tee -a $O/tune.txt
  python3 $R/tools/stats_summary.py /tmp/kp "tune" | grep -E "kronc_far|kronc_near|tile_re|axpy_norm_re" | tee -a $O/tune.txt
  grep '"metric"' /tmp/kp/log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s', d['value'], 'spmv', d['roofline']['ms_per_launch'], 'e0', d['e0'])" | tee -a $O/tune.txt
done
