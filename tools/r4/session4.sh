#!/bin/bash
# round 4, GPU session 4: parity of the in-place split (+ shards, dist); kernel traces fold on / off
R=/root/repo
O=$R/gpurun_out/r4s4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kron.py tests/test_gpu_dist.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
export TMPDIR=/tmp
cd /tmp
BA="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free"
for mode in fold nofold; do
  rm -rf /tmp/prof_$mode; mkdir -p /tmp/prof_$mode
  if [ $mode = nofold ]; then export QBH_NO_TILE_FOLD=1; else unset QBH_NO_TILE_FOLD; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode/stats -o s -- python3 $R/bench.py $BA > /tmp/prof_$mode/stats.log 2>&1
  python3 $R/tools/stats_summary.py /tmp/prof_$mode/stats "bench $mode" | head -14 > $O/${mode}_kernel_stats.txt
  grep '"metric"' /tmp/prof_$mode/stats.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'])" >> $O/${mode}_kernel_stats.txt
  cat $O/${mode}_kernel_stats.txt
done
