#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s8
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_basis.py tests/test_gpu_dist.py tests/test_gpu_reforder.py tests/test_gpu_hostcsr.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
export TMPDIR=/tmp
cd /tmp
BA="--steps 60 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free"
for r in 1 2; do
for mode in fold nofold folddummy1; do
  rm -rf /tmp/prof_$mode; mkdir -p /tmp/prof_$mode
  unset QBH_NO_TILE_FOLD QBH_FOLD_DUMMY_TILE
  if [ $mode = nofold ]; then export QBH_NO_TILE_FOLD=1; fi
  if [ $mode = folddummy1 ]; then export QBH_FOLD_DUMMY_TILE=1; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode/stats -o s -- python3 $R/bench.py $BA > /tmp/prof_$mode/stats.log 2>&1
  echo "== $mode run $r" >> $O/kernel_stats.txt
  python3 $R/tools/stats_summary.py /tmp/prof_$mode/stats "bench $mode" | grep -E "wave2|axpy|kron_tile|zero_cut" >> $O/kernel_stats.txt
  grep '"metric"' /tmp/prof_$mode/stats.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'])" >> $O/kernel_stats.txt
done
done
cat $O/kernel_stats.txt
