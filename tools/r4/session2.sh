#!/bin/bash
# round 4, GPU session 2: fold on/off, longer and alternating; kernel trace of the folded run
R=/root/repo
O=$R/gpurun_out/r4s2
mkdir -p $O
cd $R
export QBHIP_LIBRARY=$R/tools/lab/variants/r4_fold.so
BA="--steps 60 --warmup 5 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free"
for r in 1 2 3; do
  python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_fold_$r.json
  QBH_NO_TILE_FOLD=1 python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_nofold_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4s2/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], 'ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'])
    except Exception as e: print(f, 'failed', e)
PY
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_f; mkdir -p /tmp/prof_f
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f/stats -o s -- python3 $R/bench.py --steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free > /tmp/prof_f/stats.log 2>&1
python3 $R/tools/stats_summary.py /tmp/prof_f/stats "bench fold" > $O/fold_kernel_stats.txt
grep '"metric"' /tmp/prof_f/stats.log | tail -1 >> $O/fold_kernel_stats.txt
head -16 $O/fold_kernel_stats.txt
