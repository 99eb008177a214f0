#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s16
mkdir -p $O $R/gpurun_out/r4_bench
cd $R
timeout 1200 python -m pytest tests/test_gpu_ragged.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
BA="--steps 30 --warmup 5 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --workload kagome_30"
for r in 1 2; do
python bench.py $BA 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/kagome_30_unsplit_$r.json
python bench.py $BA --site-cut 18 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/kagome_30_cut18_two_passes_$r.json
python bench.py $BA --site-cut 15 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/kagome_30_cut15_two_passes_$r.json
QBH_CROSS_IN_NEAR=0 python bench.py $BA --site-cut 18 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/kagome_30_cut18_three_passes_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4_bench/kagome_30_*_[12].json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], 'it/s', j['value'], 'ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'])
    except Exception as e: print(f, 'failed', e)
PY
