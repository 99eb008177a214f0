#!/bin/bash
# round 4, GPU session 3: in-place split + shards: parity; bench C3 (fold on/off), C4 substitute
R=/root/repo
O=$R/gpurun_out/r4s3
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kron.py tests/test_gpu_dist.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
BA="--steps 40 --warmup 5 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free"
for r in 1 2; do
  python bench.py $BA 2>$O/bench_fold_$r.err | grep '"metric"' > $O/bench_fold_$r.json
  QBH_NO_TILE_FOLD=1 python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_nofold_$r.json
done
python bench.py $BA --workload hubbard_4x5_n5 2>$O/bench_4x5n5.err | grep '"metric"' > $O/bench_4x5n5.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4s3/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], 'ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'], j['config'].get('kron_split'))
    except Exception as e: print(f, 'failed', e)
PY
tail -3 $O/bench_4x5n5.err
