#!/bin/bash
# round 4, GPU session 1: parity of the folded tile copy; A/B fold on/off through bench.py (one box); near-LDS ablations
R=/root/repo
O=$R/gpurun_out/r4s1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kron.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
BA="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free"
for r in 1 2; do
  python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_fold_$r.json
  QBH_NO_TILE_FOLD=1 python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_nofold_$r.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4s1/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], 'ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'])
    except Exception as e: print(f, 'failed', e)
PY
ROUNDS=2 tools/lab/ab_libs.sh hubbard_4x4_half tools/lab/variants/r4_fold.so tools/lab/variants/r4_abl1.so tools/lab/variants/r4_abl2.so > $O/ab_abl.txt 2>&1
cat $O/ab_abl.txt
