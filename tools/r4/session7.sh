#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s7
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_basis.py tests/test_gpu_dist.py tests/test_gpu_reforder.py tests/test_gpu_hostcsr.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
BA="--steps 40 --warmup 5 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free"
for r in 1 2 3 4; do
  python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_fold_$r.json
  QBH_NO_TILE_FOLD=1 python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_nofold_$r.json
  QBH_FOLD_DUMMY_TILE=1 python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_folddummy1_$r.json
  QBH_FOLD_DUMMY_TILE=2 python bench.py $BA 2>/dev/null | grep '"metric"' > $O/bench_folddummy2_$r.json
done
python - <<'PY'
import json,glob,statistics
for mode in ('fold','nofold','folddummy1','folddummy2'):
    st,sp=[],[]
    for f in sorted(glob.glob('/root/repo/gpurun_out/r4s7/bench_%s_*.json'%mode)):
        try:
            j=json.load(open(f)); st.append(j['ms_per_step']); sp.append(j['roofline']['ms_per_launch'])
        except Exception as e: print(f,'failed',e)
    print(mode,'ms_per_step',sorted(st),'median',statistics.median(st),'| spmv',sorted(sp),'median',statistics.median(sp))
PY
