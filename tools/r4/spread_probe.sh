#!/bin/bash
# is the spread of the SpMV time a property of the process or of each (re)allocation?  the operator rebuilt 6 times inside one process, 2 processes
R=/root/repo
cd $R
export QBH_NO_AUTOTUNE=1 SPMV_REPS=10
for p in 1 2; do
  echo "== process $p"
  python3 tools/spmv_time.py hubbard_4x4_half "" "" "" "" "" "" 2>/dev/null | grep "ms/launch"
done
BA="--steps 30 --warmup 5 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free"
python bench.py $BA --deterministic 2>/dev/null | grep '"metric"' > gpurun_out/r4_bench/hubbard_4x4_half_deterministic.json
python bench.py $BA 2>/dev/null | grep '"metric"' > gpurun_out/r4_bench/hubbard_4x4_half_same_box_default.json
python - <<'PY'
import json
for f in ('hubbard_4x4_half_deterministic','hubbard_4x4_half_same_box_default'):
    j=json.load(open('/root/repo/gpurun_out/r4_bench/%s.json'%f)); print(f, j['value'], j['ms_per_step'], j['roofline']['ms_per_launch'], j['roofline']['frac'])
PY
