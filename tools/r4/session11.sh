#!/bin/bash
# kagome-30 (C2) with the cut-sector split: cuts 15 and 18 against the unsplit operator, then kernel stats + PMC of the better one
R=/root/repo
O=$R/gpurun_out/r4s11
mkdir -p $O $R/gpurun_out/r4_bench
cd $R
BA="--steps 30 --warmup 5 --no-cpu-baseline --no-fast-path --no-matrix-free --workload kagome_30"
python bench.py $BA 2>$O/k30_plain.err | grep '"metric"' > $R/gpurun_out/r4_bench/kagome_30_unsplit.json
for h in 15 18 12; do
  python bench.py $BA --site-cut $h 2>$O/k30_cut$h.err | grep '"metric"' > $R/gpurun_out/r4_bench/kagome_30_cut$h.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4_bench/kagome_30*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], 'it/s', j['value'], 'ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'], 'e0', j['e0'], j['config'].get('basis_internal'), 'build_s', j['config']['build_s'])
    except Exception as e: print(f, 'failed', e)
PY
tail -3 $O/*.err
