#!/bin/bash
# regression sweep over the workloads that did NOT change this round (compare with profiles/r3_bench)
R=/root/repo
O=$R/gpurun_out/r4_bench
mkdir -p $O
cd $R
BA="--steps 30 --warmup 5 --no-cpu-baseline --no-matrix-free"
python bench.py $BA --workload chain_26 2>/dev/null | grep '"metric"' > $O/chain_26.json
python bench.py $BA --workload chain_26 --host-csr reference-order 2>/dev/null | grep '"metric"' > $O/host_chain_26.json
python bench.py $BA --workload triangular_6x6_k10_n12 2>/dev/null | grep '"metric"' > $O/triangular_6x6_k10_n12.json
python bench.py $BA --workload triangular_6x6_k10_n15 --no-fast-path 2>/dev/null | grep '"metric"' > $O/triangular_6x6_k10_n15.json
python bench.py $BA --workload kagome_30 2>/dev/null | grep '"metric"' > $O/kagome_30_default_with_fast_path.json
python - <<'PY'
import json,glob
for f in ('chain_26','host_chain_26','triangular_6x6_k10_n12','triangular_6x6_k10_n15','kagome_30_default_with_fast_path'):
    try:
        j=json.load(open('/root/repo/gpurun_out/r4_bench/%s.json'%f)); fp=j.get('fast_path',{})
        print(f, 'it/s', j['value'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'], 'kernel', j['config']['kernel'], '| fast', fp.get('value'), (fp.get('roofline') or {}).get('ms_per_launch'), 'e0', j['e0'])
    except Exception as e: print(f, 'failed', e)
PY
