#!/bin/bash
# round 5 session 6: reproducible reductions under the dynamic walk (chunk partials): A/B against per-workgroup partials, parity, lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s6; mkdir -p $O
export TMPDIR=/tmp SPMV_REPS=6
cd $R
ROUNDS=4 bash tools/lab/ab_libs.sh hubbard_4x4_half $R/tools/lab/variants/r5_wg_partials.so $R/tools/lab/variants/r5_chunk_partials.so > $O/ab_chunk_partials.txt 2>&1
cat $O/ab_chunk_partials.txt
timeout 1500 python -m pytest tests/test_gpu_kron.py tests/test_gpu_ragged.py tests/test_gpu_dist.py tests/test_gpu_native_ranks.py tests/test_gpu_reforder.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 900 python bench.py --deterministic --processes 3 --no-cpu-baseline --no-locate --no-fast-path --no-matrix-free > $O/bench_deterministic.json 2> $O/bench_deterministic.err
timeout 900 python bench.py --processes 3 --no-cpu-baseline --no-locate --no-fast-path --no-matrix-free > $O/bench_default.json 2> $O/bench_default.err
for f in $O/*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]
    print({k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("frac"), r.get("ms_per_launch"), d.get("processes",{}).get("frac"))
except Exception as e: print("ERR", e)
PY
done
