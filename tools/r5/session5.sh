#!/bin/bash
# round 5 session 5: the table-kernel route of recognised Kronecker sums: parity tests, then the default-format lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s5; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kron.py tests/test_gpu_fullsize.py tests/test_gpu_basis.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 900 python bench.py --format fast --processes 1 --no-cpu-baseline --no-locate --no-matrix-free > $O/c3_fast_table.json 2> $O/c3_fast_table.err
timeout 900 python bench.py --format fast --workload hubbard_4x5_n5 --processes 1 --no-cpu-baseline --no-locate --no-matrix-free > $O/h45n5_fast_table.json 2> $O/h45n5_fast_table.err
for f in $O/*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]
    print({k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("kernel"), r.get("frac"), r.get("ms_per_launch"), r.get("bytes_definition"))
except Exception as e: print("ERR", e)
PY
done
tail -3 $O/*.err
