#!/bin/bash
# round 5 session 2: the new basis-search tests, then bench lines: default (median of 3 fresh processes, locate_E0, fast path),
# reference order with and without the hint, deterministic
set -x
O=gpurun_out/r5_s2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_basis.py tests/test_gpu_kron.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --order reference --no-basis-hint --processes 1 --no-cpu-baseline --no-locate > $O/bench_reforder_nohint.json 2> $O/bench_reforder_nohint.err
timeout 900 python bench.py --order reference --processes 1 --no-cpu-baseline --no-locate > $O/bench_reforder_hint.json 2> $O/bench_reforder_hint.err
timeout 900 python bench.py --deterministic --processes 3 --no-cpu-baseline --no-locate --no-fast-path --no-matrix-free > $O/bench_deterministic.json 2> $O/bench_deterministic.err
for f in $O/bench_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print({k:d.get(k) for k in ("value","ms_per_step","e0")}, d["roofline"].get("frac"), d["roofline"].get("ms_per_launch"), d.get("processes",{}).get("frac"), d.get("create"))
    print("locate", d.get("locate_E0")); print("fast", (d.get("fast_path") or {}).get("roofline"))
except Exception as e: print("ERR", e)
PY
done
tail -3 $O/*.err
