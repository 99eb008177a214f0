#!/bin/bash
# the sliced coded split as the default of the real fast path: tests that touch it, then the default bench line (fast_path block)
mkdir -p gpurun_out/r4kronc
O=gpurun_out/r4kronc
timeout 1500 python -m pytest tests/test_gpu_kron.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest_default.log 2>&1
echo "pytest rc $?" >> $O/pytest_default.log
tail -6 $O/pytest_default.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-matrix-free > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["roofline"]["frac"], d["roofline"].get("ms_per_launch"))
print("fast_path", json.dumps(d.get("fast_path"))[:1500])
PY
timeout 900 python bench.py --steps 20 --warmup 3 --format fast --deterministic --no-cpu-baseline --no-matrix-free 2>/dev/null | tail -1 > $O/bench_fast_deterministic.json
python - <<PY
import json
d = json.loads(open("$O/bench_fast_deterministic.json").read().strip().splitlines()[-1])
print("fast deterministic", d["value"], d["roofline"]["frac"], d["roofline"].get("ms_per_launch"), d.get("e0"))
PY
