#!/bin/bash
# round 5: 2-byte columns A/B on C3 (one box): parity tests first, then bench lines with and without
set -x
mkdir -p gpurun_out/r5_c16
timeout 900 python -m pytest tests/test_gpu_kron.py -x -q -m gpu -k "two_byte or split_operator" > gpurun_out/r5_c16/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_c16/pytest.log
tail -5 gpurun_out/r5_c16/pytest.log
for c in 1 0 1 0; do
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-matrix-free --no-fast-path --cols16 $c > gpurun_out/r5_c16/bench_c16_$c.$RANDOM.json 2> gpurun_out/r5_c16/bench_err_$c.log
done
grep -o '"ms_per_launch": [0-9.]*\|"frac": [0-9.]*\|"columns": "[^"]*"' gpurun_out/r5_c16/bench_c16_*.json
