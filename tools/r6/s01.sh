#!/bin/bash
# round 6 session 1: the pipelined Lanczos loop -- new tests, the driver tests, then C3 with and without the pipeline (one box, alternating)
mkdir -p gpurun_out/r6s01
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -15 > gpurun_out/r6s01/pytest_pipeline.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lanczos or locate or cg" 2>&1 | tail -8 > gpurun_out/r6s01/pytest_parity.txt
C="--steps 20 --warmup 5 --processes 1 --no-cpu-baseline --no-fast-path --no-matrix-free --no-converge --no-locate"
for i in 1 2; do
  timeout 600 python bench.py $C > gpurun_out/r6s01/c3_pipe_$i.json 2> gpurun_out/r6s01/c3_pipe_$i.err
  timeout 600 python bench.py $C --no-pipeline > gpurun_out/r6s01/c3_nopipe_$i.json 2> gpurun_out/r6s01/c3_nopipe_$i.err
done
cat gpurun_out/r6s01/pytest_pipeline.txt gpurun_out/r6s01/pytest_parity.txt
for f in gpurun_out/r6s01/c3_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["value"], d["ms_per_step"], d["roofline"].get("ms_per_launch"), d["roofline"]["frac"])
except Exception as e:
    print("ERR", e)
PY
done
