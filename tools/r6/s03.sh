#!/bin/bash
# round 6 session 3: what the pipeline hides -- a modelled host delay per step, and a host squeezed onto two busy cores
mkdir -p gpurun_out/r6s03
C="--steps 20 --warmup 5 --processes 1 --no-cpu-baseline --no-fast-path --no-matrix-free --no-converge --no-locate"
for d in 0 1500 5000; do
  QBH_DEBUG=host_delay_us=$d timeout 600 python bench.py $C > gpurun_out/r6s03/c3_delay${d}_pipe.json 2> gpurun_out/r6s03/c3_delay${d}_pipe.err
  QBH_DEBUG=host_delay_us=$d timeout 600 python bench.py $C --no-pipeline > gpurun_out/r6s03/c3_delay${d}_nopipe.json 2> gpurun_out/r6s03/c3_delay${d}_nopipe.err
done
( taskset -c 0 bash -c 'while true; do :; done' ) & B0=$!
( taskset -c 1 bash -c 'while true; do :; done' ) & B1=$!
timeout 600 taskset -c 0,1 python bench.py $C > gpurun_out/r6s03/c3_squeezed_pipe.json 2> gpurun_out/r6s03/c3_squeezed_pipe.err
timeout 600 taskset -c 0,1 python bench.py $C --no-pipeline > gpurun_out/r6s03/c3_squeezed_nopipe.json 2> gpurun_out/r6s03/c3_squeezed_nopipe.err
kill $B0 $B1
for f in gpurun_out/r6s03/c3_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["value"], d["ms_per_step"], d["roofline"].get("ms_per_launch"), d["roofline"]["frac"])
except Exception as e:
    print("ERR", e)
PY
done
