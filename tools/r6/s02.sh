#!/bin/bash
# round 6 session 2: pipeline tests; C3 with the host disturbed by a polling loop (what the driver's box does: rocm-smi beside the run), pipelined and not
mkdir -p gpurun_out/r6s02
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q 2>&1 | tail -5 > gpurun_out/r6s02/pytest_pipeline.txt
C="--steps 20 --warmup 5 --processes 1 --no-cpu-baseline --no-fast-path --no-matrix-free --no-converge --no-locate"
( while true; do rocm-smi --showuse --showmemuse > /dev/null 2>&1; sleep 0.2; done ) &
POLL=$!
for i in 1 2; do
  timeout 600 python bench.py $C > gpurun_out/r6s02/c3_poll_pipe_$i.json 2> gpurun_out/r6s02/c3_poll_pipe_$i.err
  timeout 600 python bench.py $C --no-pipeline > gpurun_out/r6s02/c3_poll_nopipe_$i.json 2> gpurun_out/r6s02/c3_poll_nopipe_$i.err
done
kill $POLL
# 8 busy host threads (the box has many cores; this only matters if the runtime's threads get descheduled)
for k in 1 2 3 4 5 6 7 8; do ( while true; do :; done ) & BURN="$BURN $!"; done
timeout 600 python bench.py $C > gpurun_out/r6s02/c3_burn_pipe.json 2> gpurun_out/r6s02/c3_burn_pipe.err
timeout 600 python bench.py $C --no-pipeline > gpurun_out/r6s02/c3_burn_nopipe.json 2> gpurun_out/r6s02/c3_burn_nopipe.err
kill $BURN
cat gpurun_out/r6s02/pytest_pipeline.txt
for f in gpurun_out/r6s02/c3_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["value"], d["ms_per_step"], d["roofline"].get("ms_per_launch"), d["roofline"]["frac"])
except Exception as e:
    print("ERR", e)
PY
done
