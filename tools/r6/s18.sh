#!/bin/bash
# round 6 session 18: the trimmed + widened native-rank tests, then session 17's full-size rank rehearsals
mkdir -p gpurun_out/r6s18
timeout 900 python -m pytest tests/test_gpu_native_ranks.py -q -x -m gpu --durations=8 2>&1 | tail -25
bash tools/r6/s17.sh
