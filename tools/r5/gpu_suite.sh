#!/bin/bash
# round 5: the whole GPU tier, log kept under gpurun_out/r5_suite/
mkdir -p gpurun_out/r5_suite
timeout ${1:-1500} python -m pytest tests -q -m gpu -x --durations=15 ${@:2} > gpurun_out/r5_suite/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r5_suite/pytest.log
tail -40 gpurun_out/r5_suite/pytest.log
