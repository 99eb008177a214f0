#!/bin/bash
mkdir -p gpurun_out/r4suite
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r4suite/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4suite/pytest_gpu.log
tail -25 gpurun_out/r4suite/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
