#!/bin/bash
# gather in parts: the dist tests (gloo 2/3 ranks with the part hooks, native 1 rank), then what splitting the far pass costs at C3
mkdir -p gpurun_out/r4parts
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_ragged.py -x -q -m gpu > gpurun_out/r4parts/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4parts/pytest.log
tail -5 gpurun_out/r4parts/pytest.log
timeout 900 python tools/r4/parts_probe.py > gpurun_out/r4parts/parts_probe.txt 2> gpurun_out/r4parts/parts_probe.err
echo "probe rc $?"
cat gpurun_out/r4parts/parts_probe.txt
tail -3 gpurun_out/r4parts/parts_probe.err
