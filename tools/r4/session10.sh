#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s10
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_kron.py tests/test_gpu_dist.py tests/test_gpu_basis.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -40 $O/pytest.log
