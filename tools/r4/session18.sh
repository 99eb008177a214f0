#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s18
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kron.py tests/test_gpu_ragged.py tests/test_gpu_dist.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
QBH_NO_AUTOTUNE=1 SPMV_REPS=4 QBHIP_LIBRARY=$R/tools/lab/variants/r4_xcdtime.so python3 tools/spmv_time.py hubbard_4x4_half "" 2>&1 | grep -E "xcd timing|ms/launch" | tail -5
ROUNDS=5 SPMV_REPS=10 tools/lab/ab_libs.sh hubbard_4x4_half tools/lab/variants/r4_steal.so tools/lab/variants/r4_nosteal.so
