#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s19
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kron.py -x -q -m gpu -k "ragged_far or split_operator or shard" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log | cut -c1-220
