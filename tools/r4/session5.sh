#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s5
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_kron.py tests/test_gpu_basis.py tests/test_gpu_dist.py tests/test_gpu_parity.py tests/test_gpu_reforder.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
python tools/r4/gap_probe.py > $O/gap_probe.txt 2>&1
cat $O/gap_probe.txt | tail -5
