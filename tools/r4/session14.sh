#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s14
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kron.py tests/test_gpu_ragged.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
export QBH_NO_AUTOTUNE=1 SPMV_REPS=10
for r in 1 2 3 4 5; do
  python3 tools/spmv_time.py hubbard_4x4_half "" "QBH_NO_FAR_ALIGN=1" 2>/dev/null | grep "ms/launch"
done
