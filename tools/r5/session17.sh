#!/bin/bash
# round 5 session 17: the matrix-free sector operator with its rows orbit by orbit (k_mf_sector_orb): parity on the small clusters and at
# 7.5e7 rows, then time per apply beside the rank-table kernel on 6+6, 8+8 and C4 as written
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s17; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_hubrepr.py -x -q -m gpu -k "matrix_free or leaked" 2>&1 | tail -15 | tee $O/pytest_mf.log
timeout 900 python tools/sector_time.py 2>&1 | grep -v amdgpu.ids | tee $O/sector_time.txt
