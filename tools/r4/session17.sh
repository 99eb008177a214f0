#!/bin/bash
R=/root/repo
cd $R
export QBH_NO_AUTOTUNE=1 SPMV_REPS=4 QBHIP_LIBRARY=$R/tools/lab/variants/r4_xcdtime.so
python3 tools/spmv_time.py hubbard_4x4_half "" 2>&1 | grep -E "xcd timing|ms/launch" | tail -9
