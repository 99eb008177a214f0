#!/bin/bash
R=/root/repo
cd $R
ROUNDS=5 SPMV_REPS=10 tools/lab/ab_libs.sh hubbard_4x4_half tools/lab/variants/r4_base.so tools/lab/variants/r4_chunk6.so tools/lab/variants/r4_chunk8.so
