#!/bin/bash
R=/root/repo
cd $R
export QBH_NO_AUTOTUNE=1 SPMV_REPS=10 QBH_PRINT_PTRS=1
for p in 1 2; do
echo "== process $p"
python3 tools/spmv_time.py hubbard_4x4_half "" "" "" "" "" "" "" "" 2>&1 | grep -E "ms/launch|qbhip kron" | sed -e 's/qbhip kron arrays: //' -e 's/ | nnz_n.*//' | cut -c1-230
done
