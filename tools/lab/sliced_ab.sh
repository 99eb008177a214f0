#!/bin/bash
# kernel-level A/B of the far-part layouts of the Kronecker split on the headline operator (per-kernel times + HBM fetch)
set -u
R=/root/repo
export TMPDIR=/tmp
OUT=$R/gpurun_out/sl
mkdir -p $OUT
cd /tmp
for m in 0 1; do
  export QBH_NO_AUTOTUNE=1 QBH_KRON_SLICED=$m SPMV_REPS=6
  rm -rf /tmp/sl$m; mkdir -p /tmp/sl$m
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/sl$m/stats -o s -- python3 $R/tools/spmv_time.py hubbard_4x4_half "" > /tmp/sl$m/log 2>&1
  python3 $R/tools/stats_summary.py /tmp/sl$m/stats "spmv_time sliced=$m" | head -14 > $OUT/stats_sliced$m.txt
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/sl$m/g1 -o p -- python3 $R/tools/spmv_time.py hubbard_4x4_half "" > /tmp/sl$m/g1.log 2>&1
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/sl$m/g4 -o p -- python3 $R/tools/spmv_time.py hubbard_4x4_half "" > /tmp/sl$m/g4.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/sl$m "%k_spmv_wave2%" > $OUT/pmc_sliced$m.txt 2>&1
  cat $OUT/stats_sliced$m.txt $OUT/pmc_sliced$m.txt
done
