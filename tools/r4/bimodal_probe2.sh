#!/bin/bash
# does the process-to-process spread of the passes come with different TRAFFIC (L2 misses) or with the same traffic served slower?
R=/root/repo
export TMPDIR=/tmp QBH_NO_AUTOTUNE=1 SPMV_REPS=6
O=$R/gpurun_out/r4_bimodal
mkdir -p $O
cd /tmp
: > $O/summary2.txt
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/bn$i; mkdir -p /tmp/bn$i
  timeout 300 rocprofv3 --kernel-trace --stats --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d /tmp/bn$i/g1 -o p -- python3 $R/tools/spmv_time.py hubbard_4x4_half "" > /tmp/bn$i/log 2>&1
  echo "== process $i" >> $O/summary2.txt
  python3 $R/tools/stats_summary.py /tmp/bn$i/g1 "x" | grep -E "wave2" | cut -c1-130 >> $O/summary2.txt
  python3 $R/tools/pmc_summary.py /tmp/bn$i "%k_spmv_wave2%" >> $O/summary2.txt
done
cat $O/summary2.txt
