#!/bin/bash
# Is the process-to-process spread of the near pass (14.0 vs 15.0-15.4 ms on one box) a translation (TLB) effect?
# N processes, each: kernel durations + UTCL1 / pending-stall counters of the two passes.
R=/root/repo
export TMPDIR=/tmp QBH_NO_AUTOTUNE=1 SPMV_REPS=6
O=$R/gpurun_out/r4_bimodal
mkdir -p $O
cd /tmp
: > $O/summary.txt
for i in 1 2 3 4 5 6 7 8; do
  rm -rf /tmp/bm$i; mkdir -p /tmp/bm$i
  timeout 300 rocprofv3 --kernel-trace --stats --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum -d /tmp/bm$i/g1 -o p -- python3 $R/tools/spmv_time.py hubbard_4x4_half "" > /tmp/bm$i/log 2>&1
  echo "== process $i" >> $O/summary.txt
  python3 $R/tools/stats_summary.py /tmp/bm$i/g1 "x" | grep -E "wave2" | cut -c1-130 >> $O/summary.txt
  python3 $R/tools/pmc_summary.py /tmp/bm$i "%k_spmv_wave2%" >> $O/summary.txt
done
cat $O/summary.txt
