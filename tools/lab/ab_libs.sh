#!/bin/bash
# A/B timing of library variants on one box: alternate processes, several rounds, min / median per variant
# usage: ROUNDS=3 tools/lab/ab_libs.sh <workload> <variant.so> [<variant.so> ...]     (environment switches pass through)
R=/root/repo
W=$1; shift
ROUNDS=${ROUNDS:-3}
export QBH_NO_AUTOTUNE=${QBH_NO_AUTOTUNE:-1} SPMV_REPS=${SPMV_REPS:-6}
T=$(mktemp -d)
for r in $(seq $ROUNDS); do
  for so in "$@"; do
    QBHIP_LIBRARY=$so python3 $R/tools/spmv_time.py $W "" 2>/dev/null | grep "ms/launch" | awk '{print $(NF-3)}' >> $T/$(basename $so).txt
  done
done
for so in "$@"; do
  sort -n $T/$(basename $so).txt | awk -v n="$(basename $so)" '{a[NR]=$1} END {printf "%-32s min %8.3f  median %8.3f  max %8.3f  (%d runs)\n", n, a[1], a[int((NR+1)/2)], a[NR], NR}'
done
rm -rf $T
