#!/bin/bash
# traffic counters only (two passes); usage as pmc_lab.sh
set -u
R=/root/repo
TAG=$1; shift
export TMPDIR=/tmp
OUT=$R/gpurun_out/lab/pmc_$TAG
mkdir -p $OUT
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d $OUT/g$i -o p -- $R/tools/lab/spmv_lab.bin "$@" > $OUT/g$i.log 2>&1
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum
GROUPS
python3 $R/tools/pmc_summary.py $OUT "%k_wave%" > $OUT/summary.txt
find $OUT -name "*.db" -delete
