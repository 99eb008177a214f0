#!/bin/bash
# rocprofv3 PMC passes (counters only, one group per run) over the kernel lab: tools/lab/pmc_lab.sh <tag> <lab args...>
# environment switches of the lab (LAB_ONLY, LAB_B, LAB_U) must be exported by the caller.
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
  timeout 400 rocprofv3 --pmc $grp -d $OUT/g$i -o p -- $R/tools/lab/spmv_lab.bin "$@" > $OUT/g$i.log 2>&1
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
GROUPS
python3 $R/tools/pmc_summary.py $OUT "%k_%" > $OUT/summary.txt
find $OUT -name "*.db" -delete
cat $OUT/summary.txt
