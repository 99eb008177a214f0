#!/bin/bash
# Collect rocprofv3 PMC counters for the SpMV probe, one counter group per pass (counters
# only: never combined with trace options).  usage: tools/pmc_passes.sh <tag> [probe args...]
# Output: gpurun_out/pmc_<tag>/<group>/..._results.db ; summarise with tools/pmc_summary.py
set -u
R=/root/repo
TAG=$1; shift
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $OUT/g$i -o p -- python3 $R/tools/spmv_probe.py --reps 3 "$@" > $OUT/g$i.log 2>&1
  echo "$grp" > $OUT/g$i.counters
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD
FETCH_SIZE
WRITE_SIZE
GROUPS
grep -h '"ms"' $OUT/g1.log | tail -1
