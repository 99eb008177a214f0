#!/bin/bash
# second counter set (memory pipeline occupancy / stalls); usage as pmc_lab.sh
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
TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
TCC_BUSY_avr TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_REQ_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum
TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_IB_REQ_sum TCC_STREAMING_REQ_sum
GROUPS
python3 $R/tools/pmc_summary.py $OUT "%k_wave%" > $OUT/summary.txt
find $OUT -name "*.db" -delete
