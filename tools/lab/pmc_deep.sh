#!/bin/bash
# Where does the vector-memory path of the split passes stall?  TA / TCP / UTCL1 / TCC / TD busy and stall counters (one group per
# run, counters only) for the SpMV kernels of the headline operator, and for the pure-stream probe as the reference.
set -u
R=/root/repo
export TMPDIR=/tmp QBH_NO_AUTOTUNE=1 SPMV_REPS=4
OUT=$R/gpurun_out/lab
mkdir -p $OUT
cd /tmp
GROUPS_=(
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
 "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum GRBM_GUI_ACTIVE"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
 "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum"
 "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum"
)
rm -rf /tmp/pd; mkdir -p /tmp/pd/k /tmp/pd/p
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $g -d /tmp/pd/k/g$i -o p -- python3 $R/tools/spmv_time.py hubbard_4x4_half "" > /tmp/pd/k/g$i.log 2>&1
  timeout 120 rocprofv3 --pmc $g -d /tmp/pd/p/g$i -o p -- $R/tools/lab/region_probe.bin > /tmp/pd/p/g$i.log 2>&1
done
{ echo "# split passes of C3 (far = <2, 3, true>, near = <2, 2, true>): mean per dispatch"; python3 $R/tools/pmc_summary.py /tmp/pd/k "%k_spmv_wave2%";
  echo "# reference: region_probe kernels (k_reg<NS, REG, COL, ST>)"; python3 $R/tools/pmc_summary.py /tmp/pd/p "%k_reg<8, 1, 0, 0>%"; python3 $R/tools/pmc_summary.py /tmp/pd/p "%k_reg<8, 1, 1, 1>%"; } > $OUT/pmc_deep.txt 2>&1
cat $OUT/pmc_deep.txt
