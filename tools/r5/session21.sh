#!/bin/bash
# round 5 session 21: counters of the orbit-order kernel as it is now (dynamic walk, 4-byte slot entries) on 4x5 8+8 and at half filling
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s21; mkdir -p $O
export TMPDIR=/tmp STEPS=6 ORBIT=1
cd /tmp
for wl in hubbard_4x5_n8_k20_mf hubbard_4x5_half_k00_mf; do
  rm -rf /tmp/pm_$wl; mkdir -p /tmp/pm_$wl
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp -d /tmp/pm_$wl/g$i -o p -- python3 $R/tools/sector_time.py $wl > /tmp/pm_$wl/g$i.log 2>&1
    grep ms_per_apply /tmp/pm_$wl/g$i.log | cut -c1-200
  done
  python3 $R/tools/pmc_summary.py /tmp/pm_$wl "%k_mf_sector%" | tee $O/pmc_$wl.txt
done
