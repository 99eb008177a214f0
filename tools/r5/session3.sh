#!/bin/bash
# round 5 session 3: (a) A/B of the 32-bit far-slot arithmetic on C3, (b) kagome-30: forms side by side + where the cut form's passes stall
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_k30; mkdir -p $O
export TMPDIR=/tmp SPMV_REPS=6
cd $R
ROUNDS=4 bash tools/lab/ab_libs.sh hubbard_4x4_half $R/tools/lab/variants/r5_farat64.so $R/tools/lab/variants/r5_farat32.so > $O/ab_farat.txt 2>&1
cat $O/ab_farat.txt
SPMV_ROUNDS=2 python3 tools/spmv_time.py kagome_30 "" "site_cut=18" "site_cut=18 kron_cross_in_near=0" "site_cut=15" "site_cut=20" > $O/forms.txt 2>&1
cat $O/forms.txt
cd /tmp
GROUPS_=(
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
)
for cfg in cut18 unsplit; do
  rm -rf /tmp/pd_$cfg; mkdir -p /tmp/pd_$cfg
  i=0
  for g in "${GROUPS_[@]}"; do
    i=$((i+1))
    if [ $cfg = cut18 ]; then c="site_cut=18"; else c=""; fi
    SPMV_REPS=3 timeout 300 rocprofv3 --pmc $g -d /tmp/pd_$cfg/g$i -o p -- python3 $R/tools/spmv_time.py kagome_30 "$c" > /tmp/pd_$cfg/g$i.log 2>&1
  done
  { echo "# kagome-30 $cfg: mean per dispatch"; python3 $R/tools/pmc_summary.py /tmp/pd_$cfg "%k_spmv_wave%"; } > $O/pmc_deep_$cfg.txt 2>&1
  cat $O/pmc_deep_$cfg.txt
done
SPMV_REPS=4 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_cut18 -o k -- python3 $R/tools/spmv_time.py kagome_30 "site_cut=18" > /tmp/ks.log 2>&1
python3 $R/tools/stats_summary.py /tmp/ks_cut18 "python3 tools/spmv_time.py kagome_30 site_cut=18 (SPMV_REPS=4)" > $O/kernel_stats_cut18.txt 2>&1 || find /tmp/ks_cut18 -name "*kernel_stats*" | head
head -20 $O/kernel_stats_cut18.txt
