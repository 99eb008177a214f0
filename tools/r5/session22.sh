#!/bin/bash
# round 5 session 22: k_mf_sector_orb after the LDS layout and uniform-base changes: parity, timing, LDS / VALU counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s22; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_hubrepr.py -x -q -m gpu -k "matrix_free or leaked" 2>&1 | tail -3 | tee $O/pytest_mf.log
export ORBIT=1 STEPS=8
for wl in hubbard_4x5_n8_k20_mf hubbard_4x5_half_k00_mf; do
  timeout 300 python tools/sector_time.py $wl 2>&1 | grep ms_per_apply | cut -c1-330 | tee -a $O/timing.txt
done
QBH_DEBUG=sec_unroll=4 timeout 300 python tools/sector_time.py hubbard_4x5_half_k00_mf 2>&1 | grep ms_per_apply | cut -c1-330 | tee -a $O/timing.txt
export TMPDIR=/tmp STEPS=6
cd /tmp
wl=hubbard_4x5_half_k00_mf
rm -rf /tmp/pm_$wl; mkdir -p /tmp/pm_$wl
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pm_$wl/g$i -o p -- python3 $R/tools/sector_time.py $wl > /tmp/pm_$wl/g$i.log 2>&1
done
python3 $R/tools/pmc_summary.py /tmp/pm_$wl "%k_mf_sector%" | tee $O/pmc_$wl.txt
