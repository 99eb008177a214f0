#!/bin/bash
# round 5 session 23: where do the 118 GB above the compulsory traffic of k_mf_sector_orb come from?  L2 counters under variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s23; mkdir -p $O
export TMPDIR=/tmp STEPS=6 ORBIT=1
cd /tmp
wl=hubbard_4x5_half_k00_mf
for dbg in "sec_nt=0" "sec_tile=512" "sec_tile=256" "sec_grid=1024" "sec_grid=768"; do
  rm -rf /tmp/pmv; mkdir -p /tmp/pmv
  QBH_DEBUG=$dbg timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/pmv/g1 -o p -- python3 $R/tools/sector_time.py $wl > /tmp/pmv/g1.log 2>&1
  echo "== QBH_DEBUG=$dbg" | tee -a $O/variants_l2.txt
  grep ms_per_apply /tmp/pmv/g1.log | cut -c150-260 | tee -a $O/variants_l2.txt
  python3 $R/tools/pmc_summary.py /tmp/pmv "%k_mf_sector%" | tee -a $O/variants_l2.txt
done
