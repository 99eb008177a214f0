#!/bin/bash
# round 5 session 29: fused reductions x up-hop unroll of k_mf_sector_orb (register pressure decides: 95 / 107 / 93 / 83 VGPRs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s29; mkdir -p $O
cd $R
export ORBIT=1 STEPS=8
for dbg in "sec_fuse=0,sec_up=8" "sec_fuse=1,sec_up=8" "sec_fuse=1,sec_up=4" "sec_fuse=0,sec_up=4"; do
  echo "== QBH_DEBUG=$dbg" | tee -a $O/variants.txt
  QBH_DEBUG=$dbg timeout 300 python tools/sector_time.py hubbard_4x5_half_k00_mf hubbard_4x5_n8_k20_mf 2>&1 | grep ms_per_apply | cut -c1-250 | tee -a $O/variants.txt
done
