#!/bin/bash
# round 5 session 20: k_mf_sector_orb with the dynamic walk by default and 4-byte slot entries: tile sizes, unroll; then C4 as written
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s20; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_hubrepr.py -x -q -m gpu -k "matrix_free or leaked" 2>&1 | tail -3 | tee $O/pytest_mf.log
export ORBIT=1 STEPS=8
for dbg in "sec_tile=1024" "sec_tile=512" "sec_tile=256" "sec_tile=2048" "sec_tile=512,sec_unroll=4" "sec_tile=1024,sec_unroll=4" "sec_tile=512,sec_nt=0"; do
  echo "== QBH_DEBUG=$dbg" | tee -a $O/variants.txt
  QBH_DEBUG=$dbg timeout 300 python tools/sector_time.py hubbard_4x5_n8_k20_mf 2>&1 | grep ms_per_apply | cut -c1-330 | tee -a $O/variants.txt
done
for dbg in "sec_tile=1024" "sec_tile=512" "sec_tile=512,sec_unroll=4"; do
  echo "== QBH_DEBUG=$dbg" | tee -a $O/variants.txt
  QBH_DEBUG=$dbg timeout 300 python tools/sector_time.py hubbard_4x5_half_k00_mf 2>&1 | grep ms_per_apply | cut -c1-330 | tee -a $O/variants.txt
done
