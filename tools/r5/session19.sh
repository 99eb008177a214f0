#!/bin/bash
# round 5 session 19: k_mf_sector_orb variants on 4x5 8+8: non-temporal streams, dynamic item walk, grid sizes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s19; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_hubrepr.py -x -q -m gpu -k "matrix_free or leaked" 2>&1 | tail -3 | tee $O/pytest_mf.log
export ORBIT=1 STEPS=8
for dbg in "sec_nt=1" "sec_nt=0" "sec_nt=1,sec_walk=1" "sec_nt=0,sec_walk=1" "sec_nt=1,sec_grid=1024" "sec_nt=1,sec_grid=1280" "sec_nt=1,sec_grid=1536" "sec_nt=1,sec_grid=2048" "sec_nt=1,sec_unroll=4" "sec_nt=1,sec_walk=1,sec_grid=1024" "sec_nt=1,sec_walk=1,sec_grid=2048"; do
  echo "== QBH_DEBUG=$dbg" | tee -a $O/variants.txt
  QBH_DEBUG=$dbg timeout 300 python tools/sector_time.py hubbard_4x5_n8_k20_mf 2>&1 | grep ms_per_apply | cut -c1-330 | tee -a $O/variants.txt
done
