#!/bin/bash
# round 5 session 18: counters of the two sector kernels on 4x5 with 8+8 (7.9e8 rows), one group per pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s18; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_hubrepr.py -x -q -m gpu -k "matrix_free or leaked" 2>&1 | tail -5 | tee $O/pytest_mf.log
export TMPDIR=/tmp STEPS=6
cd /tmp
for orb in 1 0; do
  export ORBIT=$orb
  rm -rf /tmp/pm$orb; mkdir -p /tmp/pm$orb
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
             "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp -d /tmp/pm$orb/g$i -o p -- python3 $R/tools/sector_time.py hubbard_4x5_n8_k20_mf > /tmp/pm$orb/g$i.log 2>&1
    grep ms_per_apply /tmp/pm$orb/g$i.log | cut -c1-200
  done
  python3 $R/tools/pmc_summary.py /tmp/pm$orb "%k_mf_sector%" | tee $O/pmc_orbit$orb.txt
done
