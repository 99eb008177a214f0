#!/bin/bash
# round 5 session 28: remainder kernel with 8 lanes per row + the reductions fused into the orbit-order launches: parity, timing, kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s28; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_hubrepr.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.log
export ORBIT=1,0 STEPS=8
timeout 300 python tools/sector_time.py hubbard_4x5_n8_k20_mf hubbard_4x5_half_k00_mf 2>&1 | grep ms_per_apply | cut -c1-330 | tee $O/timing.txt
export TMPDIR=/tmp ORBIT=1
cd /tmp; rm -rf /tmp/st; mkdir -p /tmp/st
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st/stats -o s -- python3 $R/tools/sector_time.py hubbard_4x5_half_k00_mf > /tmp/st/log 2>&1
python3 $R/tools/stats_summary.py /tmp/st/stats "python tools/sector_time.py hubbard_4x5_half_k00_mf (ORBIT=1 STEPS=8)" | head -16 | cut -c1-150 | tee $O/kernel_stats.txt
