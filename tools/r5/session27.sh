#!/bin/bash
# round 5 session 27: BASELINE configs[3] as written, ALL inequivalent momentum sectors on one GPU with the orbit-order kernel (real and complex
# sectors), to compare energy by energy with the round-2 run of the rank-table kernel (profiles/r2_sectors/c4_hubbard_4x5_half_all_sectors_ONE_gpu_matrix_free.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s27; mkdir -p $O
cd $R
timeout 1500 python tools/hubbard_sectors.py 4 5 10 10 --mf 2>&1 | grep -v amdgpu.ids | tee $O/c4_all_sectors_orbit_order.txt
