#!/bin/bash
R=/root/repo
cd $R
tools/profile_bench.sh r4_k30_cut18 "kagome_30|wave|plain|kron_sliced|inplace|cut18" --workload kagome_30 --site-cut 18 > gpurun_out/r4s12_profile.log 2>&1
tail -40 gpurun_out/r4s12_profile.log
