#!/bin/bash
R=/root/repo
cd $R
tools/profile_bench.sh r4_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace" > gpurun_out/r4final_profile_c3.log 2>&1
cd $R
tools/profile_bench.sh r4_k30_cut18 "kagome_30|wave|plain|kron_sliced|inplace|cut18" --workload kagome_30 --site-cut 18 > gpurun_out/r4final_profile_k30.log 2>&1
cd $R
head -9 gpurun_out/r4_c3_kernel_stats.txt | cut -c1-150
python - <<'PY'
import json
for t in ('r4_c3','r4_k30_cut18'):
    e=json.load(open('/root/repo/gpurun_out/%s_traffic_entry.json'%t)); k=list(e)[0]; print(k, e[k]['hbm_bytes'], e[k]['kernel_sources_sha16'])
    j=json.load(open('/root/repo/gpurun_out/%s_bench_line.json'%t)); print('  plain line', j['value'], j['ms_per_step'], j['roofline']['ms_per_launch'], j['roofline']['frac'])
PY
cd $R
tools/profile_bench.sh r4_c3_fast "hubbard_4x4_half|rows|dict|real|kron_sliced" --format fast > gpurun_out/r4final_profile_c3_fast.log 2>&1
cd $R
head -9 gpurun_out/r4_c3_fast_kernel_stats.txt | cut -c1-150
python tools/src_hash.py
