#!/bin/bash
# round 4, final evidence of the final code: GPU suite, headline profile (stats + PMC + plain line on ONE box), the other bench lines
R=/root/repo
O=$R/gpurun_out/r4final
mkdir -p $O $R/gpurun_out/r4_bench
cd $R
( time timeout 1500 python -m pytest tests -q -m gpu --durations=12 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -22 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -3 $O/smoke.log
tools/profile_bench.sh r4_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace" > $O/profile_c3.log 2>&1
cd $R
tools/profile_bench.sh r4_k30_cut18 "kagome_30|wave|plain|kron_sliced|inplace|cut18" --workload kagome_30 --site-cut 18 > $O/profile_k30.log 2>&1
cd $R
( time python bench.py ) > $O/bench_default_full.log 2>&1
grep '"metric"' $O/bench_default_full.log | tail -1 > $R/gpurun_out/r4_bench/bench_default_full.json
BA="--steps 30 --warmup 5 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free"
for r in 1 2 3; do python bench.py $BA 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/hubbard_4x4_half_final_$r.json; done
python bench.py $BA --deterministic 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/hubbard_4x4_half_deterministic.json
python bench.py $BA --workload hubbard_4x5_n5 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/hubbard_4x5_n5_final.json
python bench.py $BA --order reference 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/hubbard_4x4_half_reforder_hint_final.json
QBH_KRON_REUSE_TILE=1 python tools/shard_time.py hubbard_4x4_half 2 4 8 > $R/gpurun_out/r4_bench/c3_shards_one_gpu_final.jsonl 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4_bench/*final*.json')+glob.glob('/root/repo/gpurun_out/r4_bench/*deterministic*.json')+['/root/repo/gpurun_out/r4_bench/bench_default_full.json']):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], 'it/s', j['value'], 'ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'], 'traffic', j['roofline'].get('traffic'), j['roofline'].get('traffic_stale'))
    except Exception as e: print(f, 'failed', e)
PY
cat $R/gpurun_out/r4_bench/c3_shards_one_gpu_final.jsonl
grep real $O/bench_default_full.log
head -8 $R/gpurun_out/r4_c3_kernel_stats.txt | cut -c1-140
python -c "
import json
e=json.load(open('$R/gpurun_out/r4_c3_traffic_entry.json')); k=list(e)[0]; print(k, e[k]['hbm_bytes'], e[k]['kernel_sources_sha16'])
e=json.load(open('$R/gpurun_out/r4_k30_cut18_traffic_entry.json')); k=list(e)[0]; print(k, e[k]['hbm_bytes'])
"
