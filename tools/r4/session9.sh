#!/bin/bash
# round 4, GPU session 9: whole GPU suite; headline profile (stats + PMC + plain line on one box); the other bench lines
R=/root/repo
O=$R/gpurun_out/r4s9
mkdir -p $O $R/gpurun_out/r4_bench
cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -45 $O/pytest.log
tools/profile_bench.sh r4_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace" > $O/profile_c3.log 2>&1
tail -30 $O/profile_c3.log
cd $R
BA="--steps 30 --warmup 5 --no-cpu-baseline --no-fast-path --no-matrix-free"
python bench.py $BA --workload hubbard_4x5_n5 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/hubbard_4x5_n5.json
python bench.py $BA --no-converge --order reference 2>$O/reforder.err | grep '"metric"' > $R/gpurun_out/r4_bench/hubbard_4x4_half_reforder_hint.json
python bench.py $BA --no-converge --order reference --no-basis-hint 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/hubbard_4x4_half_reforder_nohint.json
python bench.py $BA --host-csr reference-order --workload hubbard_4x3_half 2>$O/hostcsr.err | grep '"metric"' > $R/gpurun_out/r4_bench/host_hubbard_4x3_half_hint.json
python bench.py $BA --host-csr reference-order --workload hubbard_4x3_half --no-basis-hint 2>/dev/null | grep '"metric"' > $R/gpurun_out/r4_bench/host_hubbard_4x3_half_nohint.json
QBH_KRON_REUSE_TILE=1 python tools/shard_time.py hubbard_4x4_half 2 4 8 > $R/gpurun_out/r4_bench/c3_shards_one_gpu.jsonl 2>$O/shards.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4_bench/*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], 'it/s', j['value'], 'ms_per_step', j['ms_per_step'], 'spmv', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'], 'kron', bool(j['config'].get('kron_split')), j.get('create'))
    except Exception as e: print(f, 'failed', e)
PY
cat $R/gpurun_out/r4_bench/c3_shards_one_gpu.jsonl
tail -3 $O/reforder.err $O/hostcsr.err $O/shards.err
