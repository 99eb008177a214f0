#!/bin/bash
# round 5 session 39 (experiment): frees inside a creation call deferred to its end (no holes between the arrays that stay) against immediate frees, alternating processes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s39; mkdir -p $O
cd $R
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1"
{
for i in 1 2 3 4 5 6; do
  for v in 0 1; do
    QBH_DEBUG=defer_free=$v timeout 200 python bench.py $ARGS 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('defer_free $v round $i frac', d['roofline']['frac'], 'ms', d['roofline']['ms_per_launch'], 'step', d['ms_per_step'], 'build_s', d['config']['build_s'])" || echo "defer_free $v round $i FAILED"
  done
done
} 2>&1 | tee $O/defer_ab.txt
