#!/bin/bash
# round 5 session 41 (experiment, bench.py only): the solver's vectors allocated BEFORE the operator (fresh memory) against after it, alternating processes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s41; mkdir -p $O
cd $R
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1"
{
for i in 1 2 3 4 5 6; do
  for v in "" 1; do
    BENCH_VEC_FIRST=$v timeout 200 python bench.py $ARGS 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vec_first [$v] round $i frac', d['roofline']['frac'], 'ms', d['roofline']['ms_per_launch'], 'step', d['ms_per_step'])" || echo "vec_first [$v] round $i FAILED"
  done
done
} 2>&1 | tee $O/vec_first_ab.txt
