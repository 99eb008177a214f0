#!/bin/bash
# round 5 session 33 (experiment): large allocations through the virtual-memory API (QBH_DEBUG=vmm=...) -- does the per-process placement lottery change?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s33; mkdir -p $O
cd $R
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1"
{
for v in 0 1 2 4 3; do
  for i in 1 2 3 4; do
    QBH_DEBUG=vmm=$v timeout 200 python bench.py $ARGS 2>$O/err_$v.txt | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vmm $v process $i frac', d['roofline']['frac'], 'ms', d['roofline']['ms_per_launch'], 'step', d['ms_per_step'], 'build_s', d['config']['build_s'], 'e0', d['e0'])" || { echo "vmm $v process $i FAILED"; tail -3 $O/err_$v.txt; }
  done
done
} 2>&1 | tee $O/vmm.txt
