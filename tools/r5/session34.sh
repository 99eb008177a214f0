#!/bin/bash
# round 5 session 34 (experiment): hipMalloc against ONE physical allocation per large array through the virtual-memory API, alternating processes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s34; mkdir -p $O
cd $R
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1"
{
for i in 1 2 3 4 5 6 7 8; do
  for v in 0 4; do
    QBH_DEBUG=vmm=$v timeout 200 python bench.py $ARGS 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vmm $v round $i frac', d['roofline']['frac'], 'ms', d['roofline']['ms_per_launch'], 'step', d['ms_per_step'])" || echo "vmm $v round $i FAILED"
  done
done
} 2>&1 | tee $O/vmm_ab.txt
