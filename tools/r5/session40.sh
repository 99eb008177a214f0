#!/bin/bash
# round 5 session 40: device_alloc in its product form (one physical allocation per array >= 256 MB, frees unmap + release, addresses stay reserved)
# against hipMalloc, alternating processes; thresholds
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s40; mkdir -p $O
cd $R
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1"
{
for i in 1 2 3 4 5; do
  for dbg in "vmm=0" "vmm=1" "vmm=1,vmm_min_mb=1024" "vmm=1,vmm_min_mb=64"; do
    QBH_DEBUG=$dbg timeout 200 python bench.py $ARGS 2>$O/err.txt | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dbg round $i frac', d['roofline']['frac'], 'ms', d['roofline']['ms_per_launch'], 'step', d['ms_per_step'], 'build_s', d['config']['build_s'])" 2>/dev/null || { echo "$dbg round $i FAILED"; grep -i "fault" $O/err.txt | head -1; }
  done
done
} 2>&1 | tee $O/vmm_product_ab.txt
