#!/bin/bash
# round 5 session 32: is the spread between fresh processes on one box an ORDER effect (temperature / clocks)?  six processes in a row, in order,
# with the device's temperature, power and clocks read in between; then the same after a pause
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s32; mkdir -p $O
cd $R
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1"
smi() { rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -i "junction\|memory\|Average Graphics\|sclk\|mclk\|fclk" | tr -s ' ' | cut -c1-90 | paste -sd';' | cut -c1-600; }
{
echo "idle: $(smi)"
for i in 1 2 3 4 5 6; do
  python bench.py $ARGS 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('process $i frac', d['roofline']['frac'], 'ms', d['roofline']['ms_per_launch'], 'step', d['ms_per_step'], 'build_s', d['config']['build_s'])"
  echo "   after: $(smi)"
done
echo "pause 60 s"; sleep 60
echo "idle: $(smi)"
for i in 7 8; do
  python bench.py $ARGS 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('process $i frac', d['roofline']['frac'], 'ms', d['roofline']['ms_per_launch'], 'step', d['ms_per_step'])"
  echo "   after: $(smi)"
done
} 2>&1 | tee $O/order_effect.txt
