#!/bin/bash
# the default (coded, real) format after the sliced split: the Hubbard workloads that have the structure, split on / off
mkdir -p gpurun_out/r4kronc
O=gpurun_out/r4kronc/fast_sweep.jsonl
: > $O
for wl in hubbard_4x5_n5 hubbard_4x4_half; do
  for form in 2 0; do
    QBH_KRON_CODED=$form timeout 900 python bench.py --workload $wl --format fast --steps 30 --warmup 5 --no-cpu-baseline --no-matrix-free 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'workload': '$wl', 'QBH_KRON_CODED': $form, 'it_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'ms_spmv': d['roofline']['ms_per_launch'], 'frac_own_format': d['roofline']['frac'], 'e0': d.get('e0'), 'kron_split': d['config'].get('kron_split')}))" | tee -a $O
  done
done
