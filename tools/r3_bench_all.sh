#!/bin/bash
# the round-3 bench lines kept under profiles/r3_bench/ (one GPU)
mkdir -p gpurun_out/r3
for w in hubbard_4x4_half kagome_30 hubbard_4x5_n5 triangular_6x6_k10_n15 chain_26; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-matrix-free 2>>gpurun_out/r3/bench.err | tail -1 > gpurun_out/r3/$w.json
done
