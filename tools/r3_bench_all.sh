mkdir -p gpurun_out/r3
for w in kagome_30 hubbard_4x5_n5 triangular_6x6_k10_n15 chain_26; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-matrix-free 2>>gpurun_out/r3/bench.err | tail -1 > gpurun_out/r3/$w.json
done
tools/profile_bench.sh r3b_c3 > /dev/null 2>&1
tools/profile_bench.sh r3b_kagome30 --workload kagome_30 --no-matrix-free --no-fast-path > /dev/null 2>&1
