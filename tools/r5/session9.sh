#!/bin/bash
# round 5 session 9: kernel stats of the user call (Lanczos + CG) at C3; kernel stats + PMC traffic of the default format's table route;
# C5-family line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s9; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --processes 1"
rm -rf /tmp/prof_cg; mkdir -p /tmp/prof_cg
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cg/stats -o s -- python3 $R/bench.py $ARGS > /tmp/prof_cg/stats.log 2>&1
python3 $R/tools/stats_summary.py /tmp/prof_cg/stats "python bench.py $ARGS  (with the locate_E0 section: Lanczos to convergence + CG eigenvector)" > $R/gpurun_out/r5_c3_cg_kernel_stats.txt
grep '"metric"' /tmp/prof_cg/stats.log | tail -1 >> $R/gpurun_out/r5_c3_cg_kernel_stats.txt
head -16 $R/gpurun_out/r5_c3_cg_kernel_stats.txt | cut -c1-150
cd $R
bash tools/profile_bench.sh r5_c3_fast "hubbard_4x4_half|rows|dict|real|kron_sliced|table" --format fast > $O/profile_fast.log 2>&1; tail -25 $O/profile_fast.log | cut -c1-200
cd $R
timeout 900 python bench.py --workload triangular_6x6_k10_n15 --processes 1 --no-cpu-baseline --no-locate --no-fast-path --no-matrix-free > $O/c5_n15.json 2> $O/c5_n15.err
python - $O/c5_n15.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print({k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("kernel"), r.get("frac"), r.get("ms_per_launch"))
PY
