#!/bin/bash
# round 5 session 30 (final sources: 8-lane remainder kernel): full GPU tier + smoke, then the evidence on one box: C3 / table route /
# matrix-free kernel stats + PMC, CG kernel stats, C4 as written (matrix-free sector operator) kernel stats + PMC + a line to convergence,
# the driver-style default line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s30; mkdir -p $O
cd $R
python tools/src_hash.py | tee $O/src_hash.txt
make -C tests/stub_rccl > /dev/null 2>&1
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x --durations=25 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -32 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_bench.sh r5_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace|c16" > $O/profile_c3.log 2>&1; tail -4 $O/profile_c3.log | cut -c1-160
cd $R; bash tools/profile_bench.sh r5_c3_fast "hubbard_4x4_half|rows|dict|real|kron_sliced|table" --format fast > $O/profile_fast.log 2>&1; tail -4 $O/profile_fast.log | cut -c1-160
cd $R; bash tools/profile_bench.sh r5_c3_mf "hubbard_4x4_half|matrix_free|plain|real" --matrix-free > $O/profile_mf.log 2>&1; tail -4 $O/profile_mf.log | cut -c1-160
cd $R; bash tools/profile_bench.sh r5_c4_half "hubbard_4x5_half_k00_mf|matrix_free|plain|real" --workload hubbard_4x5_half_k00_mf > $O/profile_c4.log 2>&1; tail -4 $O/profile_c4.log | cut -c1-160
cd /tmp
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --processes 1"
rm -rf /tmp/prof_cg; mkdir -p /tmp/prof_cg
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cg/stats -o s -- python3 $R/bench.py $ARGS > /tmp/prof_cg/stats.log 2>&1
python3 $R/tools/stats_summary.py /tmp/prof_cg/stats "python bench.py $ARGS  (with the locate_E0 section: Lanczos to convergence + CG eigenvector)" > $R/gpurun_out/r5_c3_cg_kernel_stats.txt
grep '"metric"' /tmp/prof_cg/stats.log | tail -1 >> $R/gpurun_out/r5_c3_cg_kernel_stats.txt
cd $R
timeout 900 python bench.py --workload hubbard_4x5_half_k00_mf --converge --steps 20 --warmup 3 --no-cpu-baseline --processes 1 > $O/c4_half_converged.json 2> $O/c4_half_converged.err
grep '"metric"' $O/c4_half_converged.json | cut -c1-700
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print({k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic"), r.get("traffic_stale"), d.get("processes",{}).get("frac"))
PY
