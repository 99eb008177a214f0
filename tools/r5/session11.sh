#!/bin/bash
# round 5 session 11 (final sources): whole GPU tier + smoke; evidence on ONE box for the headline (plain line + kernel stats + PMC),
# for the default format's table route and for the matrix-free operator; the driver-style default line; the CG kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s11; mkdir -p $O
export TMPDIR=/tmp
cd $R
python tools/src_hash.py | tee $O/src_hash.txt
timeout 1800 python -m pytest tests -q -m gpu -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -14 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_bench.sh r5_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace|c16" > $O/profile_c3.log 2>&1; tail -4 $O/profile_c3.log | cut -c1-160
cd $R; bash tools/profile_bench.sh r5_c3_fast "hubbard_4x4_half|rows|dict|real|kron_sliced|table" --format fast > $O/profile_fast.log 2>&1; tail -4 $O/profile_fast.log | cut -c1-160
cd $R; bash tools/profile_bench.sh r5_c3_mf "hubbard_4x4_half|matrix_free|plain|real" --matrix-free > $O/profile_mf.log 2>&1; tail -4 $O/profile_mf.log | cut -c1-160
cd /tmp
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --processes 1"
rm -rf /tmp/prof_cg; mkdir -p /tmp/prof_cg
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cg/stats -o s -- python3 $R/bench.py $ARGS > /tmp/prof_cg/stats.log 2>&1
python3 $R/tools/stats_summary.py /tmp/prof_cg/stats "python bench.py $ARGS  (with the locate_E0 section: Lanczos to convergence + CG eigenvector)" > $R/gpurun_out/r5_c3_cg_kernel_stats.txt
grep '"metric"' /tmp/prof_cg/stats.log | tail -1 >> $R/gpurun_out/r5_c3_cg_kernel_stats.txt
cd $R
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print({k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic"), r.get("traffic_stale"), d.get("processes",{}).get("frac"))
print("locate", {k:d.get("locate_E0",{}).get(k) for k in ("seconds_total","cg_ms_per_step","lanczos_ms_per_step")}, "fast", (d.get("fast_path") or {}).get("value"), ((d.get("fast_path") or {}).get("roofline") or {}).get("frac"))
PY
