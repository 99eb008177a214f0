#!/bin/bash
# round 5 session 16 (after the memset fix and the leading-entry fix of the 2-byte near columns): C3 with 2 / 4 ranks through the native communicator on the stand-in (split shards must stay split),
# then the FINAL evidence: full GPU tier, smoke, C3 / table route / matrix-free profiles on one box, CG kernel stats, default line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s16; mkdir -p $O
cd $R
python tools/src_hash.py | tee $O/src_hash.txt
make -C tests/stub_rccl > /dev/null 2>&1
for n in 2 4; do export TMPDIR=/tmp/stub$n; mkdir -p $TMPDIR;
  QBH_RCCL_LIB=$R/tests/stub_rccl/librccl_stub.so QBH_DIST_BACKEND=gloo TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 6 --warmup 2 --no-cpu-baseline --no-fast-path --no-matrix-free > $O/c3_${n}_ranks_native_stub.log 2>&1
  grep '"metric"' $O/c3_${n}_ranks_native_stub.log | tail -1 > $O/c3_${n}_ranks_native_stub.json
  python - $O/c3_${n}_ranks_native_stub.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print(d["n_gpus"], d["config"]["exchange"], "e0", d["e0"], "steps", d.get("lanczos_steps_to_converge"), "ms/step", d["ms_per_step"], d.get("exchange"))
    for p in d.get("per_rank", []): print(" ", p)
except Exception as e:
    print("ERR", e)
PY
done
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x --durations=6 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
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
PY
