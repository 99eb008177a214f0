#!/bin/bash
# round 6 session 11: partition order of the major indices + personalised exchange -- bench tests, solo model, C3 data path at 2 / 4 / 8 stub ranks
mkdir -p gpurun_out/r6s11
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_native_ranks.py -q -x -k "bench" 2>&1 | tail -4
OUT=gpurun_out/r6s11/solo.jsonl
: > $OUT
for P in 2 4 8; do
  for rank in 0 $((P/2)); do
    for rate in 100000 50 25; do
      for pt in 1 0; do
        QBH_STUB_SOLO=$rate timeout 600 python tools/solo_rank.py hubbard_4x4_half $P $rank steps=20 warmup=4 parts=4 realwire=1 sparse=1 partition=$pt 2>gpurun_out/r6s11/err.txt | grep '^{' >> $OUT
      done
    done
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r6s11/solo.jsonl"):
    d = json.loads(ln)
    ks = [k for k in d if k.startswith("ms_spmv")][0]; kg = [k for k in d if k.startswith("ms_gather")][0]
    print(d["ranks"], d["rank"], d["link_model"]["GBps_per_link"], "partition", d["major_partition"], "need", d["gather_needed_frac"], "step", d["ms_per_step"], "kernels", d[ks], "gather", d[kg])
PY
unset QBH_STUB_SOLO
R=$PWD; O=$R/gpurun_out/r6s11
for n in 2 4 8; do
  export TMPDIR=/tmp/stub$n; mkdir -p $TMPDIR
  QBH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1700 python bench.py --gpus $n --steps 4 --warmup 2 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate > $O/c3_${n}_ranks_native_stub.log 2>&1
  grep '"metric"' $O/c3_${n}_ranks_native_stub.log | tail -1 > $O/c3_${n}_ranks_native_stub.json
  rm -rf $TMPDIR
  python - $O/c3_${n}_ranks_native_stub.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print(d["n_gpus"], "e0 %.12f" % d["e0"], "steps", d.get("lanczos_steps_to_converge"), d.get("exchange"))
except Exception as e:
    print("ERR", e)
PY
done
