#!/bin/bash
# round 6 session 29: bench.py --gpus N with the comm_reserve calibration through the stand-in (tests) and C3 at 2 stub ranks (the line's new fields)
mkdir -p gpurun_out/r6s29
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_native_ranks.py tests/test_abi.py -q -x -m gpu -k "bench or abi or version or export" --durations=5 2>&1 | tail -8
R=$PWD; O=$R/gpurun_out/r6s29
export TMPDIR=/tmp/stub2; mkdir -p $TMPDIR
QBH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1500 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate > $O/c3_2_ranks.log 2>&1
grep '"metric"' $O/c3_2_ranks.log | tail -1 > $O/c3_2_ranks_native_stub_abi601.json
python - $O/c3_2_ranks_native_stub_abi601.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print(d["n_gpus"], "e0 %.12f" % d["e0"], "steps", d.get("lanczos_steps_to_converge"), d["exchange"].get("comm_reserve_workgroups"), d["exchange"].get("comm_reserve_calibration"))
PY
tail -3 $O/c3_2_ranks.log | cut -c1-300
