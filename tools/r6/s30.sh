#!/bin/bash
# round 6 session 30: FINAL sources (ABI 601, comm_reserve on, calibration in bench.py) -- the data path at full size through the stand-in:
# C3 at 4 and 8 rank processes, the C4 substitute at 4
mkdir -p gpurun_out/r6s30
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
R=$PWD; O=$R/gpurun_out/r6s30
run() {   # workload ranks
  export TMPDIR=/tmp/stub_$1_$2; mkdir -p $TMPDIR
  QBH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1700 python bench.py --workload $1 --gpus $2 --steps 4 --warmup 2 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate > $O/$1_$2_ranks.log 2>&1
  echo "rc $?" >> $O/$1_$2_ranks.log
  grep '"metric"' $O/$1_$2_ranks.log | tail -1 > $O/$1_$2_ranks_native_stub_abi601.json
  rm -rf $TMPDIR
  python - $O/$1_$2_ranks_native_stub_abi601.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    x=d["exchange"]
    print(d["config"]["workload"], d["n_gpus"], "e0 %.12f" % d["e0"], "steps", d.get("lanczos_steps_to_converge"), "reserve", x.get("comm_reserve_workgroups"), x.get("comm_reserve_calibration"), "needed", x.get("needed_frac_rank0"), "parts", x.get("gather_parts"))
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
}
run hubbard_4x4_half 4
run hubbard_4x5_n5 4
run hubbard_4x4_half 8
tail -3 $O/*.log | cut -c1-200 | tail -20
