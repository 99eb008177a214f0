#!/bin/bash
# round 6 session 17: the multi-GPU forms of the OTHER BASELINE configs on the final sources, data path through the librccl stand-in
# (ranks share the GPU): C4 substitute (Hubbard 4x5 N=5) at 4 and 3 ranks, kagome-30 at 2 ranks, a C5 sibling (triangular 6x6
# k=(1,0), 12 down spins) at 8 ranks and at 1 rank -- E0 and step counts against the one-rank lines
mkdir -p gpurun_out/r6s17
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
R=$PWD; O=$R/gpurun_out/r6s17
run() {   # workload ranks
  export TMPDIR=/tmp/stub_$1_$2; mkdir -p $TMPDIR
  QBH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1500 python bench.py --workload $1 --gpus $2 --steps 4 --warmup 2 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1 > $O/$1_$2_ranks.log 2>&1
  echo "rc $?" >> $O/$1_$2_ranks.log
  grep '"metric"' $O/$1_$2_ranks.log | tail -1 > $O/$1_$2_ranks.json
  rm -rf $TMPDIR
  python - $O/$1_$2_ranks.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    c=d["config"]
    print(c["workload"], d["n_gpus"], "e0 %.12f" % d["e0"], "steps", d.get("lanczos_steps_to_converge"), c.get("exchange"), "ms/step", d["ms_per_step"], (c.get("kron_split") or {}).get("columns"), c.get("kernel"))
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
}
run hubbard_4x5_n5 4
run hubbard_4x5_n5 3
run kagome_30 2
run triangular_6x6_k10_n12 1
run triangular_6x6_k10_n12 8
tail -5 $O/*.log | tail -60
