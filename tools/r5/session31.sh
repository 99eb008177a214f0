#!/bin/bash
# round 5 session 31: bench.py --gpus 8 at C3 as the driver launches it, eight ranks sharing this one GPU, exchange on the NATIVE communicator
# through the test-only librccl stand-in (host-staged: the timings say nothing about links; E0, split shards, parts and the schema are what is checked)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s31; mkdir -p $O
cd $R
make -C tests/stub_rccl > /dev/null 2>&1
n=8
export TMPDIR=/tmp/stub$n; mkdir -p $TMPDIR
QBH_RCCL_LIB=$R/tests/stub_rccl/librccl_stub.so QBH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 4 --warmup 1 --no-cpu-baseline --no-fast-path --no-matrix-free > $O/c3_${n}_ranks_native_stub.log 2>&1
grep '"metric"' $O/c3_${n}_ranks_native_stub.log | tail -1 > $O/c3_${n}_ranks_native_stub.json
tail -5 $O/c3_${n}_ranks_native_stub.log | cut -c1-300
python - $O/c3_${n}_ranks_native_stub.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print(d["n_gpus"], d["config"]["exchange"], "e0", d["e0"], "steps", d.get("lanczos_steps_to_converge"), "ms/step", d["ms_per_step"], d.get("exchange"))
    for p in d.get("per_rank", []): print(" ", p)
except Exception as e:
    print("ERR", e)
PY
rm -rf $TMPDIR
