#!/bin/bash
# round 6 session 5: bench.py --gpus N at C3 (N = 2, 4, 8 ranks sharing this one GPU, native communicator through the librccl stand-in's DATA
# path: host-staged, the timings say nothing about links) -- E0, steps, the schema; columns and wire format of the sharded headline.
# bench.py starts its own ranks (no torch.distributed.run on the command line).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s05; mkdir -p $O
cd $R
make -C tests/stub_rccl > /dev/null 2>&1
timeout 900 python bench.py --gpus 1 --steps 4 --warmup 2 --processes 1 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate > $O/c3_1_rank.log 2>&1
grep '"metric"' $O/c3_1_rank.log | tail -1 > $O/c3_1_rank.json
for n in 2 4 8; do
  export TMPDIR=/tmp/stub$n; mkdir -p $TMPDIR
  QBH_RCCL_LIB=$R/tests/stub_rccl/librccl_stub.so QBH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1700 python bench.py --gpus $n --steps 4 --warmup 2 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate > $O/c3_${n}_ranks_native_stub.log 2>&1
  grep '"metric"' $O/c3_${n}_ranks_native_stub.log | tail -1 > $O/c3_${n}_ranks_native_stub.json
  tail -3 $O/c3_${n}_ranks_native_stub.log | cut -c1-200
  rm -rf $TMPDIR
done
python - $O <<'PY'
import json,sys
o=sys.argv[1]
for n in (1,2,4,8):
    f = "%s/c3_%d_rank%s.json" % (o, n, "" if n == 1 else "s_native_stub")
    try:
        d=json.loads(open(f).read())
        print(n, "e0 %.12f" % d["e0"], "steps", d.get("lanczos_steps_to_converge"), "ms/step", d["ms_per_step"], d["config"].get("kron_split",{}).get("columns"), d.get("exchange",{}).get("element_bytes"), d.get("exchange",{}).get("gather_parts"))
        for p in d.get("per_rank", []): print("   ", p)
    except Exception as e:
        print(n, "ERR", e)
PY
