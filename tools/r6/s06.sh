#!/bin/bash
# round 6 session 6: solo-rank model, pipelined vs unpipelined loop under the communicator
mkdir -p gpurun_out/r6s06
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
OUT=gpurun_out/r6s06/solo.jsonl
: > $OUT
for P in 2 8; do
  for rate in 100000 50; do
    for pl in 1 0; do
      QBH_STUB_SOLO=$rate timeout 600 python tools/solo_rank.py hubbard_4x4_half $P 0 steps=20 warmup=4 parts=4 realwire=1 pipeline=$pl 2>gpurun_out/r6s06/err.txt | grep '^{' >> $OUT
    done
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r6s06/solo.jsonl"):
    d = json.loads(ln)
    ks = [k for k in d if k.startswith("ms_spmv")][0]; kg = [k for k in d if k.startswith("ms_gather")][0]
    print(d["ranks"], d["rank"], d["link_model"]["GBps_per_link"], "parts", d["gather_parts"], "elem", d["element_bytes"], "pipeline", d["lanczos_pipeline"], "step", d["ms_per_step"], "kernels", d[ks], "gather", d[kg])
PY
