#!/bin/bash
# round 6 session 15: number of gather parts under the final configuration (partition + personalised exchange)
mkdir -p gpurun_out/r6s15
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so PYTHONPATH=$PWD
OUT=gpurun_out/r6s15/solo.jsonl; : > $OUT
for P in 2 8; do for rate in 50 25; do for K in 1 2 4; do
  QBH_STUB_SOLO=$rate timeout 600 python tools/solo_rank.py hubbard_4x4_half $P 0 steps=20 warmup=4 parts=$K 2>/dev/null | grep '^{' >> $OUT
done; done; done
python - <<'PY'
import json
for ln in open("gpurun_out/r6s15/solo.jsonl"):
    d = json.loads(ln); ks = [k for k in d if k.startswith("ms_spmv")][0]; kg = [k for k in d if k.startswith("ms_gather")][0]
    print(d["ranks"], d["link_model"]["GBps_per_link"], "parts", d["gather_parts"], "step", d["ms_per_step"], "kernels", d[ks], "gather", d[kg])
PY
