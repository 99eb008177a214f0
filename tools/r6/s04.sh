#!/bin/bash
# round 6 session 4: the timing path of the sharded headline, one rank at a time with modelled peers (librccl stand-in, solo mode)
mkdir -p gpurun_out/r6s04
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
OUT=gpurun_out/r6s04/solo.jsonl
: > $OUT
for P in 2 4 8; do
  for rank in 0 $((P/2)); do
    for rate in 100000 50 150; do
      for cfg in "parts=1 realwire=1" "parts=4 realwire=1" "parts=8 realwire=1" "parts=4 realwire=0"; do
        QBH_STUB_SOLO=$rate timeout 600 python tools/solo_rank.py hubbard_4x4_half $P $rank steps=20 warmup=4 $cfg 2>gpurun_out/r6s04/err_${P}_${rank}.txt | grep '^{' >> $OUT
      done
    done
  done
done
# the unpipelined loop under the same model (what the device-resident scalars buy with a communicator: nothing yet, the comm path reads back every step)
python - <<'PY'
import json
for ln in open("gpurun_out/r6s04/solo.jsonl"):
    d = json.loads(ln)
    ks = [k for k in d if k.startswith("ms_spmv")][0]; kg = [k for k in d if k.startswith("ms_gather")][0]
    print(d["ranks"], d["rank"], d["link_model"]["GBps_per_link"], "parts", d["gather_parts"], "elem", d["element_bytes"], d["columns"], "step", d["ms_per_step"], "kernels", d[ks], "gather", d[kg])
PY
