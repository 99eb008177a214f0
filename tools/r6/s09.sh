#!/bin/bash
# round 6 session 9: solo-rank model after "only the needed major indices are placed"; default configuration (parts 4, real wire, pipelined)
mkdir -p gpurun_out/r6s09
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
OUT=gpurun_out/r6s09/solo.jsonl
: > $OUT
for P in 2 4 8; do
  for rank in 0 $((P/2)); do
    for rate in 100000 50 150; do
      QBH_STUB_SOLO=$rate timeout 600 python tools/solo_rank.py hubbard_4x4_half $P $rank steps=20 warmup=4 parts=4 realwire=1 2>gpurun_out/r6s09/err.txt | grep '^{' >> $OUT
    done
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r6s09/solo.jsonl"):
    d = json.loads(ln)
    ks = [k for k in d if k.startswith("ms_spmv")][0]; kg = [k for k in d if k.startswith("ms_gather")][0]; kf=[k for k in d if k.startswith("shard_spmv")][0]
    print(d["ranks"], d["rank"], d["link_model"]["GBps_per_link"], "need", d["gather_needed_frac"], "step", d["ms_per_step"], "kernels", d[ks], "gather", d[kg], "frac_on_step", d[kf])
PY
