#!/bin/bash
# round 6 session 25: with the exchange as a kernel of RCCL's footprint -- number of parts K, more channels than the reserve holds (112 workgroups
# against 64 / 128 left out), and the reserve's size
mkdir -p gpurun_out/r6s25
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
OUT=gpurun_out/r6s25/solo_parts_and_channels.jsonl
: > $OUT
one() {   # P rank rate kernel reserve parts
  ( [ -n "$4" ] && export QBH_STUB_SOLO_KERNEL=$4
    QBH_STUB_SOLO=$3 timeout 600 python tools/solo_rank.py hubbard_4x4_half $1 $2 steps=20 warmup=4 parts=$6 realwire=1 sparse=1 partition=1 reserve=$5 2>gpurun_out/r6s25/err.txt | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); d['occupancy_model']='$4' or 'host function'; print(json.dumps(d))" >> $OUT ) || { echo "FAILED $*"; tail -5 gpurun_out/r6s25/err.txt; }
}
for P in 8 4 2; do
  for K in 1 2 4 8; do
    one $P 0 50 28:rccl 0 $K
    one $P 0 25 28:rccl 0 $K
  done
  for R in 16 32 64 128; do
    one $P 0 50 28:rccl $R 4
    one $P 0 50 "" $R 4
  done
  one $P 0 50 112:rccl 0 4
  one $P 0 50 112:rccl 128 4
  one $P 0 25 112:rccl 0 4
  one $P 0 25 112:rccl 128 4
done
python - <<'PY'
import json
for ln in open("gpurun_out/r6s25/solo_parts_and_channels.jsonl"):
    d = json.loads(ln)
    kg = [k for k in d if k.startswith("ms_gather")][0]
    print("P", d["ranks"], "rate", d["link_model"]["GBps_per_link"], "model", d["occupancy_model"], "reserve", d["comm_reserve"], "parts", d["gather_parts"], "| step", d["ms_per_step"], "gather", d[kg])
PY
