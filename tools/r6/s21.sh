#!/bin/bash
# round 6 session 21: the exchange as a kernel with the footprint of RCCL's own (QBH_STUB_SOLO_KERNEL=W:rccl -- 256 threads, 280 registers,
# 19.7 KB LDS) beside the persistent passes: workgroups left out of the grids (comm_reserve) and a cap on the far pass's workgroups per CU
mkdir -p gpurun_out/r6s21
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
OUT=gpurun_out/r6s21/solo_occupancy_rccl.jsonl
: > $OUT
one() {   # P rank rate kernel debug
  ( [ -n "$4" ] && export QBH_STUB_SOLO_KERNEL=$4; [ -n "$5" ] && export QBH_DEBUG=$5
    QBH_STUB_SOLO=$3 timeout 600 python tools/solo_rank.py hubbard_4x4_half $1 $2 steps=20 warmup=4 parts=4 realwire=1 sparse=1 partition=1 2>gpurun_out/r6s21/err.txt | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); d['occupancy_model']='$4' or 'host function'; d['QBH_DEBUG']='$5'; print(json.dumps(d))" >> $OUT ) || { echo "FAILED $*"; tail -5 gpurun_out/r6s21/err.txt; }
}
for P in 8 4 2; do
  for rate in 50 25; do
    one $P 0 $rate "" ""
    one $P 0 $rate "" comm_far_cap=2
    one $P 0 $rate "" comm_reserve=64
    one $P 0 $rate "" comm_reserve=64,comm_far_cap=2
    one $P 0 $rate 28:rccl ""
    one $P 0 $rate 28:rccl comm_reserve=32
    one $P 0 $rate 28:rccl comm_reserve=64
    one $P 0 $rate 28:rccl comm_reserve=32,comm_far_cap=2
    one $P 0 $rate 28:rccl comm_reserve=64,comm_far_cap=2
    one $P 0 $rate 56:rccl comm_reserve=64
    one $P 0 $rate 56:rccl comm_reserve=64,comm_far_cap=2
    one $P 0 $rate 8:rccl comm_reserve=64,comm_far_cap=2
    one $P 0 $rate 8:rccl comm_reserve=8
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r6s21/solo_occupancy_rccl.jsonl"):
    d = json.loads(ln)
    ks = [k for k in d if k.startswith("ms_spmv")][0]; kg = [k for k in d if k.startswith("ms_gather")][0]
    print("P", d["ranks"], "rate", d["link_model"]["GBps_per_link"], "model", d["occupancy_model"], "debug", d["QBH_DEBUG"] or "-", "| step", d["ms_per_step"], "kernels", d[ks], "gather", d[kg])
PY
