#!/bin/bash
# round 6 session 20: does the exchange hide under the PERSISTENT near pass when it needs CUs?  Solo rank (peers modelled), the
# hold of the side stream as a KERNEL of W workgroups with an LDS footprint (QBH_STUB_SOLO_KERNEL=W:ldsKB) instead of a host
# function; RCCL's stream at the highest priority (the library's default since this session) vs the default priority
# (QBH_DEBUG=side_noprio=1) vs workgroups left out of the persistent grids (QBH_DEBUG=comm_reserve=W)
mkdir -p gpurun_out/r6s20
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
OUT=gpurun_out/r6s20/solo_occupancy.jsonl
: > $OUT
one() {   # P rank rate kernel debug
  local tag="P=$1 rank=$2 rate=$3 kernel=${4:-host} debug=${5:-none}"
  ( [ -n "$4" ] && export QBH_STUB_SOLO_KERNEL=$4; [ -n "$5" ] && export QBH_DEBUG=$5
    QBH_STUB_SOLO=$3 timeout 600 python tools/solo_rank.py hubbard_4x4_half $1 $2 steps=20 warmup=4 parts=4 realwire=1 sparse=1 partition=1 2>gpurun_out/r6s20/err.txt | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); d['occupancy_model']='$4' or 'host function'; d['QBH_DEBUG']='$5'; print(json.dumps(d))" >> $OUT ) || { echo "FAILED $tag"; tail -5 gpurun_out/r6s20/err.txt; }
}
for P in 8 2; do
  for rate in 50 25; do
    one $P 0 $rate "" ""
    one $P 0 $rate 16:0 ""
    one $P 0 $rate 16:0 side_noprio=1
    one $P 0 $rate 16:100 ""
    one $P 0 $rate 16:100 side_noprio=1
    one $P 0 $rate 56:100 ""
    one $P 0 $rate 56:100 side_noprio=1
    one $P 0 $rate 16:100 side_noprio=1,comm_reserve=16
    one $P 0 $rate 56:100 comm_reserve=56
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r6s20/solo_occupancy.jsonl"):
    d = json.loads(ln)
    ks = [k for k in d if k.startswith("ms_spmv")][0]; kg = [k for k in d if k.startswith("ms_gather")][0]
    print("P", d["ranks"], "rate", d["link_model"]["GBps_per_link"], "model", d["occupancy_model"], "debug", d["QBH_DEBUG"] or "-", "| step", d["ms_per_step"], "kernels", d[ks], "gather", d[kg])
PY
