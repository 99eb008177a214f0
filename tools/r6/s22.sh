#!/bin/bash
# round 6 session 22: qbh_opts.comm_reserve at its default (64 workgroups, far pass <= 2 per CU) against none, exchange = a kernel of RCCL's
# footprint (28 / 56 workgroups) or the host-function hold; then the native-rank tests on the final library
mkdir -p gpurun_out/r6s22
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
OUT=gpurun_out/r6s22/solo_comm_reserve_default.jsonl
: > $OUT
one() {   # P rank rate kernel reserve
  ( [ -n "$4" ] && export QBH_STUB_SOLO_KERNEL=$4
    QBH_STUB_SOLO=$3 timeout 600 python tools/solo_rank.py hubbard_4x4_half $1 $2 steps=20 warmup=4 parts=4 realwire=1 sparse=1 partition=1 reserve=$5 2>gpurun_out/r6s22/err.txt | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); d['occupancy_model']='$4' or 'host function'; print(json.dumps(d))" >> $OUT ) || { echo "FAILED $*"; tail -5 gpurun_out/r6s22/err.txt; }
}
for P in 8 4 2; do
  for rank in 0 $((P/2)); do
    for rate in 50 25; do
      for rep in 1 2; do
        one $P $rank $rate "" -1
        one $P $rank $rate "" 0
        one $P $rank $rate 28:rccl -1
        one $P $rank $rate 28:rccl 0
        one $P $rank $rate 56:rccl 0
      done
    done
  done
done
python - <<'PY'
import json, collections
acc = collections.defaultdict(list)
for ln in open("gpurun_out/r6s22/solo_comm_reserve_default.jsonl"):
    d = json.loads(ln)
    acc[(d["ranks"], d["rank"], d["link_model"]["GBps_per_link"], d["occupancy_model"], d["comm_reserve"])].append(d["ms_per_step"])
for k in sorted(acc, key=lambda k: (-k[0], k[1], -k[2], k[3], k[4])):
    print("P %d rank %d rate %g model %-13s comm_reserve %2d | ms per step %s" % (k + (" ".join("%.3f" % v for v in acc[k]),)))
PY
timeout 900 python -m pytest tests/test_gpu_native_ranks.py tests/test_gpu_dist.py tests/test_abi.py -q -x -m gpu 2>&1 | tail -5
