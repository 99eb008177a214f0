#!/bin/bash
# round 6 session 23: qbh_opts.comm_reserve on PLAIN row shards (split=0: wave kernel on the locally-owned columns under the gather),
# exchange = host-function hold or a kernel of RCCL's footprint; then the gloo / stub tests of plain shards and kagome-30 at 2 stub ranks
mkdir -p gpurun_out/r6s23
export QBH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so
export PYTHONPATH=$PWD
OUT=gpurun_out/r6s23/solo_plain_shards.jsonl
: > $OUT
one() {   # P rank rate kernel reserve
  ( [ -n "$4" ] && export QBH_STUB_SOLO_KERNEL=$4
    QBH_STUB_SOLO=$3 timeout 600 python tools/solo_rank.py hubbard_4x4_half $1 $2 steps=20 warmup=4 parts=1 realwire=0 sparse=0 partition=0 split=0 reserve=$5 2>gpurun_out/r6s23/err.txt | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); d['occupancy_model']='$4' or 'host function'; print(json.dumps(d))" >> $OUT ) || { echo "FAILED $*"; tail -5 gpurun_out/r6s23/err.txt; }
}
for P in 8 2; do
  for rate in 100 50; do
    for rep in 1 2; do
      one $P 0 $rate "" -1
      one $P 0 $rate "" 0
      one $P 0 $rate 28:rccl -1
      one $P 0 $rate 28:rccl 0
    done
  done
done
python - <<'PY'
import json, collections
acc = collections.defaultdict(list)
for ln in open("gpurun_out/r6s23/solo_plain_shards.jsonl"):
    d = json.loads(ln)
    acc[(d["ranks"], d["rank"], d["link_model"]["GBps_per_link"], d["occupancy_model"], d["comm_reserve"])].append(d["ms_per_step"])
for k in sorted(acc, key=lambda k: (-k[0], k[1], -k[2], k[3], k[4])):
    print("plain shards P %d rank %d rate %g model %-13s comm_reserve %2d | ms per step %s" % (k + (" ".join("%.3f" % v for v in acc[k]),)))
PY
unset QBH_STUB_SOLO_KERNEL
timeout 900 python -m pytest tests/test_gpu_native_ranks.py tests/test_gpu_dist.py tests/test_gpu_configs.py -q -x -m gpu -k "plain or sharded or four_ranks or fall_back" 2>&1 | tail -4
R=$PWD; O=$R/gpurun_out/r6s23
export TMPDIR=/tmp/stub_k30; mkdir -p $TMPDIR
QBH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1200 python bench.py --workload kagome_30 --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1 > $O/kagome_30_2_ranks.log 2>&1
grep '"metric"' $O/kagome_30_2_ranks.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kagome_30 2 ranks e0 %.12f steps %s' % (d['e0'], d.get('lanczos_steps_to_converge')))"
