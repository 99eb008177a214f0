#!/bin/bash
# bench.py under torch.distributed.run with 2 ranks sharing the one GPU (gloo staging): the N > 1 code path of the headline workload
R=/root/repo
O=$R/gpurun_out/r4s15
mkdir -p $O
cd $R
QBH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-fast-path --no-matrix-free > $O/bench_2ranks_gloo.log 2>&1
grep '"metric"' $O/bench_2ranks_gloo.log | tail -1 > $O/bench_2ranks_gloo.json
python - <<'PY'
import json
j=json.load(open('/root/repo/gpurun_out/r4s15/bench_2ranks_gloo.json'))
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'e0', j['e0'], 'steps', j['lanczos_steps_to_converge'], 'exchange', j['config']['exchange'], 'kron', j['config']['kron_split'])
for r in j.get('per_rank', []): print(r)
PY
tail -5 $O/bench_2ranks_gloo.log | cut -c1-300
