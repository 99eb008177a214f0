#!/bin/bash
# round 5 session 10: the table kernel with chunk-major items (can the Infinity Cache serve the neighbour rows?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s10; mkdir -p $O
cd $R
for cfg in "" "mf_chunk=2048,mf_order=1" "mf_chunk=2048,mf_window=6144,mf_order=1" "mf_chunk=4096,mf_order=1" "mf_chunk=1024,mf_window=4096,mf_order=1" "mf_chunk=2048,mf_order=0" "mf_chunk=2048,mf_window=6144,mf_order=0"; do
  QBH_DEBUG="$cfg" timeout 600 python bench.py --format fast --processes 1 --steps 30 --warmup 5 --no-converge --no-cpu-baseline --no-locate --no-matrix-free 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('[$cfg]', 'it/s', d['value'], 'ms_spmv', d['roofline']['ms_per_launch'], 'e0', d.get('e0'))" | tee -a $O/table_chunk_major.txt
done
