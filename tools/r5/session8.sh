#!/bin/bash
# round 5 session 8: whole GPU tier + smoke(); tile shapes of the producer pass (k_axpy_norm_tile8) A/B through whole bench lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s8; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu -x --durations=12 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -18 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for round in 1 2; do
for v in 32x8 16x16 8x32 64x4; do
  QBHIP_LIBRARY=$R/tools/lab/variants/r5_tile_$v.so timeout 600 python bench.py --processes 1 --steps 40 --warmup 5 --no-converge --no-cpu-baseline --no-locate --no-fast-path --no-matrix-free 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['ms_per_step'], d['roofline']['ms_per_launch'], round(d['ms_per_step']-d['roofline']['ms_per_launch'],3))" | tee -a $O/tile_shapes.txt
done; done
