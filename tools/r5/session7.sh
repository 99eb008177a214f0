#!/bin/bash
# round 5 session 7 (final sources): whole GPU tier, then the C3 evidence on ONE box: plain line + kernel stats + PMC traffic,
# and the driver-style default line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s7; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -18 $O/pytest.log
bash tools/profile_bench.sh r5_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace|c16" > $O/profile_c3.log 2>&1; tail -12 $O/profile_c3.log | cut -c1-200
cd $R
python tools/src_hash.py
