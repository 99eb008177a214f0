#!/bin/bash
# round 5 session 25: the full GPU tier again (session 24 stopped at a missing stream sync in the C4 test) + smoke, and the counters of C4 as written
# (the summary tools cut kernel names at the "(" of "(anonymous namespace)")
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s25; mkdir -p $O
cd $R
python tools/src_hash.py | tee $O/src_hash.txt
make -C tests/stub_rccl > /dev/null 2>&1
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x --durations=12 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -18 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd $R; bash tools/profile_bench.sh r5_c4_half "hubbard_4x5_half_k00_mf|matrix_free|plain|real" --workload hubbard_4x5_half_k00_mf > $O/profile_c4.log 2>&1; tail -30 $O/profile_c4.log | cut -c1-160
