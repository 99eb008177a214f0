#!/bin/bash
# the sliced coded split (QBH_KRON_CODED=2): parity tests, then C3 in the library's default format, three forms side by side
mkdir -p gpurun_out/r4kronc
O=gpurun_out/r4kronc
timeout 900 python -m pytest tests/test_gpu_kron.py -x -q -m gpu -k "coded_real_form" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
for form in 0 2 1 2 0; do
  QBH_KRON_CODED=$form timeout 600 python bench.py --format fast --steps 40 --warmup 5 --no-cpu-baseline --no-matrix-free > $O/bench_form${form}.json 2> $O/bench_form${form}.err
  echo "form $form rc $?"
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_form${form}.json").read().strip().splitlines()[-1])
    print("form $form", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "spmv", d["roofline"].get("ms_per_launch"), "frac", d["roofline"]["frac"], "e0", d.get("e0"))
except Exception as e:
    print("form $form parse failed", e)
PY
done
cd /tmp && export TMPDIR=/tmp
QBH_KRON_CODED=2 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --format fast --steps 20 --warmup 3 --no-cpu-baseline --no-matrix-free > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof_bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" > $O/kernel_stats_form2.txt && cat $O/kernel_stats_form2.txt
