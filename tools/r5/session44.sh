#!/bin/bash
# round 5 session 44: the GPU tier once more on another box (flakiness check of the final tree), smoke, and the driver's own command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s44; mkdir -p $O
cd $R
python tools/src_hash.py | tee $O/src_hash.txt
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep "Elapsed (wall" $O/bench_driver_cmd.err
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print({k:d.get(k) for k in ("value","ms_per_step","e0","steps","warmup")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic"), r.get("traffic_stale"), d.get("processes",{}).get("frac"))
PY
