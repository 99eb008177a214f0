#!/bin/bash
# round 5 session 4: generic mopr tests; C3 evidence on the final sources (plain line + kernel stats + PMC traffic on one box);
# the driver-style default line; C5 locality of the real column stream; lines of the other stored configs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s4; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_mopr.py -x -q -m gpu > $O/pytest_mopr.log 2>&1; echo "pytest rc $?" >> $O/pytest_mopr.log; tail -4 $O/pytest_mopr.log
bash tools/profile_bench.sh r5_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace|c16" > $O/profile_c3.log 2>&1; tail -30 $O/profile_c3.log | cut -c1-300
cd $R
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python tools/c5_locality.py > $O/c5_locality.txt 2>&1; cat $O/c5_locality.txt
timeout 900 python bench.py --workload kagome_30 --processes 1 --no-cpu-baseline --no-locate --no-fast-path --no-matrix-free > $O/kagome_30_default.json 2> $O/kagome_30_default.err
timeout 900 python bench.py --workload kagome_30 --site-cut 18 --processes 1 --no-cpu-baseline --no-locate --no-fast-path --no-matrix-free > $O/kagome_30_cut18.json 2> $O/kagome_30_cut18.err
timeout 900 python bench.py --workload hubbard_4x5_n5 --processes 1 --no-cpu-baseline --no-locate > $O/hubbard_4x5_n5.json 2> $O/hubbard_4x5_n5.err
for f in $O/*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]
    print({k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic"), r.get("traffic_stale"), d.get("processes",{}).get("frac"))
    print("  fast", {k:(d.get("fast_path") or {}).get(k) for k in ("value",)}, ((d.get("fast_path") or {}).get("roofline") or {}).get("frac"), "mf", {k:v.get("lanczos_iters_per_s") for k,v in d.items() if k.startswith("matrix_free")})
except Exception as e: print("ERR", e)
PY
done
