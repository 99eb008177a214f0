#!/bin/bash
# round 5 session 42: bench.py with the Lanczos vectors allocated before the operator (its default now): default lines, the old order beside it,
# and the C3 kernel stats + PMC taken with this bench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s42; mkdir -p $O
cd $R
python tools/src_hash.py | tee $O/src_hash.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --vectors-after-operator --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate > $O/bench_vectors_after.json 2> $O/bench_vectors_after.err
timeout 600 python bench.py --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate > $O/bench_default_2.json 2> $O/bench_default_2.err
for f in bench_default bench_vectors_after bench_default_2; do python - $O/$f.json $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print(sys.argv[2], {k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic_stale"), d.get("processes",{}).get("frac"), d["config"].get("lanczos_vectors"))
PY
done
bash tools/profile_bench.sh r5_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace|c16" > $O/profile_c3.log 2>&1; tail -4 $O/profile_c3.log | cut -c1-160
