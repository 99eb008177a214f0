#!/bin/bash
# round 6 session 64: the final library (ABI 601, be3b274c) against the library built from the sources this session started from (0c17238, ABI 600) alternating on ONE box,
# one fresh process per line: did anything on the one-GPU path move?  (QBHIP_LIBRARY selects the .so; it ignores the newer option fields)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s64; mkdir -p $O
C="--steps 20 --warmup 5 --processes 1 --no-cpu-baseline --no-fast-path --no-matrix-free --no-converge --no-locate"
cd $R
for i in 1 2 3 4 5; do
  timeout 600 python bench.py $C 2>/dev/null | grep '"metric"' > $O/final_$i.json
  QBHIP_LIBRARY=$R/tools/lab/ab/libqbhip_r6start.so timeout 600 python bench.py $C 2>/dev/null | grep '"metric"' > $O/r6start_$i.json
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read()); r=d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], r["ms_per_launch"], round(d["ms_per_step"]-r["ms_per_launch"],2), r["frac"])
    except Exception as e: print(f, "ERR", e)
PY
