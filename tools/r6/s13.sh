#!/bin/bash
# round 6 session 13: is the final library slower than the one of session 7?  The two trees alternating on one box, one fresh process per line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s13; mkdir -p $O
C="--steps 20 --warmup 5 --processes 1 --no-cpu-baseline --no-fast-path --no-matrix-free --no-converge --no-locate"
for i in 1 2 3 4; do
  (cd $R && timeout 600 python bench.py $C 2>/dev/null | grep '"metric"' > $O/final_$i.json)
  (cd $R/tools/r6/s07tree && timeout 600 python bench.py $C 2>/dev/null | grep '"metric"' > $O/s07_$i.json)
done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read()); r=d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], r["ms_per_launch"], round(d["ms_per_step"]-r["ms_per_launch"],2), r["frac"])
    except Exception as e: print(f, "ERR", e)
PY
