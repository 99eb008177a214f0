#!/bin/bash
# round 6 session 12: FINAL sources -- C3 kernel stats + PMC traffic (tools/profile_bench.sh), the kagome-30 default line (C2), the C4-substitute line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s12; mkdir -p $O
cd $R
bash tools/profile_bench.sh r6_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace|c16" > $O/profile.log 2>&1
tail -30 $O/profile.log | cut -c1-300
for wl in kagome_30 hubbard_4x5_n5; do
  timeout 900 python bench.py --workload $wl --steps 20 --warmup 5 --processes 1 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate 2>/dev/null | grep '"metric"' > $O/${wl}_default.json
  python - $O/${wl}_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
print(d["config"]["workload"], d["value"], d["ms_per_step"], r["frac"], r.get("ms_per_launch"), d["config"].get("basis_internal"), d.get("e0"))
PY
done
