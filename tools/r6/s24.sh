#!/bin/bash
# round 6 session 24: FINAL sources (ABI 601) -- C3 kernel stats + PMC traffic (tools/profile_bench.sh), kagome-30 and C4-substitute default lines,
# then the driver's own command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s24; mkdir -p $O
cd $R
bash tools/profile_bench.sh r6b_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace|c16" > $O/profile.log 2>&1
tail -30 $O/profile.log | cut -c1-300
for wl in kagome_30 hubbard_4x5_n5; do
  timeout 900 python bench.py --workload $wl --steps 20 --warmup 5 --processes 1 --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate 2>/dev/null | grep '"metric"' > $O/${wl}_default.json
  python - $O/${wl}_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
print(d["config"]["workload"], d["value"], d["ms_per_step"], r["frac"], r.get("ms_per_launch"), d["config"].get("basis_internal"), d.get("e0"))
PY
done
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_command.json 2> $O/driver_command.err ) 2>&1 | tail -3
python - $O/driver_command.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("driver command:", d["value"], "it/s", d["ms_per_step"], "ms/step; SpMV", r["ms_per_launch"], "ms frac", r["frac"], "min/max", r.get("frac_min"), r.get("frac_max"), "traffic_stale", r.get("traffic_stale"), "bare", d.get("bare_spmv", {}).get("frac"))
print("processes:", [(p.get("value"), p.get("roofline", {}).get("frac") if isinstance(p.get("roofline"), dict) else p.get("frac")) for p in d.get("processes", {}).get("runs", [])] if isinstance(d.get("processes"), dict) else d.get("processes"))
PY
