#!/bin/bash
# round 6 session 7: C3 evidence on the round-6 sources, one box: the driver's own command (3 fresh child processes), then kernel stats + PMC
# traffic of the same workload (tools/profile_bench.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s07; mkdir -p $O
cd $R
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
t1=$(date +%s)
echo "wall seconds of: python3 bench.py --gpus 1 --steps 20 --warmup 5 = $((t1 - t0))" | tee $O/wall.txt
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print({k:d.get(k) for k in ("value","ms_per_step","e0","steps","warmup")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic"), r.get("traffic_stale"), d.get("processes",{}))
print("bare", d.get("bare_spmv",{}).get("frac"), "fast", d.get("fast_path",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"), "locate", d.get("locate_E0_lanczos",{}).get("seconds_total"))
PY
bash tools/profile_bench.sh r6_c3 "hubbard_4x4_half|wave|plain|kron_sliced|inplace|c16"
