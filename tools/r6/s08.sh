#!/bin/bash
# round 6: the driver's own command on a fresh box (one gpurun call = one box); TAG names the output
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; O=$R/gpurun_out/r6s08; mkdir -p $O
cd $R
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$TAG.json 2> $O/driver_cmd_$TAG.err
t1=$(date +%s)
echo "wall seconds = $((t1 - t0))"
python - $O/driver_cmd_$TAG.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]; p=d.get("processes",{})
print({k:d.get(k) for k in ("value","ms_per_step")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic_stale"), "frac", p.get("frac"), "value", p.get("value"), "step-launch", [round(a-b,2) for a,b in zip(p.get("ms_per_step",[]), p.get("ms_per_launch",[]))])
print("bare", d.get("bare_spmv",{}).get("frac"), "fast", d.get("fast_path",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"), "locate", (d.get("locate_E0_lanczos") or {}).get("seconds_total"))
PY
