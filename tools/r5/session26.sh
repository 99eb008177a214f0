#!/bin/bash
# round 5 session 26: the non-abelian group test of the orbit order; the default line with its bare_spmv block
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s26; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_hubrepr.py -x -q -m gpu -k "non_abelian or matrix_free" 2>&1 | tail -25 | tee $O/pytest.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print({k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic"), r.get("traffic_stale"), d.get("processes",{}).get("frac"))
print(d.get("bare_spmv"))
PY
