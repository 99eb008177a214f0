#!/bin/bash
# round 5 session 35: large arrays as one physical allocation each (device_alloc through the virtual-memory API): the full GPU tier, smoke, and the
# default line; then alternating processes with the switch off / on in the product form
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s35; mkdir -p $O
cd $R
python tools/src_hash.py | tee $O/src_hash.txt
make -C tests/stub_rccl > /dev/null 2>&1
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -14 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1"
for i in 1 2 3 4; do
  for v in 0 1; do
    QBH_DEBUG=vmm=$v timeout 200 python bench.py $ARGS 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vmm $v round $i frac', d['roofline']['frac'], 'ms', d['roofline']['ms_per_launch'], 'step', d['ms_per_step'])" || echo "vmm $v round $i FAILED"
  done
done 2>&1 | tee $O/vmm_ab_product.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print({k:d.get(k) for k in ("value","ms_per_step","e0")}, r.get("frac"), r.get("ms_per_launch"), r.get("traffic"), r.get("traffic_stale"), d.get("processes",{}).get("frac"))
print(d.get("bare_spmv",{}).get("frac"), d.get("fast_path",{}).get("value"), d.get("locate_E0",{}).get("seconds_total"))
PY
