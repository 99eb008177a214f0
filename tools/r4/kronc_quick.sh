#!/bin/bash
# QBH_KRON_CODED=2 on C3: kernel times + read requests of the two passes, one configuration
mkdir -p gpurun_out/r4kronc
O=$GRAFT_REPO_ROOT/gpurun_out/r4kronc
R=$GRAFT_REPO_ROOT
TAG=${1:-quick}
cd /tmp && export TMPDIR=/tmp
export QBH_KRON_CODED=2
rm -rf /tmp/kp; mkdir -p /tmp/kp /tmp/kq/g1 /tmp/kq/g2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kp -o k -- python3 $R/bench.py --format fast --steps 20 --warmup 3 --no-cpu-baseline --no-matrix-free > /tmp/kp/log 2>&1
echo "== $TAG" | tee -a $O/quick.txt
python3 $R/tools/stats_summary.py /tmp/kp "quick" | grep -E "kronc_far|kronc_near|tile_re|axpy_norm" | tee -a $O/quick.txt
grep '"metric"' /tmp/kp/log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s', d['value'], 'spmv', d['roofline']['ms_per_launch'], 'e0', d['e0'])" | tee -a $O/quick.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/kq/g1 -o p -- python3 $R/bench.py --format fast --steps 10 --warmup 2 --no-converge --no-cpu-baseline --no-matrix-free > /tmp/kq/g1/log 2>&1
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d /tmp/kq/g2 -o p -- python3 $R/bench.py --format fast --steps 10 --warmup 2 --no-converge --no-cpu-baseline --no-matrix-free > /tmp/kq/g2/log 2>&1
for pat in "%k_kronc_far%" "%k_kronc_near%"; do python3 $R/tools/pmc_summary.py /tmp/kq "$pat"; done | tee -a $O/quick.txt
