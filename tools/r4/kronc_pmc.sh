#!/bin/bash
# QBH_KRON_CODED=2 on C3: fabric bytes of the two passes
mkdir -p gpurun_out/r4kronc
O=$GRAFT_REPO_ROOT/gpurun_out/r4kronc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export QBH_KRON_CODED=2
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/kp$i; mkdir -p /tmp/kp$i
  timeout 600 rocprofv3 --pmc $grp -d /tmp/kp$i -o p -- python3 $R/bench.py --format fast --steps 10 --warmup 2 --no-converge --no-cpu-baseline --no-matrix-free > /tmp/kp$i/log 2>&1
  echo "rc $?"
done
mkdir -p /tmp/kpall; for j in 1 2 3 4; do mkdir -p /tmp/kpall/g$j; cp -r /tmp/kp$j/* /tmp/kpall/g$j/; done
for pat in "%k_kronc_far%" "%k_kronc_near%" "%k_kron_tile_re%"; do python3 $R/tools/pmc_summary.py /tmp/kpall "$pat"; done > $O/pmc_form2.txt 2>&1
cat $O/pmc_form2.txt
