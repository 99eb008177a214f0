#!/bin/bash
# Evidence for bench.py's roofline block: (1) rocprofv3 --kernel-trace --stats of the bench
# command, (2) separate PMC passes (counters only) for FETCH_SIZE / WRITE_SIZE / L2 traffic.
# usage: tools/profile_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>_*.txt
set -u
R=/root/repo
TAG=$1; shift
export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
ARGS="--steps 10 --warmup 3 --no-converge --no-cpu-baseline $*"
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/stats -o s -- python3 $R/bench.py $ARGS > /tmp/prof_$TAG/stats.log 2>&1
python3 $R/tools/stats_summary.py /tmp/prof_$TAG/stats "python bench.py $ARGS" > $OUT/${TAG}_kernel_stats.txt
grep '"metric"' /tmp/prof_$TAG/stats.log | tail -1 >> $OUT/${TAG}_kernel_stats.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d /tmp/prof_$TAG/g$i -o p -- python3 $R/bench.py $ARGS > /tmp/prof_$TAG/g$i.log 2>&1
done
{ echo "# rocprofv3 --pmc passes (one counter group per run) of: python bench.py $ARGS"; echo "# mean per dispatch of the SpMV kernel; FETCH_SIZE/WRITE_SIZE are in KiB as reported (see DESIGN.md for the gfx950 x2 read correction)"; python3 $R/tools/pmc_summary.py /tmp/prof_$TAG; python3 $R/tools/pmc_summary.py /tmp/prof_$TAG "%k_kron_tile%"; python3 $R/tools/pmc_summary.py /tmp/prof_$TAG "%k_zero_cut%"; python3 $R/tools/pmc_summary.py /tmp/prof_$TAG "%mf_hubbard%"; python3 $R/tools/pmc_summary.py /tmp/prof_$TAG "%mf_heis%"; } > $OUT/${TAG}_pmc.txt
tail -3 $OUT/${TAG}_kernel_stats.txt
cat $OUT/${TAG}_pmc.txt
