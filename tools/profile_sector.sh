#!/bin/bash
# rocprofv3 evidence for the matrix-free sector kernel: kernel-trace stats, then separate PMC passes (counters only).
# usage: tools/profile_sector.sh <tag> <workload>   -> gpurun_out/<tag>_kernel_stats.txt, gpurun_out/<tag>_pmc.txt
set -u
R=/root/repo
TAG=$1; WL=$2
export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
ARGS="--workload $WL --steps 10 --warmup 3 --processes 1 --no-locate"
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/stats -o s -- python3 $R/bench.py $ARGS > /tmp/prof_$TAG/stats.log 2>&1
python3 $R/tools/stats_summary.py /tmp/prof_$TAG/stats "python bench.py $ARGS" > $OUT/${TAG}_kernel_stats.txt
grep '"metric"' /tmp/prof_$TAG/stats.log | tail -1 >> $OUT/${TAG}_kernel_stats.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp -d /tmp/prof_$TAG/g$i -o p -- python3 $R/bench.py $ARGS > /tmp/prof_$TAG/g$i.log 2>&1
done
{ echo "# rocprofv3 --pmc passes (one counter group per run) of: python bench.py $ARGS"; echo "# mean per dispatch; FETCH_SIZE/WRITE_SIZE in KiB as reported (gfx950: read bytes = 2 * FETCH_SIZE * 1024, see DESIGN.md)"; python3 $R/tools/pmc_summary.py /tmp/prof_$TAG "%k_mf_sector%"; python3 $R/tools/pmc_summary.py /tmp/prof_$TAG "%k_sec_%"; } > $OUT/${TAG}_pmc.txt
head -12 $OUT/${TAG}_kernel_stats.txt | cut -c1-150
cat $OUT/${TAG}_pmc.txt
