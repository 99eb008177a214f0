#!/bin/bash
# Evidence for bench.py's roofline block, taken in ONE gpurun call on ONE box: (0) a plain bench.py line, (1) rocprofv3
# --kernel-trace --stats of the same command, (2) separate PMC passes (counters only, one group per run) -> per-SpMV HBM
# traffic entry stamped with the hash of the kernel sources (tools/traffic_entry.py).
# usage: tools/profile_bench.sh <tag> <traffic key> [bench args...]   -> gpurun_out/<tag>_*.{txt,json}
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; KEY=$2; shift; shift
export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
# --processes 1: under rocprofv3 a process must not start children (the profiler has initialised the GPU before main() runs)
ARGS="--steps 20 --warmup 3 --no-converge --no-cpu-baseline --no-fast-path --no-matrix-free --no-locate --processes 1 $*"
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
python3 $R/bench.py $ARGS 2>/dev/null | grep '"metric"' > $OUT/${TAG}_bench_line.json
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/stats -o s -- python3 $R/bench.py $ARGS > /tmp/prof_$TAG/stats.log 2>&1
python3 $R/tools/stats_summary.py /tmp/prof_$TAG/stats "python bench.py $ARGS" > $OUT/${TAG}_kernel_stats.txt
grep '"metric"' /tmp/prof_$TAG/stats.log | tail -1 >> $OUT/${TAG}_kernel_stats.txt
echo "# plain bench.py line of the same command on the same box (no profiler):" >> $OUT/${TAG}_kernel_stats.txt
cat $OUT/${TAG}_bench_line.json >> $OUT/${TAG}_kernel_stats.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp -d /tmp/prof_$TAG/g$i -o p -- python3 $R/bench.py $ARGS > /tmp/prof_$TAG/g$i.log 2>&1
done
{ echo "# rocprofv3 --pmc passes (one counter group per run) of: python bench.py $ARGS"; echo "# mean per dispatch; FETCH_SIZE/WRITE_SIZE are in KiB as reported (see DESIGN.md for the gfx950 x2 read correction)"; for pat in "%spmv%" "%k_kronc%" "%k_kron_tile%" "%k_zero_cut%" "%k_kron_combine%" "%k_axpy_norm%" "%mf_hubbard%" "%mf_heis%" "%k_mf_sector%" "%k_sec_re%"; do python3 $R/tools/pmc_summary.py /tmp/prof_$TAG "$pat"; done; } > $OUT/${TAG}_pmc.txt
python3 $R/tools/traffic_entry.py /tmp/prof_$TAG "$KEY" "profiles/${TAG}_pmc.txt" $OUT/${TAG}_traffic_entry.json > /dev/null
head -12 $OUT/${TAG}_kernel_stats.txt; tail -3 $OUT/${TAG}_kernel_stats.txt | cut -c1-400
cat $OUT/${TAG}_traffic_entry.json | head -40
