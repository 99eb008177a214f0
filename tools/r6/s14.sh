#!/bin/bash
# round 6 session 14: kernel statistics of ONE rank of 8 (and of 2) alone with modelled peers: where the per-shard overhead goes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s14; mkdir -p $O
export QBH_RCCL_LIB=$R/tests/stub_rccl/librccl_stub.so PYTHONPATH=$R TMPDIR=/tmp QBH_STUB_SOLO=100000
cd /tmp
for P in 8 2; do
  rm -rf /tmp/prof_solo$P
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_solo$P -o s -- python3 $R/tools/solo_rank.py hubbard_4x4_half $P 0 steps=20 warmup=4 parts=4 > $O/solo_$P.log 2>&1
  python3 $R/tools/stats_summary.py /tmp/prof_solo$P "QBH_STUB_SOLO=100000 python tools/solo_rank.py hubbard_4x4_half $P 0 steps=20 warmup=4 parts=4" > $O/solo_${P}_kernel_stats.txt
  grep '^{' $O/solo_$P.log | tail -1 >> $O/solo_${P}_kernel_stats.txt
  head -24 $O/solo_${P}_kernel_stats.txt | cut -c1-170
done
