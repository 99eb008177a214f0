#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
make -C tests/stub_rccl > /dev/null 2>&1
export QBH_RCCL_LIB=$R/tests/stub_rccl/librccl_stub.so HSA_ENABLE_IPC_MODE_LEGACY=0 PROBE_MAXIT=6
i=0
for cfg in "0 1 kron_cols16=0" "0 1 tile_fold=0" "0 1 wave_walk=2" "0 0"; do
i=$((i+1))
export TMPDIR=/tmp/run$i; mkdir -p $TMPDIR
echo "== C3 P=4 [gather_parts kron_split more] = [$cfg]"
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29700+i)) tools/r5/parts_probe.py hubbard_4x4_half $cfg 2>&1 | grep "after\|ran \|fault\|rccl stub" | sort | head -12
rm -rf $TMPDIR
done
