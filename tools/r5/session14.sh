#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
make -C tests/stub_rccl > /dev/null 2>&1
export QBH_RCCL_LIB=$R/tests/stub_rccl/librccl_stub.so TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i)) tools/r5/parts_probe.py hubbard_4x5_n4 0 1 2>&1 | grep "after\|set_comm" | sort
done
export QBH_DEBUG=trace_create=1
for i in 5 6; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i)) tools/r5/parts_probe.py hubbard_4x5_n4 0 1 2>&1 | grep "after\|set_comm" | sort
done
