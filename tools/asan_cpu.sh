#!/bin/bash
# Host-side AddressSanitizer run (CPU build container only: GPU ASan is not available on the pool).  Builds the same sources
# with -fsanitize=address on the host pass and runs the CPU tests that exercise host code: argument validation and the
# Hermitian check of qbh_csr_create, qbh_hess_eigen, the checkpoint protocol / CRC-32, struct layouts.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/asan_build
cd $R/quantum_basis_amd/csrc
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address -fno-omit-frame-pointer -I../../include -I. \
    -shared -o $R/tools/asan_build/libqbhip_asan.so -x hip qbh_kernels.hip qbh_kron.hip qbh_kronc.hip qbh_gen.hip qbh_build.hip qbh_mopr.hip qbh_reorder.hip \
    qbh_api.cpp qbh_split.cpp qbh_commattach.cpp qbh_spmv.cpp qbh_solvers.cpp qbh_comm.cpp qbh_ckpt.cpp qbh_hess.cpp -ldl
cd $R
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
LD_PRELOAD=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1) \
QBHIP_LIBRARY=tools/asan_build/libqbhip_asan.so \
python -m pytest tests/test_abi.py tests/test_ckpt.py tests/test_integration_patch.py -x -q -m "not gpu"
