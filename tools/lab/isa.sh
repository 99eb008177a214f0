#!/bin/bash
# disassemble one kernel of libqbhip's kernel object: tools/lab/isa.sh '<mangled-name regex>' > out.s
set -e
D=$(mktemp -d)
cp /root/repo/quantum_basis_amd/csrc/build/qbh_kernels.hip.o $D/k.o
(cd $D && /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading k.o > /dev/null 2>&1 || true)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $D/k.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | awk -v pat="$1" '$0 ~ "^[0-9a-f]+ <" pat ">:" {p=1} p&&/s_endpgm/{print; exit} p{print}'
rm -rf $D
