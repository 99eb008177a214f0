#!/bin/bash
# why did bench --gpus 2 at C3 report gather_parts 1?  default gather_parts through the C++ rank program (small) and through bench.py (4x3, kron_split forced by size? no: hubbard_4x3 stays unsplit) 
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s13; mkdir -p $O
cd $R
make -C tests/stub_rccl > /dev/null 2>&1
python - <<'PY'
import os, subprocess, tempfile, numpy as np, sys
sys.path.insert(0, "tests")
import quantum_basis_amd as q
from quantum_basis_amd import lattices
from test_cxx_adaptor import _build
tmp = tempfile.mkdtemp()
exe = _build(tmp, "sharded_main")
G = q.csr_mat.hubbard(8, 4, 4, lattices.square(4, 2), t=1.0, U=1.1, opts=q.make_opts(kron_split=0, value_dict=0, real_fast_path=0))
ia, ja, val = G.download(); dim = G.dim; G.destroy()
path = os.path.join(tmp, "csr.bin")
with open(path, "wb") as f:
    np.array([dim, len(ja), 0], dtype=np.int64).tofile(f); ia.astype(np.int64).tofile(f); ja.astype(np.int64).tofile(f); val.tofile(f)
env = dict(os.environ, QBH_RCCL_LIB=os.path.join(os.getcwd(), "tests/stub_rccl/librccl_stub.so"), TMPDIR=tmp)
for args in (["plain=1", "kron=70"], ["plain=1", "kron=70", "uniform"]):
    uid = os.path.join(tmp, "uid%d" % len(args))
    ps = [subprocess.Popen([exe, path, str(r), "2", uid] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    for p in ps:
        out, _ = p.communicate(timeout=300)
        print(args, [l for l in out.splitlines() if l.startswith(("OK", "ERR"))])
PY
