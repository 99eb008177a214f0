"""Worker for the multi-process tests (spawned by test_dist_cpu.py / test_gpu_dist.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init(rank, world, port, backend):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def cpu_sharded_lanczos(rank, world, port, case_name, steps, out_dir, cuts=None):
    """The sharded three-term recurrence with the real exchange hooks (ShardComm, gloo, CPU
    tensors); the shard-local arithmetic is done by the ORACLE here (test double for the HIP
    kernels, which cannot run without a GPU).  Checks the partition, the gather layout and
    the scalar reductions against the unsharded oracle."""
    import torch
    dist = _init(rank, world, port, "gloo")
    import helpers
    from oracle import qb_oracle as qo
    from quantum_basis_amd import dist as qdist

    d, ia, ja, val, sym = helpers.case(case_name)
    full = qo.Csr(d, ia, ja, val, sym).expand_full()
    comm = qdist.ShardComm(d, rank=rank, world=world, device=torch.device("cpu"), cuts=cuts)
    r0, r1 = comm.ranges[rank]
    n = r1 - r0
    # shard-local CSR (rows r0..r1, global columns)
    sia = full.ia[r0:r1 + 1] - full.ia[r0]
    sja = full.ja[full.ia[r0]:full.ia[r1]]
    sval = full.val[full.ia[r0]:full.ia[r1]]

    def gather(x_loc):
        buf = np.zeros(comm.nblk, dtype=np.complex128)
        buf[:n] = x_loc
        comm.xsend.copy_(torch.from_numpy(buf.view(np.float64)))
        assert comm._allgather(None, 0) == 0, comm.errors
        return comm.xfull.numpy().view(np.complex128)

    def allreduce(vals):
        comm.scal[:len(vals)] = torch.tensor(vals, dtype=torch.float64)
        assert comm._allreduce(None, 0, len(vals)) == 0, comm.errors
        return comm.scal[:len(vals)].tolist()

    def spmv(x_loc):
        xf = gather(x_loc)
        y = np.zeros(n, dtype=np.complex128)
        for i in range(n):
            sl = slice(sia[i], sia[i + 1])
            y[i] = np.dot(sval[sl], xf[sja[sl]])
        return y

    x_full = qo.vec_randomize(d, 1)
    v_prev = np.zeros(n, dtype=np.complex128)
    v = x_full[r0:r1].copy()
    a, b = [], [0.0]
    for m in range(steps):
        w = spmv(v) - b[-1] * v_prev
        (am,) = allreduce([np.vdot(v, w).real])
        w -= am * v
        (nn,) = allreduce([np.vdot(w, w).real])
        bm = np.sqrt(nn)
        a.append(am)
        b.append(bm)
        v_prev, v = v, w / bm
    if rank == 0:
        np.save(os.path.join(out_dir, "ab.npy"), np.array([a, b[1:]]))
    dist.barrier()
    dist.destroy_process_group()


def gpu_sharded_solver(rank, world, port, backend, out_dir, matrix_free=False, shape="4x2"):
    """Real thing on the GPU: every rank builds its row shard on cuda:0 (single-GPU rig: the
    ranks share one device and exchange through gloo + host staging; with backend nccl and one
    rank the RCCL calls themselves are exercised) and runs qbh_lanczos_dev / CG under the
    communicator hooks."""
    import torch
    dist = _init(rank, world, port, backend)
    torch.cuda.set_device(0)
    import quantum_basis_amd as q
    from quantum_basis_amd import dist as qdist, lattices

    # 4x3 (924 x 924 configurations) is large enough for the row-staged matrix-free kernel; with 5 ranks the shard
    # boundaries fall inside a row of the N_up x N_dn layout
    L, ne, bonds, dim = {"4x2": (8, 4, lattices.square(4, 2), 4900), "4x3": (12, 6, lattices.square(4, 3), 853776),
                         "4x5n3": (20, 3, lattices.square(4, 5), 1299600)}[shape]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        opts = q.make_opts(device=0, stream=stream.cuda_stream)
        nblk, ranges = qdist.row_partition(dim, world)
        r0, r1 = ranges[rank]
        A = q.csr_mat.hubbard(L, ne, ne, bonds, rows=(r0, r1), opts=opts, matrix_free=matrix_free)
        comm = qdist.ShardComm(dim, rank=rank, world=world, device=torch.device("cuda", 0), stream=stream).attach(A)
        res = q.locate_E0_lanczos(A, nev=2, ncv=1, maxit=400)
        assert not comm.errors, comm.errors
        assert comm.n_packed > 100, comm.n_packed        # real operator + real vectors: 8-byte wire format
        if backend == "nccl":
            assert comm.n_async > 100, comm.n_async      # every SpMV went through the asynchronous RCCL all-gather
        x = q.vec_randomize(A, seed=1)                 # shard of the global start vector
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
        np.save(os.path.join(out_dir, "vec_%d.npy" % rank), res.eigenvecs)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array([res.E0, res.E1, res.steps["E0"], res.steps["V0"], res.steps["E1"]]))
            np.save(os.path.join(out_dir, "hess.npy"), res.hessenberg_E0)
    dist.barrier()
    dist.destroy_process_group()


def gpu_sharded_kron(rank, world, port, backend, out_dir, mixed=False, parts=0, native=False):
    """Row shards of WHOLE MAJOR INDICES of a product-basis operator (complex128 CSR, Hubbard 4x3: dim 853,776, S = 924): every
    shard is split in place, the ranks exchange the TILED copies of their blocks, the near pass runs while the gather is in
    flight.  mixed: rank 1 creates its shard unsplit -- the ranks then agree on the plain exchange and rank 0 merges its parts
    back into a CSR (kron_restore).  parts: the gather in that many band ranges (qbh_comm.allgather_part_begin), the far pass of
    a range following its piece; native: the library's own RCCL communicator (one rank: every piece is the rank's own)."""
    import torch
    dist = _init(rank, world, port, backend)
    torch.cuda.set_device(0)
    import quantum_basis_amd as q
    from quantum_basis_amd import dist as qdist, lattices

    L, ne, bonds, dim, S = 12, 6, lattices.square(4, 3), 853776, 924
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        split = 0 if (mixed and rank == 1) else 2              # 2: split whatever has the structure (1 leaves small operators alone)
        opts = q.make_opts(device=0, stream=stream.cuda_stream, value_dict=0, real_fast_path=0, kron_split=split, gather_parts=parts)
        cuts = qdist.kron_row_cuts(dim, S, world)
        r0, r1 = int(cuts[rank]), int(cuts[rank + 1])
        A = q.csr_mat.hubbard(L, ne, ne, bonds, rows=(r0, r1), opts=opts)
        assert A.info().kron_minor == (S if split else 0)
        if native:
            comm = qdist.NativeComm(dim, rank=rank, world=world, cuts=cuts if world > 1 else None).attach(A)
        else:
            comm = qdist.ShardComm(dim, rank=rank, world=world, device=torch.device("cuda", 0), stream=stream, cuts=cuts, parts=bool(parts)).attach(A)
        assert A.info().kron_minor == (0 if mixed else S)          # agreed by all ranks, or merged back
        assert A.info().gather_parts == (parts if parts else 1)
        res = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=400)
        if not native:
            assert not comm.errors, comm.errors
            assert (comm.n_parts_begun > 0) == bool(parts), comm.n_parts_begun
        x = q.vec_randomize(A, seed=1)
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
        np.save(os.path.join(out_dir, "vec_%d.npy" % rank), res.eigenvecs)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array([res.E0, res.steps["E0"], res.steps["V0"]]))
            np.save(os.path.join(out_dir, "hess.npy"), res.hessenberg_E0)
        if native:
            comm.detach(A)
            A.destroy()
    dist.barrier()
    dist.destroy_process_group()


def gpu_sharded_complex(rank, world, port, backend, out_dir):
    """A genuinely complex Hermitian operator (translation-symmetric sector, helpers.case('chain16_k3')) row-sharded
    from host arrays through qbh_csr_create_device: the exchange must stay complex (16 B per element)."""
    import ctypes as C

    import torch
    dist = _init(rank, world, port, backend)
    torch.cuda.set_device(0)
    import helpers
    import quantum_basis_amd as q
    from quantum_basis_amd import _lib, dist as qdist

    d, ia, ja, val, sym = helpers.case("chain16_k3")
    nblk, ranges = qdist.row_partition(d, world)
    r0, r1 = ranges[rank]
    dev = torch.device("cuda", 0)
    sia = torch.from_numpy((ia[r0:r1 + 1] - ia[r0]).astype(np.int64)).to(dev)
    sja = torch.from_numpy(ja[ia[r0]:ia[r1]].astype(np.int32)).to(dev)
    sval = torch.from_numpy(val[ia[r0]:ia[r1]].copy().view(np.float64)).to(dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        opts = q.make_opts(device=0, stream=stream.cuda_stream)
        h = C.c_void_p()
        _lib.check(_lib.lib().qbh_csr_create_device(C.byref(h), r1 - r0, d, r0, int(sia[-1].item()), sia.data_ptr(),
                                                     sja.data_ptr(), sval.data_ptr(), 0, C.byref(opts)), "qbh_csr_create_device")
        A = q.csr_mat(0, None, None, None, opts=opts, _handle=h)
        comm = qdist.ShardComm(d, rank=rank, world=world, device=dev, stream=stream).attach(A)
        res = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=600)
        nconv, w, _ = q.iram(A.dim, A, None, 2, 8, 300, "sr")
        assert not comm.errors, comm.errors
        assert comm.n_packed == 0, comm.n_packed           # complex operator: never the real wire format
        np.save(os.path.join(out_dir, "vec_%d.npy" % rank), res.eigenvecs)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array([res.E0, res.steps["E0"], res.steps["V0"], w[0], w[1]]))
    dist.barrier()
    dist.destroy_process_group()


def gpu_sharded_repr(rank, world, port, backend, out_dir):
    """Translation-symmetric sector generated shard by shard on the device (qbh_gen_heisenberg_repr with
    (shard, n_shards) = (rank, world)): triangular 4x4, k = (0, 1), complex phases, 22 zero-norm rows."""
    import torch
    dist = _init(rank, world, port, backend)
    torch.cuda.set_device(0)
    import quantum_basis_amd as q
    from quantum_basis_amd import dist as qdist, lattices

    perms, shifts = lattices.translations(4, 4)
    chars = lattices.characters(shifts, (0, 1), (4, 4))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        opts = q.make_opts(device=0, stream=stream.cuda_stream)
        A = q.csr_mat.heisenberg_repr(16, 8, lattices.triangular(4, 4), perms, chars, shard=(rank, world), opts=opts)
        d = int(A.info().ncols)
        assert A.info().row_offset == qdist.row_partition(d, world)[1][rank][0]
        comm = qdist.ShardComm(d, rank=rank, world=world, device=dev, stream=stream).attach(A)
        res = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=600)
        assert not comm.errors, comm.errors
        assert comm.n_packed == 0, comm.n_packed
        np.save(os.path.join(out_dir, "vec_%d.npy" % rank), res.eigenvecs)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array([res.E0, res.steps["E0"], res.steps["V0"], d]))
    dist.barrier()
    dist.destroy_process_group()


def gpu_sharded_hubbard_repr(rank, world, port, backend, out_dir, k=(1, 1)):
    """A Hubbard momentum sector generated shard by shard (qbh_gen_hubbard_repr with (shard, n_shards) = (rank, world)): the
    reference's 4x2 torus with 4+4 electrons; the local / remote split is made when the communicator is attached."""
    import torch
    dist = _init(rank, world, port, backend)
    torch.cuda.set_device(0)
    import quantum_basis_amd as q
    from quantum_basis_amd import dist as qdist, lattices

    perms, shifts = lattices.translations(4, 2)
    chars = lattices.characters(shifts, k, (4, 2))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        opts = q.make_opts(device=0, stream=stream.cuda_stream)
        A = q.csr_mat.hubbard_repr(8, 4, 4, lattices.square(4, 2), perms, chars, t=1.0, U=1.1, shard=(rank, world), opts=opts)
        d = int(A.info().ncols)
        assert A.info().row_offset == qdist.row_partition(d, world)[1][rank][0]
        comm = qdist.ShardComm(d, rank=rank, world=world, device=torch.device("cuda", 0), stream=stream).attach(A)
        res = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=600)
        assert not comm.errors, comm.errors
        np.save(os.path.join(out_dir, "vec_%d.npy" % rank), res.eigenvecs)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array([res.E0, res.steps["E0"], res.steps["V0"], d]))
    dist.barrier()
    dist.destroy_process_group()


def gpu_sharded_hostcsr(rank, world, port, backend, out_dir, native=False, case_name="hubbard_4x2"):
    """The unchanged host code's CSR (reference order, Hermitian-upper, int64) sharded over the ranks: every rank calls
    qbh_csr_create_rows on the SAME host arrays with its nnz-balanced (ragged) row block (qbh_balanced_row_cuts) and
    exchanges through either the native RCCL communicator (qbh_comm_create_rccl) or the gloo hooks with row_cuts."""
    import torch
    dist = _init(rank, world, port, backend)
    torch.cuda.set_device(0)
    import helpers
    import quantum_basis_amd as q
    from quantum_basis_amd import dist as qdist

    d, ia, ja, val, sym = helpers.case(case_name)
    cuts = q.balanced_row_cuts(d, ia, ja, sym, world)
    if world > 1:
        cuts = cuts.copy()
        cuts[1] += 3                                  # make sure the blocks are not the uniform ones
    r0, r1 = int(cuts[rank]), int(cuts[rank + 1])
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        opts = q.make_opts(device=0, stream=stream.cuda_stream)
        A = q.csr_mat(d, ia, ja, val, sym=sym, opts=opts, rows=(r0, r1))
        assert (A.info().row_offset, A.dim) == (r0, r1 - r0)
        if native:
            comm = qdist.NativeComm(d, rank=rank, world=world, cuts=cuts if world > 1 else None).attach(A)
        else:
            comm = qdist.ShardComm(d, rank=rank, world=world, device=torch.device("cuda", 0), stream=stream, cuts=cuts).attach(A)
        res = q.locate_E0_lanczos(A, nev=1, ncv=1, maxit=600)
        if not native:
            assert not comm.errors, comm.errors
        x = q.vec_randomize(A, seed=1)
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
        np.save(os.path.join(out_dir, "vec_%d.npy" % rank), res.eigenvecs)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array([res.E0, res.steps["E0"], res.steps["V0"]]))
            np.save(os.path.join(out_dir, "cuts.npy"), np.asarray(cuts))
        if native:
            comm.detach(A)
            A.destroy()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    pass
