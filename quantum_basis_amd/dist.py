"""Row sharding of the operator over ranks (one process per GPU) and the two exchange steps
the path has, implemented with torch.distributed (backend "nccl" == RCCL over xGMI on ROCm):

  * all-gather of the input vector x before every SpMV (each rank owns a contiguous row block)
  * all-reduce(sum) of <= 3 doubles after each fused reduction (Lanczos a_m, b_m; CG dots)

The reference has no distributed code (single process, OpenMP + MKL); the partition follows
SURVEY.md section 8(e).  libqbhip.so calls back into this module through the qbh_comm hooks
(include/qbhip.h), so the Lanczos / CG drivers are the same code for 1 and N GPUs.

torch is plumbing here: device buffers, stream ordering and the collectives.
"""
import ctypes as C
import traceback

import numpy as np

from . import _lib
from ._lib import check, lib


def row_partition(ncols, world):
    """Uniform contiguous row blocks: rank r owns [r*nblk, min((r+1)*nblk, ncols))."""
    nblk = (ncols + world - 1) // world
    return nblk, [(min(r * nblk, ncols), min((r + 1) * nblk, ncols)) for r in range(world)]


def partition_from_cuts(cuts):
    """(longest block, [(r0, r1)...]) of a ragged partition given by its global row cuts (qbh_balanced_row_cuts)."""
    cuts = [int(c) for c in cuts]
    ranges = list(zip(cuts[:-1], cuts[1:]))
    return max(b - a for a, b in ranges), ranges


class NativeComm:
    """The exchange steps on the library's own RCCL communicator (qbh_comm_create_rccl): no Python in the SpMV loop.
    torch.distributed is used once, to hand rank 0's ncclUniqueId to the other ranks."""

    def __init__(self, ncols, rank=None, world=None, cuts=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if (rank is None and dist.is_initialized()) else (rank or 0)
        self.world = dist.get_world_size(group) if (world is None and dist.is_initialized()) else (world or 1)
        self.cuts = None if cuts is None else [int(c) for c in cuts]
        if self.cuts is None:
            self.nblk, self.ranges = row_partition(ncols, self.world)
        else:
            self.nblk, self.ranges = partition_from_cuts(self.cuts)

    def attach(self, mat):
        import numpy as np
        uid = np.zeros(128, dtype=np.uint8)
        if self.rank == 0:
            check(lib().qbh_rccl_unique_id(uid.ctypes.data_as(C.c_void_p)), "qbh_rccl_unique_id")
        if self.world > 1:
            box = [uid.tobytes()]
            self.dist.broadcast_object_list(box, src=0, group=self.group)
            uid = np.frombuffer(box[0], dtype=np.uint8).copy()
        cuts = None
        if self.cuts is not None:
            cuts = np.asarray(self.cuts, dtype=np.int64)
        check(lib().qbh_comm_create_rccl(mat.handle, uid.ctypes.data_as(C.c_void_p), self.rank, self.world,
                                         None if cuts is None else cuts.ctypes.data_as(C.c_void_p)), "qbh_comm_create_rccl")
        mat._comm = self
        return self

    def detach(self, mat):
        check(lib().qbh_comm_destroy(mat.handle), "qbh_comm_destroy")


class ShardComm:
    """Owns the exchange buffers (torch tensors in HBM) and the hook callbacks.  cuts: global row cuts of a ragged
    (nnz-balanced) partition, or None for uniform blocks."""

    def __init__(self, ncols, rank=None, world=None, device=None, stream=None, group=None, cuts=None, parts=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.parts = parts           # offer the gather in parts (qbh_comm.allgather_part_begin) to a split shard
        self.n_parts_begun = 0
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.cuts = None if cuts is None else [int(c) for c in cuts]
        if self.cuts is None:
            self.nblk, self.ranges = row_partition(ncols, self.world)
            full = self.nblk * self.world
        else:
            self.nblk, self.ranges = partition_from_cuts(self.cuts)
            full = int(ncols)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stream = stream
        f64 = torch.float64
        # complex128 stored as interleaved float64 pairs (NCCL has no complex dtype)
        self.xsend = torch.zeros(2 * self.nblk, dtype=f64, device=self.device)
        self.xfull = torch.zeros(2 * full, dtype=f64, device=self.device)
        self.scal = torch.zeros(16, dtype=f64, device=self.device)
        # real wire format: when a solve is real only the real parts travel (nblk doubles per rank)
        self.xfull_r = torch.zeros(full, dtype=f64, device=self.device)
        # ragged partition: equal-sized slots are gathered into a scratch buffer and copied to their global offsets
        self.scratch = torch.zeros(2 * self.nblk * self.world, dtype=f64, device=self.device) if self.cuts is not None else None
        self.real_wire = True
        self.n_packed = 0
        backend = dist.get_backend(group)
        self.direct = (backend == "nccl") or self.device.type == "cpu"
        self.errors = []
        self._ag = _lib.ALLGATHER_FN(self._allgather)
        self._ar = _lib.ALLREDUCE_FN(self._allreduce)
        self._ag_begin = _lib.ALLGATHER_FN(self._allgather_begin)
        self._ag_wait = _lib.ALLWAIT_FN(self._allgather_wait)
        self._pt_begin = _lib.PART_BEGIN_FN(self._part_begin)
        self._pt_wait = _lib.PART_WAIT_FN(self._part_wait)
        self._pt_begin_w = _lib.PART_BEGIN_W_FN(self._part_begin_w)
        self._work = None
        self.n_async = 0             # exchanges started through the begin/wait pair
        self.overlap = True          # hand the begin/wait pair to the library (local columns overlap the gather)
        self._struct = None

    # -- hooks: called from inside libqbhip.so on the calling Python thread ------------------
    def _ctx(self):
        if self.stream is not None:
            return self.torch.cuda.stream(self.stream)
        import contextlib
        return contextlib.nullcontext()

    def _bufs(self, packed):
        """(recv, send) of one exchange: complex128 as float64 pairs, or real parts only."""
        if packed:
            self.n_packed += 1
            return self.xfull_r, self.xsend[:self.nblk]
        return self.xfull, self.xsend

    def _place_ragged(self, recv, w):
        """scratch holds world slots of nblk*w doubles; copy block q to its global offset cuts[q]*w in recv"""
        slot = self.nblk * w
        for q, (a, b) in enumerate(self.ranges):
            if b > a:
                recv[a * w:b * w].copy_(self.scratch[q * slot:q * slot + (b - a) * w])

    def _allgather(self, _ctx, packed):
        try:
            recv, send = self._bufs(packed)
            w = 1 if packed else 2
            with self._ctx():
                target = recv if self.cuts is None else self.scratch[:self.nblk * self.world * w]
                if self.direct:
                    self.dist.all_gather_into_tensor(target, send, group=self.group)
                else:   # gloo with device tensors (single-GPU test rigs): stage through the host
                    self.torch.cuda.current_stream().synchronize()
                    hsend = send.cpu()
                    parts = [self.torch.empty_like(hsend) for _ in range(self.world)]
                    self.dist.all_gather(parts, hsend, group=self.group)
                    target.copy_(self.torch.cat(parts))
                if self.cuts is not None:
                    self._place_ragged(recv, w)
            return 0
        except Exception:            # never let an exception unwind through the C frames
            self.errors.append(traceback.format_exc())
            return 1

    def _allgather_begin(self, _ctx, packed):
        """Enqueue the exchange and return: RCCL runs it on its own stream, ordered after what the
        operator's stream has enqueued so far (the copy into xsend)."""
        try:
            if not self.direct or self.cuts is not None:   # host-staged test rigs / ragged cuts: no real overlap here
                return self._allgather(_ctx, packed)
            recv, send = self._bufs(packed)
            with self._ctx():
                self._work = self.dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
            self.n_async += 1
            return 0
        except Exception:
            self.errors.append(traceback.format_exc())
            return 1

    def _allgather_wait(self, _ctx):
        """Order the operator's stream after the exchange started by _allgather_begin."""
        try:
            if self._work is not None:
                with self._ctx():
                    self._work.wait()
                self._work = None
            return 0
        except Exception:
            self.errors.append(traceback.format_exc())
            return 1

    def _part_begin(self, _ctx, part, nparts, off_len):
        return self._part_begin_w(_ctx, part, nparts, off_len, 0)

    def _part_begin_w(self, _ctx, part, nparts, off_len, packed):
        """One part of the gather in parts: elements [off, off + len) of every rank's block, one broadcast per rank, on the
        operator's stream (no overlap on this rig: what it exercises is the library's part geometry across real ranks).
        packed: the elements are doubles (real parts only, qbh_opts.real_wire), from xsend[:nblk] into xfull_r."""
        try:
            if packed:
                self.n_packed += 1
                w, full, send = 1, self.xfull_r, self.xsend
            else:
                w, full, send = 2, self.xfull, self.xsend
            with self._ctx():
                for q, (a, _b) in enumerate(self.ranges):
                    off, ln = int(off_len[2 * q]), int(off_len[2 * q + 1])
                    if ln <= 0:
                        continue
                    base = a if self.cuts is not None else q * self.nblk
                    dst = full[w * (base + off):w * (base + off + ln)]
                    if q == self.rank:
                        dst.copy_(send[w * off:w * (off + ln)])
                    src = q if self.group is None else self.dist.get_global_rank(self.group, q)
                    if self.direct:
                        self.dist.broadcast(dst, src=src, group=self.group)
                    else:
                        self.torch.cuda.current_stream().synchronize()
                        h = dst.cpu()
                        self.dist.broadcast(h, src=src, group=self.group)
                        dst.copy_(h)
            self.n_parts_begun += 1
            return 0
        except Exception:
            self.errors.append(traceback.format_exc())
            return 1

    def _part_wait(self, _ctx, part):
        return 0                     # everything above ran on (or was ordered with) the operator's stream

    def _allreduce(self, _ctx, off, n):
        try:
            with self._ctx():
                view = self.scal[off:off + n]
                if self.direct:
                    self.dist.all_reduce(view, group=self.group)
                else:
                    self.torch.cuda.current_stream().synchronize()
                    h = view.cpu()
                    self.dist.all_reduce(h, group=self.group)
                    view.copy_(h)
            return 0
        except Exception:
            self.errors.append(traceback.format_exc())
            return 1

    # -----------------------------------------------------------------------------------------
    def attach(self, mat):
        """Install the hooks on a row-shard operator (qbh_csr_set_comm)."""
        r0, r1 = self.ranges[self.rank]
        # The collectives below are enqueued on torch's current stream; the library launches on the operator's stream.
        # They must be the SAME stream (include/qbhip.h: "enqueued on / ordered with the operator's stream"), or the
        # pack -> gather -> SpMV and reduce -> all-reduce sequences race silently.
        if self.device.type == "cuda":
            op_stream = mat.info().stream or 0
            if self.stream is None:
                self.stream = self.torch.cuda.ExternalStream(op_stream, device=self.device)
            elif int(self.stream.cuda_stream) != int(op_stream):
                raise ValueError("ShardComm stream %#x is not the operator's stream %#x: create both with the same stream"
                                 % (int(self.stream.cuda_stream), int(op_stream)))
        if mat.row_offset != r0 or mat.dim != r1 - r0:
            raise ValueError("operator rows [%d,%d) do not match rank %d's block [%d,%d)"
                             % (mat.row_offset, mat.row_offset + mat.dim, self.rank, r0, r1))
        c = _lib.Comm()
        c.rank, c.nranks, c.nblk = self.rank, self.world, self.nblk
        c.d_xsend = self.xsend.data_ptr()
        c.d_xfull = self.xfull.data_ptr()
        c.d_scal = self.scal.data_ptr()
        c.d_xfull_r = self.xfull_r.data_ptr() if self.real_wire else None
        c.ctx = None
        c.allgather_x = self._ag
        c.allreduce_sum = self._ar
        if self.cuts is not None:
            import numpy as np
            self._cuts_arr = np.asarray(self.cuts, dtype=np.int64)
            c.row_cuts = self._cuts_arr.ctypes.data
        if self.overlap:
            c.allgather_begin = self._ag_begin
            c.allgather_wait = self._ag_wait
        if self.parts:
            c.allgather_part_begin = self._pt_begin
            c.allgather_part_wait = self._pt_wait
            c.allgather_part_begin_w = self._pt_begin_w
        self._struct = c
        check(lib().qbh_csr_set_comm(mat.handle, C.byref(c)), "qbh_csr_set_comm")
        mat._comm = self          # keep the callbacks and buffers alive as long as the operator
        return self


def rebalance_cuts(cuts, cost):
    """Row cuts that equalise a measured per-shard cost (e.g. ms per SpMV of each rank on the current cuts), assuming the
    cost density is constant inside every current shard: cuts [0, ..., dim] (world + 1 entries), cost one number per shard.
    Uniform row blocks of a momentum sector are unbalanced in work although their nonzero counts agree to a few per cent
    (BASELINE configs[3], 4 ranks: 63 to 98 ms), so the balancing quantity is the measured time, not nnz.  One or two
    rounds (generate, time, rebalance, generate with row_cuts=) bring the ranks together."""
    cuts = np.asarray(cuts, dtype=np.int64)
    cost = np.asarray(cost, dtype=np.float64)
    world = cost.size
    assert cuts.size == world + 1 and np.all(np.diff(cuts) > 0) and np.all(cost > 0)
    cum = np.concatenate([[0.0], np.cumsum(cost)])             # cumulative cost at the current cuts
    want = cum[-1] * np.arange(1, world) / world
    new = np.interp(want, cum, cuts.astype(np.float64))        # piecewise-linear inverse of the cumulative cost
    out = np.concatenate([[cuts[0]], np.round(new).astype(np.int64), [cuts[-1]]])
    for q in range(1, world):                                  # strictly increasing, every shard keeps at least one row:
        out[q] = max(out[q], out[q - 1] + 1)                   # ... pushed up from the left
    for q in range(world - 1, 0, -1):
        out[q] = min(out[q], out[q + 1] - 1)                   # ... and held back from the right (the last cut stays at dim)
    assert np.all(np.diff(out) > 0), "fewer rows than shards"
    return out


def kron_row_cuts(dim, minor, world):
    """Row cuts at WHOLE MAJOR INDICES of a product basis (index = major * minor + minor_index) for `world` row shards, as
    even as the major count allows: what a shard of an operator with a Kronecker split (qbh_opts.kron_split) needs to keep its
    two-part form and to exchange the tiled copy of its block under a communicator (cuts not at multiples of `minor` fall back to
    the plain CSR shard)."""
    assert minor > 0 and dim % minor == 0 and dim // minor >= world
    nu = dim // minor
    return np.asarray([(q * nu) // world * minor for q in range(world + 1)], dtype=np.int64)
