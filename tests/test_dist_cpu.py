"""world_size-2 gloo tests of the N > 1 host logic (runs without a GPU): row partition,
exchange-buffer layout, the ShardComm hooks that libqbhip.so calls back into, and the sharded
recurrence built on them."""
import socket
import tempfile

import numpy as np
import pytest

import helpers
from oracle import qb_oracle as qo
from quantum_basis_amd import dist as qdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_from_cuts():
    nblk, ranges = qdist.partition_from_cuts([0, 5, 5, 12, 20])
    assert nblk == 8 and ranges == [(0, 5), (5, 5), (5, 12), (12, 20)]


def test_row_partition_covers_and_pads():
    for ncols, world in [(10, 1), (10, 3), (4900, 2), (4900, 8), (7, 8), (165636900, 8)]:
        nblk, ranges = qdist.row_partition(ncols, world)
        assert nblk * world >= ncols and nblk * (world - 1) < ncols + nblk
        assert ranges[0][0] == 0 and ranges[-1][1] == ncols
        for r, (a, b) in enumerate(ranges):
            assert a == min(r * nblk, ncols) and 0 <= b - a <= nblk
        assert sum(b - a for a, b in ranges) == ncols


def test_kron_row_cuts_are_whole_major_indices_and_even():
    """cuts of a product basis (index = major * S + minor) for shards that keep the Kronecker split: multiples of S, covering,
    as even as the major count allows (C3: 12870 majors over 8 ranks)"""
    for dim, S, world in [(4900, 70, 1), (4900, 70, 2), (4900, 70, 3), (853776, 924, 5), (165636900, 12870, 8), (165636900, 12870, 7)]:
        cuts = qdist.kron_row_cuts(dim, S, world)
        assert cuts[0] == 0 and cuts[-1] == dim and len(cuts) == world + 1 and np.all(cuts % S == 0)
        sizes = np.diff(cuts) // S
        assert sizes.min() >= 1 and sizes.max() - sizes.min() <= 1
        nblk, ranges = qdist.partition_from_cuts(cuts)
        assert nblk == sizes.max() * S and ranges[-1][1] == dim


def test_rebalance_cuts_keeps_every_shard_non_empty():
    """the strictly-increasing fix-up must hold from BOTH sides: a cost profile that piles everything into the last shard used
    to push interior cuts past dim (advisor, round 3)"""
    cuts = qdist.rebalance_cuts([0, 2, 4, 6, 8], [1e-9, 1e-9, 1e-9, 1.0])
    assert cuts[0] == 0 and cuts[-1] == 8 and np.all(np.diff(cuts) > 0)
    cuts = qdist.rebalance_cuts([0, 100, 200, 300], [3.0, 1.0, 1.0])
    assert cuts[0] == 0 and cuts[-1] == 300 and np.all(np.diff(cuts) > 0) and cuts[1] < 100
    with pytest.raises(AssertionError):
        qdist.rebalance_cuts([0, 1, 2, 3], [1.0, 1.0])          # wrong shape is refused loudly


@pytest.mark.parametrize("world,ragged", [(2, False), (3, False), (2, True), (3, True)])
def test_sharded_lanczos_with_gloo_hooks(world, ragged):
    import torch.multiprocessing as mp
    import dist_worker
    steps = 12
    name = "kagome_12"
    cuts = None
    if ragged:                 # nnz-balanced style partition: blocks of different lengths at their global offsets
        d = helpers.case(name)[0]
        cuts = [0, d // 3, d] if world == 2 else [0, d // 5, d // 5 + 7, d]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(dist_worker.cpu_sharded_lanczos, args=(world, _free_port(), name, steps, tmp, cuts), nprocs=world, join=True)
        ab = np.load(tmp + "/ab.npy")
    d, ia, ja, val, sym = helpers.case(name)
    O = qo.Csr(d, ia, ja, val, sym)
    maxit = 64
    v = np.zeros(2 * d, dtype=np.complex128)
    v[:d] = qo.vec_randomize(d, 1)
    hess = np.zeros(2 * maxit)
    qo.lanczos(0, steps, maxit, O, v, hess, "dnmcs")
    assert np.allclose(ab[0], hess[maxit:maxit + steps], rtol=1e-10)
    assert np.allclose(ab[1], hess[1:steps + 1], rtol=1e-10)
