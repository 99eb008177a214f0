// qbh_commattach.cpp -- qbh_csr_set_comm: attaching a communicator is COLLECTIVE; the ranks agree on errors, on the form of the
// exchange (tiled blocks of split shards or plain blocks) and on the number of gather parts before anything is decided.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "qbh_api_priv.hpp"

using qbh::d2;
using namespace qbhapi;

namespace qbhapi {
// The gather in parts (qbh_comm::allgather_part_begin): the far pass sweeps the gathered x band range by band range, a band
// range is one contiguous piece of every rank's tiled block, so the far pass of the first range can run while the later ranges
// are still on the links -- the step then costs max(wire, near + far) instead of max(wire, near) + far.  Default: 4 parts when
// there are ranks to receive from (QBH_GATHER_PARTS overrides; 1 = the single gather), none when the communicator has no
// part hooks (the Python ShardComm) or the far part is not sliced.
// what THIS rank could do (1 = the single gather); the ranks then take the smallest proposal (qbh_csr_set_comm): a rank that
// issued one whole-block group while its peers issue four part groups would hang the exchange
int kron_parts_wanted(const qbh_csr *A, const qbh_comm *comm)
{
    const qbh_csr::KronSplit &K = A->kron;
    if (!K.active || !K.sliced || !comm->allgather_part_begin || !comm->allgather_part_wait || !A->d_wctr || K.nwb_f <= 0) return 1;
    int64_t want = comm->nranks > 1 ? 4 : 1;
    if (A->opts.gather_parts > 0) want = A->opts.gather_parts;
    const int64_t nfb = K.t.S / K.t.B;                       // full bands (the far pass covers exactly these)
    return (int)std::max<int64_t>(1, std::min<int64_t>({want, 8, nfb}));
}

int kron_gather_parts(qbh_csr *A, const qbh_comm *comm, int64_t want)
{
    qbh_csr::KronSplit &K = A->kron;
    K.n_parts = 1;
    K.part_off_len.clear();
    const int64_t nfb = K.t.S / K.t.B;
    if (want <= 1 || kron_parts_wanted(A, comm) < want) return QBH_OK;
    int64_t band[9];
    for (int64_t k = 0; k <= want; ++k) band[k] = k * nfb / want;
    K.part_blk[0] = 0;
    for (int64_t k = 0; k <= want; ++k) K.part_band[k] = band[k];
    for (int64_t k = 1; k < want; ++k) {                     // first slot of the range's first group -> the block that holds it
        int64_t slot = 0;
        QBH_HIP(hipMemcpy(&slot, K.ia_f + band[k] * K.t.NU, sizeof(int64_t), hipMemcpyDeviceToHost));
        K.part_blk[k] = std::min<int64_t>(slot / 512, K.nwb_f);
    }
    K.part_blk[want] = K.nwb_f;
    K.part_off_len.assign((size_t)want * 2 * (size_t)comm->nranks, 0);
    for (int64_t k = 0; k < want; ++k)
        for (int q = 0; q < comm->nranks; ++q) {
            const int64_t nu = K.rank_cu[q + 1] - K.rank_cu[q];
            const int64_t off = band[k] * K.t.B * nu;
            const int64_t end = k == want - 1 ? nu * K.t.S : band[k + 1] * K.t.B * nu;
            K.part_off_len[((size_t)k * (size_t)comm->nranks + (size_t)q) * 2] = off;
            K.part_off_len[((size_t)k * (size_t)comm->nranks + (size_t)q) * 2 + 1] = end - off;
        }
    K.n_parts = (int)want;
    return QBH_OK;
}

// Which major indices of its peers does this shard read?  The far part of a shard gathers, for each of its own major indices,
// the hop targets of that up configuration: a subset of the gathered x (C3, 8 ranks in the generator's order: 42-68 %,
// tools/needed_columns.py).  Only those are moved into the tiled x (k_kron_place with a list): the rest of every block
// arrives and is never touched.  need_frac = what a sparse exchange would still have to carry.
int kron_needed_majors(qbh_csr *A, int nranks, std::vector<uint8_t> *bits_out)
{
    qbh_csr::KronSplit &K = A->kron;
    if (K.d_need) (void)hipFree(K.d_need);
    K.d_need = nullptr;
    K.need_frac = 1.0;
    const int64_t NUg = K.NUg;
    uint8_t *d_bits = nullptr;
    QBH_HIP(qbh::dev_alloc(&d_bits, (size_t)NUg));
    std::vector<uint8_t> bits((size_t)NUg, 0);
    int rc = QBH_OK;
    hipError_t e = hipMemsetAsync(d_bits, 0, (size_t)NUg, A->stream);
    if (e == hipSuccess) rc = qbh::launch_kron_need(K.c16_f, K.c16_f ? nullptr : K.ja_f, K.far_slots, K.ja_x, K.nnz_x, K.t.S, NUg, K.t.B, d_bits, A->stream);
    if (e == hipSuccess && rc == QBH_OK) e = hipMemcpyAsync(bits.data(), d_bits, (size_t)NUg, hipMemcpyDeviceToHost, A->stream);
    if (e == hipSuccess && rc == QBH_OK) e = hipStreamSynchronize(A->stream);
    (void)hipFree(d_bits);
    if (rc != QBH_OK) return rc;
    if (e != hipSuccess) {
        qbh::set_error("kron_needed_majors: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return QBH_EHIP;
    }
    std::vector<int32_t> list;
    int64_t peers_all = 0, peers_need = 0;
    K.need_lo[0] = 0;
    for (int q = 0; q < nranks; ++q) {
        for (int64_t u = K.rank_cu[q]; u < K.rank_cu[q + 1]; ++u)
            if (bits[(size_t)u]) list.push_back((int32_t)(u - K.rank_cu[q]));
        K.need_lo[q + 1] = (int64_t)list.size();
        if (K.rank_cu[q] != K.U0) {
            peers_all += K.rank_cu[q + 1] - K.rank_cu[q];
            peers_need += K.need_lo[q + 1] - K.need_lo[q];
        }
    }
    if (peers_all > 0) K.need_frac = (double)peers_need / (double)peers_all;
    if (list.empty()) list.push_back(0);
    QBH_HIP(qbh::dev_alloc(&K.d_need, list.size() * sizeof(int32_t)));
    QBH_HIP(hipMemcpy(K.d_need, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (bits_out) bits_out->swap(bits);
    return QBH_OK;
}

// Personalised exchange (qbh_opts.sparse_gather): every rank tells every other which major indices it reads -- the need bitmaps
// travel ONCE through the communicator's own all-gather (in front of the rank's send block) -- and keeps, per destination, the
// list of its own major indices that destination reads.  Collective (every rank of a communicator that agreed on it calls it).
int kron_sparse_setup(qbh_csr *A, const qbh_comm *comm, const std::vector<uint8_t> &bits)
{
    qbh_csr::KronSplit &K = A->kron;
    const int np = comm->nranks;
    const int64_t NUg = K.NUg, S = K.t.S, NUq = K.t.NU;
    if ((int64_t)bits.size() != NUg || NUg > comm->nblk * (int64_t)sizeof(qbh::d2)) return QBH_EINVAL;
    auto base = [&](int q) { return comm->row_cuts ? comm->row_cuts[q] : (int64_t)q * comm->nblk; };
    QBH_HIP(hipMemcpyAsync(comm->d_xsend, bits.data(), (size_t)NUg, hipMemcpyHostToDevice, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    if (comm->allgather_x(comm->ctx, 0) != 0) {
        qbh::set_error("qbh_csr_set_comm: allgather hook failed (need bitmaps)");
        return QBH_ECOMM;
    }
    std::vector<int32_t> list;
    std::vector<uint8_t> theirs((size_t)NUg);
    K.send_lo[0] = 0;
    for (int p = 0; p < np; ++p) {
        if (p != comm->rank) {
            QBH_HIP(hipMemcpyAsync(theirs.data(), reinterpret_cast<const char *>(comm->d_xfull) + (size_t)base(p) * sizeof(qbh::d2), (size_t)NUg,
                                   hipMemcpyDeviceToHost, A->stream));
            QBH_HIP(hipStreamSynchronize(A->stream));
            for (int64_t ul = 0; ul < NUq; ++ul)
                if (theirs[(size_t)(K.U0 + ul)]) list.push_back((int32_t)ul);
        }
        K.send_lo[p + 1] = (int64_t)list.size();
    }
    // what this rank receives: its own need list, peers only (kron_needed_majors)
    int64_t recv_majors = 0;
    for (int q = 0; q < np; ++q)
        if (q != comm->rank) recv_majors += K.need_lo[q + 1] - K.need_lo[q];
    const int64_t send_elems = std::max<int64_t>(1, (int64_t)list.size() * S), recv_elems = std::max<int64_t>(1, recv_majors * S);
    if (list.empty()) list.push_back(0);
    for (void **q : {(void **)&K.d_send_list, (void **)&K.d_vsend, (void **)&K.d_vrecv}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    QBH_HIP(qbh::dev_alloc(&K.d_send_list, list.size() * sizeof(int32_t)));
    QBH_HIP(hipMemcpy(K.d_send_list, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    QBH_HIP(qbh::dev_alloc(&K.d_vsend, (size_t)send_elems * sizeof(qbh::d2)));
    QBH_HIP(qbh::dev_alloc(&K.d_vrecv, (size_t)recv_elems * sizeof(qbh::d2)));
    K.vsend_cap = send_elems;
    K.vrecv_cap = recv_elems;
    K.sparse = true;
    return QBH_OK;
}

}  // namespace qbhapi

extern "C" int qbh_csr_set_comm(qbh_csr *A, const qbh_comm *comm)
{
    if (!A) return QBH_EINVAL;
    if (!comm || comm->nranks < 1) {          // NULL detaches; a 1-rank communicator is valid (hooks still run)
        A->has_comm = false;
        if (A->kron.active && A->kron.comm_tiled) {       // (the far columns never left the tiled order of the whole vector)
            A->kron.comm_tiled = false;
            A->kron.n_ranks = 1;
            A->kron.n_parts = 1;
            A->kron.xt_of = nullptr;
            if (A->kron.d_need) (void)hipFree(A->kron.d_need);
            A->kron.d_need = nullptr;
            for (void **q : {(void **)&A->kron.d_send_list, (void **)&A->kron.d_vsend, (void **)&A->kron.d_vrecv}) {
                if (*q) (void)hipFree(*q);
                *q = nullptr;
            }
            A->kron.sparse = false;
        }
        return QBH_OK;
    }
    if (A->kind == 3) {
        qbh::set_error("qbh_csr_set_comm: the matrix-free sector operator is a single-GPU form");
        return QBH_EUNSUPP;
    }
    if (!comm->d_xsend || !comm->d_xfull || !comm->d_scal || !comm->allgather_x || !comm->allreduce_sum ||
        comm->rank < 0 || comm->rank >= comm->nranks || comm->nblk < A->nrows) {
        // without buffers and hooks there is nothing to tell the peers with: the one failure that stays local
        qbh::set_error("qbh_csr_set_comm: incomplete communicator (rank %d/%d nblk %lld nrows %lld)", comm->rank, comm->nranks,
                       (long long)comm->nblk, (long long)A->nrows);
        return QBH_EINVAL;
    }
    // This call is COLLECTIVE for nranks > 1: every rank's local verdict travels through the communicator's own all-reduce
    // before anything is decided, so that no rank returns early while its peers wait in a collective, and the form of the
    // exchange (tiled blocks or plain, how many parts) is the same everywhere by construction.
    int local_err = QBH_OK;
    std::vector<int64_t> cuts_new;
    int64_t full_new = 0;
    if (comm->row_cuts) {
        const int64_t *c = comm->row_cuts;
        bool ok = c[0] == 0 && c[comm->nranks] == A->ncols && c[comm->rank] == A->row_offset &&
                  c[comm->rank + 1] - c[comm->rank] == A->nrows;
        for (int q = 0; q < comm->nranks && ok; ++q) ok = c[q + 1] >= c[q] && c[q + 1] - c[q] <= comm->nblk;
        if (!ok) {
            qbh::set_error("qbh_csr_set_comm: row_cuts do not describe this shard (rank %d/%d rows [%lld, %lld))", comm->rank,
                           comm->nranks, (long long)A->row_offset, (long long)(A->row_offset + A->nrows));
            local_err = QBH_EINVAL;
        } else {
            cuts_new.assign(c, c + comm->nranks + 1);
            full_new = A->ncols;
        }
    } else {
        if (comm->nblk * comm->rank != A->row_offset || comm->nblk * comm->nranks < A->ncols) {
            qbh::set_error("qbh_csr_set_comm: inconsistent communicator (rank %d/%d nblk %lld row_offset %lld)",
                           comm->rank, comm->nranks, (long long)comm->nblk, (long long)A->row_offset);
            local_err = QBH_EINVAL;
        }
        full_new = comm->nblk * (int64_t)comm->nranks;
    }
    Bind bind(A);
    // sums of indicators over the ranks: [0] failures, [1] ranks that can exchange tiled blocks, [2 + k] ranks proposing k + 1 parts
    auto agree = [&](double (&v)[12]) -> int {
        if (comm->nranks == 1) return QBH_OK;
        QBH_HIP(hipMemcpyAsync(comm->d_scal, v, sizeof(v), hipMemcpyHostToDevice, A->stream));
        if (comm->allreduce_sum(comm->ctx, 0, 12) != 0) {
            qbh::set_error("qbh_csr_set_comm: allreduce_sum hook failed");
            return QBH_ECOMM;
        }
        QBH_HIP(hipMemcpyAsync(v, comm->d_scal, sizeof(v), hipMemcpyDeviceToHost, A->stream));
        QBH_HIP(hipStreamSynchronize(A->stream));
        return QBH_OK;
    };
    qbh_csr::KronSplit &K = A->kron;
    const int64_t S = K.active ? K.t.S : 1;
    bool mine = local_err == QBH_OK && A->kind == 0 && K.active && K.map.nc == 1 && comm->nranks <= qbh::kKronMaxRanks;
    if (mine) {
        if (comm->row_cuts) {
            for (int q = 0; q <= comm->nranks; ++q) mine = mine && comm->row_cuts[q] % S == 0;
        } else {
            mine = comm->nblk % S == 0;
        }
    }
    const int my_parts = mine ? kron_parts_wanted(A, comm) : 1;
    double v[12] = {0};
    v[0] = local_err != QBH_OK ? 1.0 : 0.0;
    v[1] = mine ? 1.0 : 0.0;
    v[2 + (my_parts - 1)] = 1.0;
    v[10] = (mine && A->opts.sparse_gather != 0 && comm->exchange_v != nullptr && comm->allgather_part_wait != nullptr) ? 1.0 : 0.0;
    QBH_TRY(agree(v));
    if (v[0] > 0.0) {
        if (local_err == QBH_OK) qbh::set_error("qbh_csr_set_comm: a peer rank rejected the communicator (its qbh_last_error says why)");
        return local_err != QBH_OK ? local_err : QBH_ECOMM;
    }
    A->comm_cuts = cuts_new;
    A->comm_full = full_new;
    const bool all_tiled = v[1] == (double)comm->nranks;
    if (qbh::debug_sw().trace_create)
        fprintf(stderr, "[qbh_csr_set_comm] rank %d/%d: split %d, can exchange tiled %d (S %lld, cuts %s), proposes %d parts; agreed: failures %.0f, tiled %.0f of %d\n",
                comm->rank, comm->nranks, (int)K.active, (int)mine, (long long)S, comm->row_cuts ? "given" : "uniform", my_parts, v[0], v[1], comm->nranks);
    int parts = 1;
    for (int k = 0; k < 8; ++k)
        if (v[2 + k] > 0.0) {
            parts = k + 1;                       // the smallest proposal
            break;
        }
    if (A->kind == 0) {
        // A shard split in place (kron_build) exchanges the TILED copy of its block -- which only works when every rank does:
        // cuts at whole major indices and every operator split.  Without agreement a split shard is merged back into its CSR
        // and takes the generic path below; whether that worked is agreed on once more (a rank out of memory there must not
        // leave its peers attached and waiting in their first gather).
        if (K.active && all_tiled) {
            // Every rank sends the tiled copy of its own block; the far (and cross) columns of every shard keep indexing the tiled
            // order of the WHOLE vector -- 2 bytes each, relative to the block's band, exactly the one-GPU operator's -- and the
            // gathered blocks are moved to their place in the handle's tiled x (k_kron_place) piece by piece as they arrive.
            // Nothing of the operator changes when a communicator comes or goes.
            K.n_ranks = comm->nranks;
            for (int q = 0; q <= comm->nranks; ++q) {
                const int64_t cut = comm->row_cuts ? comm->row_cuts[q] : std::min<int64_t>((int64_t)q * comm->nblk, A->ncols);
                K.rank_cu[q] = cut / S;
            }
            K.comm_tiled = true;
            K.xt_of = nullptr;
            // the rest of this branch can fail on one rank only (a copy inside kron_gather_parts): agreed on once more, so that
            // no rank is left attached and waiting in its first gather while a peer has returned an error
            int prc = kron_gather_parts(A, comm, parts);
            std::vector<uint8_t> bits;
            if (prc == QBH_OK) prc = kron_needed_majors(A, comm->nranks, &bits);
            K.sparse = false;
            const bool all_sparse = comm->nranks > 1 && v[10] == (double)comm->nranks;
            if (all_sparse) {
                // the bitmap exchange is a collective of its own: every rank enters it, a rank whose preparation failed with a
                // bitmap of zeros (its verdict travels in the agreement below)
                if (prc != QBH_OK) bits.assign((size_t)K.NUg, 0);
                const int src = kron_sparse_setup(A, comm, bits);
                if (prc == QBH_OK) prc = src;
            }
            double w[12] = {0};
            w[0] = prc != QBH_OK ? 1.0 : 0.0;
            const int arc = agree(w);
            if (prc != QBH_OK || arc != QBH_OK || w[0] > 0.0) {
                K.comm_tiled = false;
                K.sparse = false;
                K.n_ranks = 1;
                K.n_parts = 1;
                if (prc == QBH_OK && arc == QBH_OK) qbh::set_error("qbh_csr_set_comm: a peer rank could not set up the gather in parts");
                return prc != QBH_OK ? prc : arc != QBH_OK ? arc : QBH_ECOMM;
            }
        } else if (!all_tiled) {
            int rrc = QBH_OK;
            if (K.active) {
                rrc = kron_restore(A);
                if (rrc == QBH_OK) rrc = build_geometry(A);
            }
            double w[12] = {0};
            w[0] = rrc != QBH_OK ? 1.0 : 0.0;
            const int arc = agree(w);
            if (rrc != QBH_OK || arc != QBH_OK || w[0] > 0.0) {
                if (rrc == QBH_OK && arc == QBH_OK) qbh::set_error("qbh_csr_set_comm: a peer rank could not merge its split operator back into a CSR");
                return rrc != QBH_OK ? rrc : arc != QBH_OK ? arc : QBH_ECOMM;
            }
        }
    }
    if (A->kind == 0 && !A->kron.active && !A->has_rem && A->nrows < A->ncols) {      // first communicator on a stored row shard: split it now
        // the split comes FIRST and the communicator is committed only when it succeeded: a failure of the split itself (out
        // of memory) leaves the operator exactly as it was, unattached, with its single-part geometry; a failure AFTER it
        // (geometry of the two parts) leaves a handle that refuses every further SpMV
        Bind bind(A);
        QBH_HIP(hipStreamSynchronize(A->stream));
        QBH_TRY(split_shard(A));
        if (A->has_rem) {
            const int rc = build_geometry(A);
            if (rc != QBH_OK) {                     // the shard IS split but has no geometry: nothing can run on it any more
                A->has_comm = false;
                A->broken = true;
                return rc;
            }
        }
        QBH_HIP(hipStreamSynchronize(A->stream));
    }
    A->comm = *comm;
    A->comm.row_cuts = A->comm_cuts.empty() ? nullptr : A->comm_cuts.data();
    A->has_comm = true;
    return QBH_OK;
}

