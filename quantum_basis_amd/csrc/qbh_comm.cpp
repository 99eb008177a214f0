// qbh_comm.cpp -- the native communicator: the two exchange steps of the row-sharded path on RCCL over
// xGMI, implemented in C++ behind the same qbh_comm hooks the library already drives, so that a C++ host -- which is
// what the reference is (src/model.cc:1177-1181 calls lanczos() from one host thread) -- can run N > 1 ranks without
// any Python in the SpMV loop:
//   all-gather of x      ncclAllGather for uniform row blocks; for nnz-balanced (ragged) cuts an all-gather-v made of
//                        grouped point-to-point calls: every rank posts one ncclSend of its block and one ncclRecv
//                        of the peer's block per peer inside ONE group, so each xGMI link carries exactly one peer
//                        block in each direction (round 2 issued P grouped ncclBroadcasts here, which RCCL runs one
//                        after the other over its own rings).  Runs on a side stream; begin() / wait() bracket it
//                        with events so the locally-owned columns of a split shard are applied while the links are busy.
//   all-reduce(sum)      <= 16 doubles (Lanczos a_m / b_m, CG dots, the real-wire flags), same side stream.
// librccl is resolved at run time (dlopen), so libqbhip.so still loads where RCCL is absent; then
// qbh_comm_create_rccl fails loudly with QBH_EUNSUPP.  Types come from the real <rccl/rccl.h>.
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "qbh_internal.hpp"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

std::string g_rccl_error;          // why the loader failed (written once, under the call_once below)

void load_rccl(RcclApi &api)
{
    // QBH_RCCL_LIB names the library outright -- test rigs put tests/stub_rccl/librccl_stub.so there to run N ranks on one
    // GPU, which real RCCL refuses; it is loaded RTLD_LOCAL so that its nccl* symbols never shadow a real librccl in the process
    if (const char *path = getenv("QBH_RCCL_LIB")) {
        if (*path) {
            api.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
            if (!api.handle) {
                const char *e = dlerror();
                g_rccl_error = std::string("QBH_RCCL_LIB=") + path + ": " + (e ? e : "cannot be loaded");
                return;
            }
        }
    }
    // a copy that is already mapped (e.g. the one torch bundles) is reused: same soname
    if (!api.handle)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
            const char *e = dlerror();          // dlerror() clears the message: read it exactly once per failure
            g_rccl_error = e ? e : "no such library";
        }
    if (!api.handle) return;
#define QBH_SYM(field, sym)                                                   \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym)); \
    if (!api.field) {                                                         \
        g_rccl_error = std::string("symbol missing: ") + sym;                 \
        api.handle = nullptr;                                                 \
        return;                                                               \
    }
    QBH_SYM(GetUniqueId, "ncclGetUniqueId")
    QBH_SYM(CommInitRank, "ncclCommInitRank")
    QBH_SYM(CommDestroy, "ncclCommDestroy")
    QBH_SYM(GetErrorString, "ncclGetErrorString")
    QBH_SYM(AllGather, "ncclAllGather")
    QBH_SYM(AllReduce, "ncclAllReduce")
    QBH_SYM(Send, "ncclSend")
    QBH_SYM(Recv, "ncclRecv")
    QBH_SYM(GroupStart, "ncclGroupStart")
    QBH_SYM(GroupEnd, "ncclGroupEnd")
#undef QBH_SYM
}

RcclApi *rccl()
{
    static RcclApi api;
    static std::once_flag once;             // two host threads may create their first communicator at the same time
    std::call_once(once, [] { load_rccl(api); });
    return api.handle ? &api : nullptr;
}

}  // namespace

// everything the hooks need; owned by the operator handle (qbh_csr::native)
struct qbh_native_comm {
    RcclApi    *api = nullptr;
    ncclComm_t  comm = nullptr;
    int         rank = 0, nranks = 1;
    bool        ragged = false;
    std::vector<int64_t> cuts;        // [nranks+1]
    int64_t     nblk = 0;             // longest block (size of d_xsend)
    hipStream_t op = nullptr;         // the operator's stream
    hipStream_t side = nullptr;       // RCCL's stream
    hipEvent_t  ready = nullptr, done = nullptr;
    hipEvent_t  part_done[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // the gather in parts
    double     *d_xsend = nullptr, *d_xfull = nullptr, *d_xfull_r = nullptr, *d_scal = nullptr;
    bool        in_flight = false;
    char        err[256] = "";
    // event timing of the gather on the side stream (harvested at the next gather / by qbh_get_stats)
    qbh_csr    *owner = nullptr;
    hipEvent_t  t0 = nullptr, t1 = nullptr;
    bool        timing_pending = false;
};

namespace {

int fail(qbh_native_comm *c, const char *what, ncclResult_t r)
{
    snprintf(c->err, sizeof(c->err), "%s: %s", what, c->api->GetErrorString(r));
    qbh::set_error("RCCL %s", c->err);
    return 1;
}

void harvest_gather_time(qbh_native_comm *c)
{
    if (!c->timing_pending) return;
    c->timing_pending = false;
    float ms = 0.f;
    if (hipEventSynchronize(c->t1) == hipSuccess && hipEventElapsedTime(&ms, c->t0, c->t1) == hipSuccess && c->owner)
        c->owner->stats.ms_gather += ms;
}

// the exchange itself, enqueued on the side stream after everything the operator's stream has enqueued so far
int enqueue_gather(qbh_native_comm *c, int packed)
{
    harvest_gather_time(c);                     // the previous exchange finished long ago: no stall
    if (hipEventRecord(c->ready, c->op) != hipSuccess || hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess) return 1;
    const bool timed = c->owner && c->owner->opts.profile != 0;
    if (timed && hipEventRecord(c->t0, c->side) != hipSuccess) return 1;
    const size_t w = packed ? 1 : 2;                               // doubles per element on the wire
    double *recv = packed ? c->d_xfull_r : c->d_xfull;
    ncclResult_t r;
    if (!c->ragged) {
        r = c->api->AllGather(c->d_xsend, recv, (size_t)c->nblk * w, ncclDouble, c->comm, c->side);
        if (r != ncclSuccess) return fail(c, "ncclAllGather", r);
    } else {
        // all-gather-v: one send + one receive per peer in a single group; the own block is a device copy
        const size_t mine = (size_t)(c->cuts[(size_t)c->rank + 1] - c->cuts[(size_t)c->rank]);
        if ((r = c->api->GroupStart()) != ncclSuccess) return fail(c, "ncclGroupStart", r);
        for (int q = 0; q < c->nranks; ++q) {
            if (q == c->rank) continue;
            const size_t len = (size_t)(c->cuts[(size_t)q + 1] - c->cuts[(size_t)q]);
            if (len > 0) {
                r = c->api->Recv(recv + (size_t)c->cuts[(size_t)q] * w, len * w, ncclDouble, q, c->comm, c->side);
                if (r != ncclSuccess) {
                    (void)c->api->GroupEnd();
                    return fail(c, "ncclRecv", r);
                }
            }
            if (mine > 0) {
                r = c->api->Send(c->d_xsend, mine * w, ncclDouble, q, c->comm, c->side);
                if (r != ncclSuccess) {
                    (void)c->api->GroupEnd();
                    return fail(c, "ncclSend", r);
                }
            }
        }
        if ((r = c->api->GroupEnd()) != ncclSuccess) return fail(c, "ncclGroupEnd", r);
        if (mine > 0 && hipMemcpyAsync(recv + (size_t)c->cuts[(size_t)c->rank] * w, c->d_xsend, mine * w * sizeof(double),
                                       hipMemcpyDeviceToDevice, c->side) != hipSuccess) {
            qbh::set_error("qbh_comm: device copy of the rank's own block failed: %s", hipGetErrorString(hipGetLastError()));
            return 1;
        }
    }
    if (timed) {
        if (hipEventRecord(c->t1, c->side) != hipSuccess) return 1;
        c->timing_pending = true;
    }
    if (hipEventRecord(c->done, c->side) != hipSuccess) return 1;
    c->in_flight = true;
    return 0;
}

int hook_wait(void *ctx)
{
    auto *c = static_cast<qbh_native_comm *>(ctx);
    if (!c->in_flight) return 0;
    c->in_flight = false;
    return hipStreamWaitEvent(c->op, c->done, 0) == hipSuccess ? 0 : 1;
}

int hook_begin(void *ctx, int packed) { return enqueue_gather(static_cast<qbh_native_comm *>(ctx), packed); }

int hook_gather(void *ctx, int packed)
{
    if (enqueue_gather(static_cast<qbh_native_comm *>(ctx), packed) != 0) return 1;
    return hook_wait(ctx);
}

// The gather in parts (qbh_comm::allgather_part_begin): part `part` of `nparts` = elements [off, off + len) of every rank's
// block, one send + one receive per peer in a single group (each xGMI link carries one piece per direction), the own piece a
// device copy; all on the side stream, in the order of the calls.  An event per part lets the operator's stream wait for one.
int hook_part_begin_w(void *ctx, int part, int nparts, const int64_t *off_len, int packed)
{
    auto *c = static_cast<qbh_native_comm *>(ctx);
    const size_t w = packed ? 1 : 2;                               // doubles per element on the wire
    double *recv = packed ? c->d_xfull_r : c->d_xfull;
    if (part < 0 || part >= 8 || nparts > 8) return 1;
    const bool timed = c->owner && c->owner->opts.profile != 0;
    if (part == 0) {
        harvest_gather_time(c);
        if (hipEventRecord(c->ready, c->op) != hipSuccess || hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess) return 1;
        if (timed && hipEventRecord(c->t0, c->side) != hipSuccess) return 1;
    }
    if (!c->part_done[part] && hipEventCreateWithFlags(&c->part_done[part], hipEventDisableTiming) != hipSuccess) return 1;
    auto base = [&](int q) { return c->ragged ? c->cuts[(size_t)q] : (int64_t)q * c->nblk; };
    ncclResult_t r;
    const int64_t my_off = off_len[2 * c->rank], my_len = off_len[2 * c->rank + 1];
    if (c->nranks > 1) {
        if ((r = c->api->GroupStart()) != ncclSuccess) return fail(c, "ncclGroupStart", r);
        for (int q = 0; q < c->nranks; ++q) {
            if (q == c->rank) continue;
            const int64_t off = off_len[2 * q], len = off_len[2 * q + 1];
            if (len > 0) {
                r = c->api->Recv(recv + (size_t)(base(q) + off) * w, (size_t)len * w, ncclDouble, q, c->comm, c->side);
                if (r != ncclSuccess) {
                    (void)c->api->GroupEnd();
                    return fail(c, "ncclRecv", r);
                }
            }
            if (my_len > 0) {
                r = c->api->Send(c->d_xsend + (size_t)my_off * w, (size_t)my_len * w, ncclDouble, q, c->comm, c->side);
                if (r != ncclSuccess) {
                    (void)c->api->GroupEnd();
                    return fail(c, "ncclSend", r);
                }
            }
        }
        if ((r = c->api->GroupEnd()) != ncclSuccess) return fail(c, "ncclGroupEnd", r);
    }
    if (my_len > 0 && hipMemcpyAsync(recv + (size_t)(base(c->rank) + my_off) * w, c->d_xsend + (size_t)my_off * w, (size_t)my_len * w * sizeof(double),
                                     hipMemcpyDeviceToDevice, c->side) != hipSuccess) {
        qbh::set_error("qbh_comm: device copy of the rank's own piece failed: %s", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    if (hipEventRecord(c->part_done[part], c->side) != hipSuccess) return 1;
    if (part == nparts - 1 && timed) {
        if (hipEventRecord(c->t1, c->side) != hipSuccess) return 1;
        c->timing_pending = true;
    }
    return 0;
}

int hook_part_begin(void *ctx, int part, int nparts, const int64_t *off_len) { return hook_part_begin_w(ctx, part, nparts, off_len, 0); }

// The personalised exchange in parts (qbh_comm::exchange_v): one send + one receive per peer in a single group, each with its
// own offset and length in the library's packed buffers; zero lengths are skipped (both sides know them: the lists were agreed
// at attach time).  Same stream, events and timing as the gather in parts.
int hook_exchange_v(void *ctx, int part, int nparts, const int64_t *send_off_len, const int64_t *recv_off_len, int elem_doubles, const void *d_send,
                    void *d_recv)
{
    auto *c = static_cast<qbh_native_comm *>(ctx);
    if (part < 0 || part >= 8 || nparts > 8 || (elem_doubles != 1 && elem_doubles != 2)) return 1;
    const bool timed = c->owner && c->owner->opts.profile != 0;
    if (part == 0) {
        harvest_gather_time(c);
        if (hipEventRecord(c->ready, c->op) != hipSuccess || hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess) return 1;
        if (timed && hipEventRecord(c->t0, c->side) != hipSuccess) return 1;
    }
    if (!c->part_done[part] && hipEventCreateWithFlags(&c->part_done[part], hipEventDisableTiming) != hipSuccess) return 1;
    const size_t w = (size_t)elem_doubles;
    const double *sb = static_cast<const double *>(d_send);
    double *rb = static_cast<double *>(d_recv);
    ncclResult_t r;
    if (c->nranks > 1) {
        if ((r = c->api->GroupStart()) != ncclSuccess) return fail(c, "ncclGroupStart", r);
        for (int q = 0; q < c->nranks; ++q) {
            if (q == c->rank) continue;
            const int64_t roff = recv_off_len[2 * q], rlen = recv_off_len[2 * q + 1], soff = send_off_len[2 * q], slen = send_off_len[2 * q + 1];
            if (rlen > 0) {
                r = c->api->Recv(rb + (size_t)roff * w, (size_t)rlen * w, ncclDouble, q, c->comm, c->side);
                if (r != ncclSuccess) {
                    (void)c->api->GroupEnd();
                    return fail(c, "ncclRecv", r);
                }
            }
            if (slen > 0) {
                r = c->api->Send(sb + (size_t)soff * w, (size_t)slen * w, ncclDouble, q, c->comm, c->side);
                if (r != ncclSuccess) {
                    (void)c->api->GroupEnd();
                    return fail(c, "ncclSend", r);
                }
            }
        }
        if ((r = c->api->GroupEnd()) != ncclSuccess) return fail(c, "ncclGroupEnd", r);
    }
    if (hipEventRecord(c->part_done[part], c->side) != hipSuccess) return 1;
    if (part == nparts - 1 && timed) {
        if (hipEventRecord(c->t1, c->side) != hipSuccess) return 1;
        c->timing_pending = true;
    }
    return 0;
}

int hook_part_wait(void *ctx, int part)
{
    auto *c = static_cast<qbh_native_comm *>(ctx);
    if (part < 0 || part >= 8 || !c->part_done[part]) return 1;
    return hipStreamWaitEvent(c->op, c->part_done[part], 0) == hipSuccess ? 0 : 1;
}

int hook_allreduce(void *ctx, int off, int n)
{
    auto *c = static_cast<qbh_native_comm *>(ctx);
    if (hipEventRecord(c->ready, c->op) != hipSuccess || hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess) return 1;
    ncclResult_t r = c->api->AllReduce(c->d_scal + off, c->d_scal + off, (size_t)n, ncclDouble, ncclSum, c->comm, c->side);
    if (r != ncclSuccess) return fail(c, "ncclAllReduce", r);
    if (hipEventRecord(c->done, c->side) != hipSuccess || hipStreamWaitEvent(c->op, c->done, 0) != hipSuccess) return 1;
    return 0;
}

void destroy_native(qbh_native_comm *c)
{
    if (!c) return;
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->comm && c->api) (void)c->api->CommDestroy(c->comm);
    if (c->d_xsend) (void)hipFree(c->d_xsend);
    if (c->d_xfull) (void)hipFree(c->d_xfull);
    if (c->d_xfull_r) (void)hipFree(c->d_xfull_r);
    if (c->d_scal) (void)hipFree(c->d_scal);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    for (hipEvent_t e : c->part_done)
        if (e) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
}

}  // namespace

void qbh::harvest_native_comm(qbh_csr *A)
{
    if (A && A->native) harvest_gather_time(A->native);
}

void qbh::release_native_comm(qbh_csr *A)
{
    if (A && A->native) {
        destroy_native(A->native);
        A->native = nullptr;
    }
}

extern "C" int qbh_rccl_unique_id(void *uid128)
{
    if (!uid128) return QBH_EINVAL;
    RcclApi *api = rccl();
    if (!api) {
        qbh::set_error("librccl could not be loaded (%s)", g_rccl_error.c_str());
        return QBH_EUNSUPP;
    }
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) {
        qbh::set_error("ncclGetUniqueId: %s", api->GetErrorString(r));
        return QBH_ECOMM;
    }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(uid128, &id, sizeof(id));
    return QBH_OK;
}

extern "C" int qbh_comm_create_rccl(qbh_csr *A, const void *uid128, int rank, int nranks, const int64_t *row_cuts)
{
    if (!A || !uid128 || nranks < 1 || rank < 0 || rank >= nranks) {
        qbh::set_error("qbh_comm_create_rccl: invalid argument");
        return QBH_EINVAL;
    }
    RcclApi *api = rccl();
    if (!api) {
        qbh::set_error("librccl could not be loaded: the native communicator is unavailable");
        return QBH_EUNSUPP;
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != A->device) QBH_HIP(hipSetDevice(A->device));
    qbh::release_native_comm(A);
    qbh_native_comm *c = new (std::nothrow) qbh_native_comm();
    if (!c) return QBH_ENOMEM;
    c->api = api;
    c->rank = rank;
    c->nranks = nranks;
    c->op = A->stream;
    c->cuts.resize((size_t)nranks + 1);
    const int64_t uni = (A->ncols + nranks - 1) / nranks;
    c->ragged = false;
    for (int q = 0; q <= nranks; ++q) {
        const int64_t u = std::min<int64_t>((int64_t)q * uni, A->ncols);
        c->cuts[(size_t)q] = row_cuts ? row_cuts[q] : u;
        if (c->cuts[(size_t)q] != u) c->ragged = true;
    }
    if (qbh::debug_sw().force_ragged) c->ragged = true;         // test rigs: the send/recv all-gather-v even for uniform cuts
    c->nblk = uni;
    for (int q = 0; q < nranks; ++q) c->nblk = std::max(c->nblk, c->cuts[(size_t)q + 1] - c->cuts[(size_t)q]);
    int rc = QBH_OK;
    auto bail = [&](int code) {
        destroy_native(c);
        if (prev >= 0 && prev != A->device) (void)hipSetDevice(prev);
        return code;
    };
    if (c->cuts[0] != 0 || c->cuts[(size_t)nranks] != A->ncols || c->cuts[(size_t)rank] != A->row_offset ||
        c->cuts[(size_t)rank + 1] - c->cuts[(size_t)rank] != A->nrows) {
        qbh::set_error("qbh_comm_create_rccl: rank %d owns rows [%lld, %lld) but the cuts say [%lld, %lld) of %lld", rank,
                       (long long)A->row_offset, (long long)(A->row_offset + A->nrows), (long long)c->cuts[(size_t)rank],
                       (long long)c->cuts[(size_t)rank + 1], (long long)A->ncols);
        return bail(QBH_EINVAL);
    }
    // uniform blocks are gathered with one ncclAllGather into nranks * nblk slots; ragged cuts land at their global offsets
    const size_t full = c->ragged ? (size_t)A->ncols : (size_t)c->nblk * (size_t)nranks;
#define QBH_C(call)                                                                       \
    if ((call) != hipSuccess) {                                                           \
        qbh::set_error("%s failed (qbh_comm_create_rccl)", #call);                        \
        return bail(QBH_EHIP);                                                            \
    }
    // RCCL's stream takes the HIGHEST priority the device offers.  The two passes of a split shard are persistent launches that fill
    // every CU with as many wavefronts as their registers allow; a send / receive kernel that becomes runnable at the same moment as
    // the near pass (both wait for the pack kernel) must be placed FIRST, and the kernel of the next band range must get the CUs the
    // previous one frees -- otherwise the exchange would start when the near pass ends and nothing would be hidden.  (The stand-in of
    // the test rigs models the exchange as a delay on this stream and cannot see this; a node can.)
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
            (void)hipGetLastError();
            least = greatest = 0;
        }
        QBH_C(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, qbh::debug_sw().side_noprio ? least : greatest));
    }
    QBH_C(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    QBH_C(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    QBH_C(hipEventCreate(&c->t0));
    QBH_C(hipEventCreate(&c->t1));
    c->owner = A;
    QBH_C(qbh::dev_alloc(&c->d_xsend, (size_t)c->nblk * 2 * sizeof(double)));
    QBH_C(qbh::dev_alloc(&c->d_xfull, full * 2 * sizeof(double)));
    QBH_C(qbh::dev_alloc(&c->d_xfull_r, full * sizeof(double)));
    QBH_C(qbh::dev_alloc(&c->d_scal, 16 * sizeof(double)));
    // On the OPERATOR's stream, and waited for: hipMemset runs on the null stream and may return before it has executed, and a host's
    // stream (torch creates non-blocking ones) is not ordered with the null stream -- the zeroing could land AFTER the first data the
    // operator's stream puts into these buffers.  Found by the first 2-rank run through the librccl stand-in (round 5): the agreement
    // all-reduce of qbh_csr_set_comm lost one rank's contribution now and then, and both ranks fell back to the plain exchange.
    QBH_C(hipMemsetAsync(c->d_xsend, 0, (size_t)c->nblk * 2 * sizeof(double), c->op));
    QBH_C(hipMemsetAsync(c->d_xfull, 0, full * 2 * sizeof(double), c->op));
    QBH_C(hipMemsetAsync(c->d_xfull_r, 0, full * sizeof(double), c->op));
    QBH_C(hipMemsetAsync(c->d_scal, 0, 16 * sizeof(double), c->op));
    QBH_C(hipStreamSynchronize(c->op));
#undef QBH_C
    ncclUniqueId id;
    std::memcpy(&id, uid128, sizeof(id));
    ncclResult_t r = api->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        qbh::set_error("ncclCommInitRank: %s", api->GetErrorString(r));
        c->comm = nullptr;
        return bail(QBH_ECOMM);
    }
    qbh_comm h{};
    h.rank = rank;
    h.nranks = nranks;
    h.nblk = c->nblk;
    h.d_xsend = reinterpret_cast<qbh_z *>(c->d_xsend);
    h.d_xfull = reinterpret_cast<qbh_z *>(c->d_xfull);
    h.d_scal = c->d_scal;
    h.d_xfull_r = c->d_xfull_r;
    h.ctx = c;
    h.allgather_x = hook_gather;
    h.allreduce_sum = hook_allreduce;
    h.allgather_begin = hook_begin;
    h.allgather_wait = hook_wait;
    h.allgather_part_begin = hook_part_begin;
    h.allgather_part_wait = hook_part_wait;
    h.allgather_part_begin_w = hook_part_begin_w;
    h.exchange_v = hook_exchange_v;
    h.row_cuts = c->ragged ? c->cuts.data() : nullptr;
    rc = qbh_csr_set_comm(A, &h);
    if (rc != QBH_OK) return bail(rc);
    A->native = c;
    if (prev >= 0 && prev != A->device) (void)hipSetDevice(prev);
    return QBH_OK;
}

extern "C" int qbh_comm_destroy(qbh_csr *A)
{
    if (!A) return QBH_EINVAL;
    (void)qbh_csr_set_comm(A, nullptr);
    qbh::release_native_comm(A);
    return QBH_OK;
}
