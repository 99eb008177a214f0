// rccl_stub.cpp -- TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl that lets N rank processes which
// share ONE GPU (or sit on several) run the library's native communicator (quantum_basis_amd/csrc/qbh_comm.cpp:
// qbh_comm_create_rccl, the grouped ncclSend / ncclRecv all-gather-v, the gather in parts, the all-reduce of the Lanczos
// scalars) before a multi-GPU node ever does.  Real RCCL refuses two ranks on one device; this stub does not.
// Selected by the environment variable QBH_RCCL_LIB=<path to librccl_stub.so> (qbh_comm.cpp load_rccl).
//
// Semantics kept from RCCL: every call is ordered after the work already enqueued on its stream and its result is visible
// to work enqueued afterwards; point-to-point calls inside ncclGroupStart / ncclGroupEnd are matched per (source,
// destination) pair in the order they were posted; element counts of a matched send / receive must agree (real RCCL hangs
// or corrupts -- the stub FAILS, which is what a test wants); collectives must be called by all ranks in the same order.
// SOLO mode (QBH_STUB_SOLO=<GB/s per link>): ONE rank process of an N-rank job runs alone and its peers are MODELLED -- nothing
// synchronises the device: a group of receives is a host function on the stream that holds it for (longest message) / link rate
// + QBH_STUB_LATENCY_US (default 20; every peer has a link of its own) -- the receive buffers keep what they held (zeros: the
// peers' blocks of x are zero, every number stays finite, the results mean nothing; QBH_STUB_SOLO_COPY=1 fills them with the
// rank's own block by hipMemcpyAsync, i.e. adds the HBM writes the arriving data would cause) -- and an
// all-reduce returns nranks times the rank's own contribution (so the collective agreements of qbh_csr_set_comm come out as
// they would with real peers) after the latency.  A rehearsal of the TIMING path -- events, side stream, what the near pass
// hides of the gather -- on one GPU.
// Mechanics: every call synchronises its stream, stages device memory through files mapped by all ranks (under $TMPDIR --
// a container's /dev/shm is often 64 MB) and meets the other ranks at a sense-reversing barrier in a mapped control block.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {

constexpr int kMaxRanks = 16;
constexpr int kMaxMsgs = 64;

struct Ctl {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> failed;                 // a rank saw a mismatch: everybody returns an error from the current call
    std::atomic<int> attached;
    std::atomic<long long> box_bytes[kMaxRanks];     // current size of every rank's outbox file
};

struct MsgHdr {
    int dst;                                 // -1: for everybody (collectives)
    long long off, bytes;
};

struct BoxHdr {
    int n_msgs;
    MsgHdr msg[kMaxMsgs];
};

struct Pending {
    bool send;
    int peer;
    const void *src;
    void *dst;
    size_t bytes;
    hipStream_t stream;
};

struct Box {
    int fd = -1;
    char *map = nullptr;
    long long mapped = 0;
};

}  // namespace

struct ncclComm {
    bool solo = false, solo_copy = false;
    double link_gbps = 50.0, latency_us = 20.0;
    double *h_red = nullptr;                 // pinned staging of the solo all-reduce (a ring of 64 slots of 16 doubles)
    long long red_slot = 0;
    int rank = 0, nranks = 1;
    std::string base;
    Ctl *ctl = nullptr;
    Box box[kMaxRanks];
    int sense = 0;
    long long n_calls = 0;
};

namespace {

thread_local int g_group_depth = 0;
thread_local std::vector<Pending> g_pending;
thread_local ncclComm *g_group_comm = nullptr;
thread_local char g_err[256] = "";

// without a GPU (the build container) the buffers are host memory: the stub's own matching logic is testable there
bool host_mode()
{
    static const bool host = [] {
        int n = 0;
        const bool none = hipGetDeviceCount(&n) != hipSuccess || n == 0;
        (void)hipGetLastError();
        return none;
    }();
    return host;
}
hipError_t copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (host_mode()) {
        std::memcpy(dst, src, bytes);
        return hipSuccess;
    }
    return hipMemcpy(dst, src, bytes, kind);
}
hipError_t sync(hipStream_t s) { return host_mode() ? hipSuccess : hipStreamSynchronize(s); }

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

ncclResult_t fail(const char *fmt, long long a = 0, long long b = 0, long long c = 0)
{
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    fprintf(stderr, "[rccl stub] %s\n", g_err);
    return ncclInvalidUsage;
}

bool barrier(ncclComm *c)
{
    // sense-reversing; a rank that died leaves the others here: give up after 120 s instead of hanging the test box
    const int gen = c->ctl->generation.load();
    if (c->ctl->arrived.fetch_add(1) + 1 == c->nranks) {
        c->ctl->arrived.store(0);
        c->ctl->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (c->ctl->generation.load() == gen) {
        if (++spins > 2000) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
        }
    }
    return true;
}

bool map_box(ncclComm *c, int q, long long want)
{
    Box &b = c->box[q];
    if (b.fd < 0) {
        const std::string path = c->base + "." + std::to_string(q);
        b.fd = open(path.c_str(), O_RDWR | O_CREAT, 0600);
        if (b.fd < 0) return false;
    }
    if (b.mapped >= want) return true;
    if (b.map) munmap(b.map, (size_t)b.mapped);
    b.map = nullptr;
    if (q == c->rank) {
        if (ftruncate(b.fd, (off_t)want) != 0) return false;
        c->ctl->box_bytes[q].store(want);
    }
    void *p = mmap(nullptr, (size_t)want, PROT_READ | PROT_WRITE, MAP_SHARED, b.fd, 0);
    if (p == MAP_FAILED) return false;
    b.map = static_cast<char *>(p);
    b.mapped = want;
    return true;
}

// the rank's own outbox with room for `payload` bytes behind the header
char *own_box(ncclComm *c, long long payload)
{
    long long want = (long long)sizeof(BoxHdr) + payload;
    want = ((want + (1 << 20) - 1) >> 20) << 20;
    if (!map_box(c, c->rank, std::max(want, c->box[c->rank].mapped))) return nullptr;
    return c->box[c->rank].map;
}

const char *peer_box(ncclComm *c, int q)
{
    const long long sz = c->ctl->box_bytes[q].load();
    if (sz <= 0 || !map_box(c, q, sz)) return nullptr;
    return c->box[q].map;
}

// ---- solo mode ----
struct Hold {
    double us;
};
void hold_stream(void *p)
{
    Hold *h = static_cast<Hold *>(p);
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < h->us) { }
    delete h;
}
// QBH_STUB_SOLO_KERNEL=W[:ldsKB]: the hold is a KERNEL of W workgroups (256 threads, ldsKB of LDS each) that spin for the modelled
// time from the moment they START -- like RCCL's send / receive kernels it needs CU resources, so it competes with the persistent
// passes of a split shard for a place on the chip: what the near pass hides of the exchange then depends on who is dispatched
// first (the priority of the communicator's side stream, qbh_comm.cpp), which the host-function hold cannot show.
__global__ __launch_bounds__(256) void k_stub_occupy(unsigned long long ticks)
{
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) lds[0] = 1;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// QBH_STUB_SOLO_KERNEL=W:rccl -- the footprint of RCCL's own kernel on gfx950 (rcclGenericKernel in the librccl this image ships:
// 256 threads, 261-280 registers per lane of which 17-32 AGPRs, 19.7 KB of LDS; llvm-readelf --notes of the unbundled code object):
// the clobbers make the compiler allocate that many registers
__global__ __launch_bounds__(256) void k_stub_occupy_rccl(unsigned long long ticks)
{
    __shared__ volatile char lds[19744];
    asm volatile("" ::: "v255", "a23");
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x * 77] = 1;
    while (wall_clock64() - t0 < ticks && lds[threadIdx.x * 77] == 1) __builtin_amdgcn_s_sleep(32);
}
bool g_occ_rccl = false;
int g_occ_wg = -1, g_occ_lds_kb = 0;
double g_wall_khz = 0.0;
bool occupy_mode()
{
    if (g_occ_wg < 0) {
        g_occ_wg = 0;
        if (const char *e = getenv("QBH_STUB_SOLO_KERNEL")) {
            g_occ_wg = atoi(e);
            if (const char *c = strchr(e, ':')) {
                g_occ_rccl = strcmp(c + 1, "rccl") == 0;
                g_occ_lds_kb = g_occ_rccl ? 0 : atoi(c + 1);
            }
            int dev = 0, khz = 0;
            if (g_occ_wg > 0 && (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0 ||
                                 hipFuncSetAttribute(reinterpret_cast<const void *>(k_stub_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, g_occ_lds_kb * 1024) != hipSuccess)) {
                fprintf(stderr, "rccl_stub: QBH_STUB_SOLO_KERNEL cannot be set up (%s): host-function hold\n", hipGetErrorString(hipGetLastError()));
                g_occ_wg = 0;
            }
            g_wall_khz = (double)khz;
        }
    }
    return g_occ_wg > 0;
}
ncclResult_t solo_hold(hipStream_t s, double us)
{
    if (us <= 0.0) return ncclSuccess;
    if (occupy_mode()) {
        if (g_occ_rccl) hipLaunchKernelGGL(k_stub_occupy_rccl, dim3((unsigned)g_occ_wg), dim3(256), 0, s, (unsigned long long)(us * g_wall_khz * 1e-3));
        else hipLaunchKernelGGL(k_stub_occupy, dim3((unsigned)g_occ_wg), dim3(256), (size_t)g_occ_lds_kb * 1024, s, (unsigned long long)(us * g_wall_khz * 1e-3));
        if (hipGetLastError() != hipSuccess) return fail("k_stub_occupy launch failed (solo mode)");
        return ncclSuccess;
    }
    if (hipLaunchHostFunc(s, hold_stream, new Hold{us}) != hipSuccess) return fail("hipLaunchHostFunc failed (solo mode)");
    return ncclSuccess;
}
ncclResult_t solo_exchange(ncclComm *c, std::vector<Pending> &ops)
{
    const void *src = nullptr;
    size_t src_bytes = 0, longest = 0;
    hipStream_t s = nullptr;
    for (const Pending &p : ops) {
        s = p.stream;
        if (p.send && p.bytes > src_bytes) {
            src = p.src;
            src_bytes = p.bytes;
        }
    }
    for (const Pending &p : ops) {
        if (p.send || p.bytes == 0) continue;
        longest = std::max(longest, p.bytes);
        // QBH_STUB_SOLO_COPY: as many bytes as the peer would have delivered, from the rank's own block (repeated when the peer's block is longer)
        for (size_t done = 0; c->solo_copy && done < p.bytes && src_bytes > 0;) {
            const size_t n = std::min(src_bytes, p.bytes - done);
            if (hipMemcpyAsync(static_cast<char *>(p.dst) + done, src, n, hipMemcpyDeviceToDevice, p.stream) != hipSuccess) return fail("hipMemcpyAsync failed (solo mode)");
            done += n;
        }
    }
    ++c->n_calls;
    return solo_hold(s, c->latency_us + (double)longest / (c->link_gbps * 1e3));
}

// one exchange round: the posted sends go into the outbox, everybody meets, the posted receives are served from the peers'
// outboxes (k-th receive from q <- k-th message of q addressed to this rank), everybody meets again
ncclResult_t exchange(ncclComm *c, std::vector<Pending> &ops)
{
    if (c->solo) return solo_exchange(c, ops);
    std::vector<hipStream_t> streams;
    for (const Pending &p : ops) {
        bool seen = false;
        for (hipStream_t s : streams) seen = seen || s == p.stream;
        if (!seen) streams.push_back(p.stream);
    }
    for (hipStream_t s : streams)
        if (sync(s) != hipSuccess) return fail("hipStreamSynchronize failed before an exchange");
    long long payload = 0;
    int n_send = 0;
    for (const Pending &p : ops)
        if (p.send) {
            payload += (long long)((p.bytes + 63) / 64) * 64;
            ++n_send;
        }
    if (n_send > kMaxMsgs) return fail("more than %lld sends in one group", kMaxMsgs);
    char *mine = own_box(c, payload);
    if (!mine) return fail("cannot map the outbox (%lld bytes)", payload);
    BoxHdr *h = reinterpret_cast<BoxHdr *>(mine);
    h->n_msgs = 0;
    long long off = sizeof(BoxHdr);
    bool ok = true;
    for (const Pending &p : ops)
        if (p.send) {
            MsgHdr &m = h->msg[h->n_msgs++];
            m.dst = p.peer;
            m.off = off;
            m.bytes = (long long)p.bytes;
            if (p.bytes > 0 && copy(mine + off, p.src, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
            off += (long long)((p.bytes + 63) / 64) * 64;
        }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (!ok) c->ctl->failed.store(1);
    if (!barrier(c)) return fail("a rank did not reach the exchange (rank %lld of %lld, call %lld)", c->rank, c->nranks, c->n_calls);
    int taken[kMaxRanks] = {0};
    for (const Pending &p : ops)
        if (!p.send) {
            const char *pb = peer_box(c, p.peer);
            if (!pb) {
                ok = false;
                fail("cannot map the outbox of rank %lld", p.peer);
                break;
            }
            const BoxHdr *ph = reinterpret_cast<const BoxHdr *>(pb);
            int k = -1, seen = 0;
            for (int i = 0; i < ph->n_msgs; ++i)
                if (ph->msg[i].dst == c->rank || ph->msg[i].dst == -1) {
                    if (seen == taken[p.peer]) {
                        k = i;
                        break;
                    }
                    ++seen;
                }
            if (k < 0) {
                ok = false;
                fail("rank %lld posted a receive from rank %lld that no send matches (real RCCL would hang)", c->rank, p.peer);
                continue;
            }
            ++taken[p.peer];
            if (ph->msg[k].bytes != (long long)p.bytes) {
                ok = false;
                fail("receive of %lld bytes from rank %lld meets a send of %lld bytes (real RCCL would hang or corrupt)", (long long)p.bytes, p.peer,
                     ph->msg[k].bytes);
                continue;
            }
            if (p.bytes > 0 && copy(p.dst, pb + ph->msg[k].off, p.bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false;
        }
    // sends addressed to this rank that it never asked for are a mismatch too
    for (int q = 0; q < c->nranks && ok; ++q) {
        if (q == c->rank) continue;
        const char *pb = peer_box(c, q);
        if (!pb) continue;
        const BoxHdr *ph = reinterpret_cast<const BoxHdr *>(pb);
        int addressed = 0;
        for (int i = 0; i < ph->n_msgs; ++i) addressed += ph->msg[i].dst == c->rank ? 1 : 0;
        if (addressed > taken[q]) {
            ok = false;
            fail("rank %lld sent %lld messages to rank %lld which posted fewer receives", q, addressed, c->rank);
        }
    }
    if (!ok) c->ctl->failed.store(1);
    if (!barrier(c)) return fail("a rank did not finish the exchange (rank %lld of %lld)", c->rank, c->nranks);
    ++c->n_calls;
    return c->ctl->failed.load() ? ncclInvalidUsage : ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof(*id));
    const char *tmp = getenv("TMPDIR");
    const unsigned long long stamp = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof(id->internal), "%s/qbh_rccl_stub_%d_%llx", tmp && *tmp ? tmp : "/tmp", (int)getpid(), stamp);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return fail("ncclCommInitRank: bad arguments");
    ncclComm *c = new ncclComm();
    c->rank = rank;
    c->nranks = nranks;
    c->base = std::string(id.internal, strnlen(id.internal, sizeof(id.internal)));
    if (const char *solo = getenv("QBH_STUB_SOLO")) {
        if (*solo && atof(solo) > 0.0) {
            c->solo = true;
            c->link_gbps = atof(solo);
            if (const char *lat = getenv("QBH_STUB_LATENCY_US")) c->latency_us = atof(lat);
            if (const char *cp = getenv("QBH_STUB_SOLO_COPY")) c->solo_copy = atoi(cp) != 0;
            if (hipHostMalloc(&c->h_red, 64 * 16 * sizeof(double)) != hipSuccess) {
                delete c;
                return fail("hipHostMalloc failed (solo mode)");
            }
            *out = c;
            return ncclSuccess;
        }
    }
    const std::string ctl_path = c->base + ".ctl";
    const int fd = open(ctl_path.c_str(), O_RDWR | O_CREAT, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Ctl)) != 0) {
        delete c;
        return fail("cannot create the control file");
    }
    void *p = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return fail("cannot map the control file");
    }
    c->ctl = static_cast<Ctl *>(p);          // a fresh file is all zeros: counters start at 0
    if (!own_box(c, 0)) {
        delete c;
        return fail("cannot create the outbox");
    }
    c->ctl->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->attached.load() < nranks) {          // the real call is collective as well
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
            delete c;
            return fail("ncclCommInitRank: only %lld of %lld ranks arrived", c->ctl->attached.load(), nranks);
        }
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    if (c->solo) {
        if (c->h_red) (void)hipHostFree(c->h_red);
        delete c;
        return ncclSuccess;
    }
    for (int q = 0; q < kMaxRanks; ++q) {
        if (c->box[q].map) munmap(c->box[q].map, (size_t)c->box[q].mapped);
        if (c->box[q].fd >= 0) close(c->box[q].fd);
    }
    unlink((c->base + "." + std::to_string(c->rank)).c_str());
    if (c->ctl) {
        if (c->ctl->attached.fetch_sub(1) == 1) unlink((c->base + ".ctl").c_str());
        munmap(c->ctl, sizeof(Ctl));
    }
    delete c;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (g_err[0] ? g_err : "rccl stub error"); }

ncclResult_t ncclGroupStart()
{
    ++g_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_group_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
    if (--g_group_depth > 0) return ncclSuccess;
    ncclComm *c = g_group_comm;
    g_group_comm = nullptr;
    std::vector<Pending> ops;
    ops.swap(g_pending);
    if (!c) return ncclSuccess;              // an empty group
    return exchange(c, ops);
}

static ncclResult_t post(ncclComm *c, const Pending &p)
{
    if (g_group_depth > 0) {
        if (g_group_comm && g_group_comm != c) return fail("one communicator per group in this stub");
        g_group_comm = c;
        g_pending.push_back(p);
        return ncclSuccess;
    }
    std::vector<Pending> ops{p};
    return exchange(c, ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return fail("ncclSend: bad peer %lld", peer);
    return post(c, Pending{true, peer, buf, nullptr, count * type_size(t), s});
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return fail("ncclRecv: bad peer %lld", peer);
    return post(c, Pending{false, peer, nullptr, buf, count * type_size(t), s});
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    if (!c) return fail("ncclAllGather: no communicator");
    if (g_group_depth > 0) return fail("collectives inside a group are not supported by this stub");
    const size_t bytes = count * type_size(t);
    std::vector<Pending> ops;
    ops.push_back(Pending{true, -1, send, nullptr, bytes, s});
    for (int q = 0; q < c->nranks; ++q)
        if (q != c->rank) ops.push_back(Pending{false, q, nullptr, static_cast<char *>(recv) + (size_t)q * bytes, bytes, s});
    const ncclResult_t r = exchange(c, ops);
    if (r != ncclSuccess) return r;
    char *own = static_cast<char *>(recv) + (size_t)c->rank * bytes;
    if (c->solo) {
        if (own != send && bytes > 0 && hipMemcpyAsync(own, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("ncclAllGather: own block copy failed");
        return ncclSuccess;
    }
    if (own != send && bytes > 0 && copy(own, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return fail("ncclAllGather: own block copy failed");
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    if (!c) return fail("ncclAllReduce: no communicator");
    if (t != ncclFloat64 || op != ncclSum) return fail("this stub reduces doubles with ncclSum only");
    if (g_group_depth > 0) return fail("collectives inside a group are not supported by this stub");
    if (c->solo) {                           // nranks times the rank's own contribution stands for the sum; all in stream order
        if (count > 16) return fail("solo all-reduce of more than 16 doubles");
        double *slot = c->h_red + (c->red_slot++ % 64) * 16;
        if (count > 0 && hipMemcpyAsync(slot, send, count * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return fail("hipMemcpyAsync failed (solo mode)");
        struct Scale { double *p; size_t n; double f, us; };
        auto fn = [](void *u) {
            Scale *q = static_cast<Scale *>(u);
            for (size_t i = 0; i < q->n; ++i) q->p[i] *= q->f;
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < q->us) { }
            delete q;
        };
        if (hipLaunchHostFunc(s, fn, new Scale{slot, count, (double)c->nranks, c->latency_us}) != hipSuccess) return fail("hipLaunchHostFunc failed (solo mode)");
        if (count > 0 && hipMemcpyAsync(recv, slot, count * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return fail("hipMemcpyAsync failed (solo mode)");
        ++c->n_calls;
        return ncclSuccess;
    }
    if (sync(s) != hipSuccess) return fail("hipStreamSynchronize failed");
    const size_t bytes = count * sizeof(double);
    char *mine = own_box(c, (long long)bytes);
    if (!mine) return fail("cannot map the outbox");
    BoxHdr *h = reinterpret_cast<BoxHdr *>(mine);
    h->n_msgs = 1;
    h->msg[0] = MsgHdr{-1, (long long)sizeof(BoxHdr), (long long)bytes};
    bool ok = bytes == 0 || copy(mine + sizeof(BoxHdr), send, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (!ok) c->ctl->failed.store(1);
    if (!barrier(c)) return fail("a rank did not reach the all-reduce (rank %lld of %lld)", c->rank, c->nranks);
    std::vector<double> sum(count, 0.0);
    for (int q = 0; q < c->nranks; ++q) {                 // rank order: every rank gets the same bits
        const char *pb = peer_box(c, q);
        const BoxHdr *ph = pb ? reinterpret_cast<const BoxHdr *>(pb) : nullptr;
        if (!ph || ph->n_msgs != 1 || ph->msg[0].bytes != (long long)bytes) {
            ok = false;
            fail("all-reduce of %lld doubles meets a different call on rank %lld", (long long)count, q);
            continue;
        }
        const double *v = reinterpret_cast<const double *>(pb + ph->msg[0].off);
        for (size_t i = 0; i < count; ++i) sum[i] += v[i];
    }
    if (ok && bytes > 0 && copy(recv, sum.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false;
    if (!ok) c->ctl->failed.store(1);
    if (!barrier(c)) return fail("a rank did not finish the all-reduce");
    ++c->n_calls;
    return c->ctl->failed.load() ? ncclInvalidUsage : ncclSuccess;
}

}  // extern "C"
