// qbh_ckpt.cpp -- the reference's Lanczos checkpoints from the C ABI: file format of vec_disk_write /
// vec_disk_read (src/miscellaneous.cc:391-469: int64 n | n * sizeof(T) payload | CRC-32 of header + payload) and the
// two-phase directory protocol of ckpt_lanczos_init / ckpt_lanczos_update for the "val" purposes
// (src/ckpt.cc:23-297), so that a C++ host can checkpoint and resume the long runs; a run written by either side is
// readable by the other.  The CRC-32 here is an independent table implementation (reflected polynomial 0xEDB88320,
// initial value and final xor 0xFFFFFFFF = boost::crc_32_type); the Python mirror uses zlib's, tests compare the two.
//
// The device loop is not interrupted per step: qbh_lanczos_ckpt advances the recurrence in chunks through the
// continuation form lanczos(k, np, ...) (src/qbasis.h:1030) and commits a checkpoint after every chunk.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <string>
#include <vector>

#include "qbh_internal.hpp"

namespace fs = std::filesystem;

namespace {

struct Crc32 {
    uint32_t table[256];
    Crc32()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            table[i] = c;
        }
    }
    uint32_t update(uint32_t crc, const void *data, size_t n) const       // crc: running value, start with 0
    {
        const unsigned char *p = static_cast<const unsigned char *>(data);
        uint32_t c = crc ^ 0xFFFFFFFFu;
        for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
        return c ^ 0xFFFFFFFFu;
    }
};
const Crc32 g_crc;

constexpr size_t kChunk = 1024 * 1024;       // 1 MiB pieces, as the reference streams them

std::string P(const std::string &dir, const std::string &name) { return (fs::path(dir) / name).string(); }

bool write_pod(const std::string &file, const void *data, size_t n)
{
    std::ofstream f(file, std::ios::out | std::ios::binary);
    f.write(static_cast<const char *>(data), (std::streamsize)n);
    return (bool)f;
}

void rm(const std::string &file)
{
    std::error_code ec;
    fs::remove(fs::path(file), ec);
}

std::vector<int64_t> lanczos_vec_indices(const std::string &dir)
{
    std::vector<int64_t> ks;
    std::error_code ec;
    for (auto &e : fs::directory_iterator(dir, ec)) {
        const std::string n = e.path().filename().string();
        if (n.size() > 12 && n.compare(0, 8, "lanczosV") == 0 && n.compare(n.size() - 4, 4, ".dat") == 0) {
            const std::string mid = n.substr(8, n.size() - 12);
            if (!mid.empty() && mid.find_first_not_of("0123456789") == std::string::npos) ks.push_back(std::stoll(mid));
        }
    }
    std::sort(ks.begin(), ks.end());
    return ks;
}

}  // namespace

extern "C" uint32_t qbh_crc32(uint32_t crc, const void *data, int64_t nbytes)
{
    return (data && nbytes > 0) ? g_crc.update(crc, data, (size_t)nbytes) : crc;
}

// src/miscellaneous.cc:439-469
extern "C" int qbh_vec_disk_write(const char *filename, int64_t n, int elem_size, const void *x)
{
    if (!filename || n < 0 || elem_size <= 0 || (n > 0 && !x)) return QBH_EINVAL;
    std::ofstream f(filename, std::ios::out | std::ios::binary);
    if (!f) {
        qbh::set_error("qbh_vec_disk_write: cannot open %s", filename);
        return QBH_EINVAL;
    }
    uint32_t crc = g_crc.update(0, &n, sizeof(int64_t));
    f.write(reinterpret_cast<const char *>(&n), sizeof(int64_t));
    const char *p = static_cast<const char *>(x);
    size_t left = (size_t)n * (size_t)elem_size;
    while (left > 0) {
        const size_t c = left < kChunk ? left : kChunk;
        f.write(p, (std::streamsize)c);
        crc = g_crc.update(crc, p, c);
        p += c;
        left -= c;
    }
    f.write(reinterpret_cast<const char *>(&crc), sizeof(uint32_t));
    f.close();
    return f ? QBH_OK : QBH_EINVAL;
}

// src/miscellaneous.cc:391-436: 0 on success, 1 where the reference returns 1 (missing file, wrong size, wrong n, bad CRC)
extern "C" int qbh_vec_disk_read(const char *filename, int64_t n, int elem_size, void *x)
{
    if (!filename || n < 0 || elem_size <= 0 || (n > 0 && !x)) return QBH_EINVAL;
    std::error_code ec;
    if (!fs::exists(fs::path(filename), ec)) return 1;
    const uint64_t ideal = sizeof(int64_t) + (uint64_t)n * (uint64_t)elem_size + sizeof(uint32_t);
    if (fs::file_size(fs::path(filename), ec) != ideal) return 1;
    std::ifstream f(filename, std::ios::in | std::ios::binary);
    int64_t n_check = 0;
    f.read(reinterpret_cast<char *>(&n_check), sizeof(int64_t));
    if (!f || n_check != n) return 1;
    uint32_t crc = g_crc.update(0, &n, sizeof(int64_t));
    char *p = static_cast<char *>(x);
    size_t left = (size_t)n * (size_t)elem_size;
    while (left > 0) {
        const size_t c = left < kChunk ? left : kChunk;
        f.read(p, (std::streamsize)c);
        if (!f) return 1;
        crc = g_crc.update(crc, p, c);
        p += c;
        left -= c;
    }
    uint32_t stored = 0;
    f.read(reinterpret_cast<char *>(&stored), sizeof(uint32_t));
    return (f && stored == crc) ? 0 : 1;
}

// ckpt_lanczos_update, "val" purposes (src/ckpt.cc:178-297).  v: host vectors in the reference's slots (v[j] at
// (j%2)*dim, phi0 at 2*dim for sr_val1).  In two phases so that the ranks of a row-sharded run can meet between them
// (qbh_lanczos_ckpt): WRITE leaves the first marker and the complete new data beside the old; COMMIT leaves the second marker,
// removes the old data and the markers.  Until every rank has finished WRITE no rank starts COMMIT, so a crash anywhere leaves
// either the old step readable on every rank or the new step complete on every rank.
static int lanczos_update_write(const std::string &d, int64_t m, int64_t maxit, int64_t dim, int cnt_accuE0, double accuracy, double theta0_prev,
                                double theta1_prev, const qbh_z *v, const double *hessenberg, const std::string &pur)
{
    std::error_code ec;
    if (fs::exists(d, ec) && !fs::is_directory(d, ec)) fs::remove_all(d, ec);
    fs::create_directories(d, ec);
    rm(P(d, "lczs_updt.Qckpt1"));
    rm(P(d, "lczs_updt.Qckpt2"));
    if (!write_pod(P(d, "lczs_updt.Qckpt1"), &m, sizeof(int64_t))) return QBH_EINVAL;
    QBH_TRY(qbh_vec_disk_write(P(d, "HessenbergA.dat.new").c_str(), m, 8, hessenberg + maxit));
    QBH_TRY(qbh_vec_disk_write(P(d, "HessenbergB.dat.new").c_str(), m + 1, 8, hessenberg));
    // The reference skips V(m-1) when a file of that name exists (it wrote it one step earlier).  Here updates are
    // `every` steps apart, so an existing V(m-1) may be a stale file of another run and is rewritten -- but with
    // every == 1 it is also a file of the LAST COMMITTED checkpoint, which must stay readable until Qckpt2 exists:
    // both vectors go to a temporary name and are renamed over the final one (atomic: old or new, never torn).
    auto write_vec_atomic = [&](int64_t k) -> int {
        const std::string fin = P(d, "lanczosV" + std::to_string(k) + ".dat"), tmp = fin + ".tmp";
        QBH_TRY(qbh_vec_disk_write(tmp.c_str(), dim, 16, v + (k % 2) * dim));
        std::error_code e2;
        fs::rename(tmp, fin, e2);
        return e2 ? QBH_EINVAL : QBH_OK;
    };
    if (m > 0) QBH_TRY(write_vec_atomic(m - 1));
    QBH_TRY(write_vec_atomic(m));
    const bool val0 = pur.find("val0") != std::string::npos;
    if (!val0) QBH_TRY(qbh_vec_disk_write(P(d, "lanczosY0.dat.new").c_str(), dim, 16, v + 2 * dim));
    {
        char buf[28];
        std::memcpy(buf, &cnt_accuE0, 4);
        std::memcpy(buf + 4, &accuracy, 8);
        std::memcpy(buf + 12, &theta0_prev, 8);
        std::memcpy(buf + 20, &theta1_prev, 8);
        if (!write_pod(P(d, "lczs_mlns.dat.new"), buf, sizeof(buf))) return QBH_EINVAL;
    }
    return QBH_OK;
}
static int lanczos_update_commit(const std::string &d, int64_t m, const std::string &pur)
{
    std::error_code ec;
    const bool val0 = pur.find("val0") != std::string::npos;
    if (!write_pod(P(d, "lczs_updt.Qckpt2"), &m, sizeof(int64_t))) return QBH_EINVAL;   // before / after this point: old / new data
    rm(P(d, "HessenbergA.dat"));
    rm(P(d, "HessenbergB.dat"));
    for (int64_t k : lanczos_vec_indices(d))
        if (k < m - 1 || k > m) rm(P(d, "lanczosV" + std::to_string(k) + ".dat"));      // older steps, and stale higher ones
    rm(P(d, "lanczosY0.dat"));
    rm(P(d, "lanczosY1.dat"));
    rm(P(d, "lczs_mlns.dat"));
    fs::rename(P(d, "HessenbergA.dat.new"), P(d, "HessenbergA.dat"), ec);
    fs::rename(P(d, "HessenbergB.dat.new"), P(d, "HessenbergB.dat"), ec);
    if (!val0) fs::rename(P(d, "lanczosY0.dat.new"), P(d, "lanczosY0.dat"), ec);
    fs::rename(P(d, "lczs_mlns.dat.new"), P(d, "lczs_mlns.dat"), ec);
    rm(P(d, "lczs_updt.Qckpt1"));
    rm(P(d, "lczs_updt.Qckpt2"));
    return QBH_OK;
}
extern "C" int qbh_ckpt_lanczos_update(const char *dir, int64_t m, int64_t maxit, int64_t dim, int cnt_accuE0, double accuracy,
                                       double theta0_prev, double theta1_prev, const qbh_z *v, const double *hessenberg,
                                       const char *purpose)
{
    if (!dir || !v || !hessenberg || !purpose || m < 0 || m >= maxit || dim <= 0) return QBH_EINVAL;
    const std::string d(dir), pur(purpose);
    if (pur.find("val") == std::string::npos) {
        qbh::set_error("qbh_ckpt_lanczos_update: only the \"val\" purposes are checkpointed (got %s)", purpose);
        return QBH_EUNSUPP;
    }
    QBH_TRY(lanczos_update_write(d, m, maxit, dim, cnt_accuE0, accuracy, theta0_prev, theta1_prev, v, hessenberg, pur));
    return lanczos_update_commit(d, m, pur);
}

// ckpt_lanczos_init, "val" purposes (src/ckpt.cc:23-176): finishes or rewinds an interrupted update, finds the last
// step on disk and loads it.  *k_out = 0 when there is nothing usable (start from scratch); otherwise v, hessenberg
// and the bookkeeping scalars hold step *k_out.
extern "C" int qbh_ckpt_lanczos_init(const char *dir, int64_t *k_out, int64_t maxit, int64_t dim, int *cnt_accuE0, double *accuracy,
                                     double *theta0_prev, double *theta1_prev, qbh_z *v, double *hessenberg, const char *purpose)
{
    if (!dir || !k_out || !cnt_accuE0 || !accuracy || !theta0_prev || !theta1_prev || !v || !hessenberg || !purpose) return QBH_EINVAL;
    *k_out = 0;
    const std::string d(dir), pur(purpose);
    std::error_code ec;
    if (!fs::is_directory(d, ec)) return QBH_OK;
    const std::string mk1 = P(d, "lczs_updt.Qckpt1"), mk2 = P(d, "lczs_updt.Qckpt2");
    const char *renames[] = {"HessenbergA.dat", "HessenbergB.dat", "lanczosY0.dat", "lanczosY1.dat", "lczs_mlns.dat"};
    if (fs::exists(mk1, ec) && fs::file_size(mk1, ec) == sizeof(int64_t)) {
        int64_t k = 0;
        {
            std::ifstream f(mk1, std::ios::in | std::ios::binary);
            f.read(reinterpret_cast<char *>(&k), sizeof(int64_t));
        }
        if (fs::exists(mk2, ec)) {                               // src/ckpt.cc:50-79: new data complete, finish the clean-up
            for (const char *n : renames)
                if (fs::exists(P(d, std::string(n) + ".new"), ec)) {
                    rm(P(d, n));
                    fs::rename(P(d, std::string(n) + ".new"), P(d, n), ec);
                }
            for (int64_t kk : lanczos_vec_indices(d))
                if (kk < k - 1 || kk > k) rm(P(d, "lanczosV" + std::to_string(kk) + ".dat"));
            rm(mk1);
            rm(mk2);
        } else {
            // src/ckpt.cc:80-97 rewinds ONE step because its updates are one step apart.  Here they are `every` steps
            // apart: the committed step is the one the old HessenbergA.dat was written for (its int64 header); only its
            // two vectors may survive, whatever the torn update had already written (V(m_old+1) for a 2-step chunk).
            int64_t m_old = -1;
            {
                std::ifstream f(P(d, "HessenbergA.dat"), std::ios::in | std::ios::binary);
                int64_t h = 0;
                f.read(reinterpret_cast<char *>(&h), sizeof(int64_t));
                if (f) m_old = h;
            }
            for (const char *n : renames) rm(P(d, std::string(n) + ".new"));
            for (int64_t kk : lanczos_vec_indices(d))
                if (m_old < 1 || (kk != m_old - 1 && kk != m_old)) rm(P(d, "lanczosV" + std::to_string(kk) + ".dat"));
            rm(mk1);
        }
        std::error_code e3;                                      // temporary names of an interrupted vector write
        for (auto &e : fs::directory_iterator(d, e3)) {
            const std::string nm = e.path().filename().string();
            if (nm.size() > 4 && nm.compare(nm.size() - 4, 4, ".tmp") == 0) rm(e.path().string());
        }
    } else {
        rm(mk1);
        rm(mk2);
    }
    const std::vector<int64_t> ks = lanczos_vec_indices(d);
    if (ks.empty()) return QBH_OK;
    int64_t m = ks[0];                                           // src/ckpt.cc:101-111
    for (size_t i = 1; i < ks.size() && ks[i] == m + 1; ++i) m = ks[i];
    if (m == 0 || m >= maxit || std::find(ks.begin(), ks.end(), m - 1) == ks.end()) return QBH_OK;
    // where the reference asserts on unreadable files, nothing is loaded and the caller starts from scratch
    std::vector<double> a((size_t)m), b((size_t)m + 1);
    std::vector<qbh_z> v1((size_t)dim), v2((size_t)dim);
    if (qbh_vec_disk_read(P(d, "HessenbergA.dat").c_str(), m, 8, a.data()) != 0) return QBH_OK;
    if (qbh_vec_disk_read(P(d, "HessenbergB.dat").c_str(), m + 1, 8, b.data()) != 0) return QBH_OK;
    if (qbh_vec_disk_read(P(d, "lanczosV" + std::to_string(m - 1) + ".dat").c_str(), dim, 16, v1.data()) != 0) return QBH_OK;
    if (qbh_vec_disk_read(P(d, "lanczosV" + std::to_string(m) + ".dat").c_str(), dim, 16, v2.data()) != 0) return QBH_OK;
    std::vector<qbh_z> y0;
    if (pur.find("val0") == std::string::npos) {
        y0.resize((size_t)dim);
        if (qbh_vec_disk_read(P(d, "lanczosY0.dat").c_str(), dim, 16, y0.data()) != 0) return QBH_OK;
    }
    char buf[28];
    {
        std::ifstream f(P(d, "lczs_mlns.dat"), std::ios::in | std::ios::binary);
        f.read(buf, sizeof(buf));
        if (!f) return QBH_OK;
    }
    std::memcpy(cnt_accuE0, buf, 4);
    std::memcpy(accuracy, buf + 4, 8);
    std::memcpy(theta0_prev, buf + 12, 8);
    std::memcpy(theta1_prev, buf + 20, 8);
    std::memcpy(hessenberg + maxit, a.data(), (size_t)m * 8);
    std::memcpy(hessenberg, b.data(), ((size_t)m + 1) * 8);
    std::memcpy(v + ((m - 1) % 2) * dim, v1.data(), (size_t)dim * 16);
    std::memcpy(v + (m % 2) * dim, v2.data(), (size_t)dim * 16);
    if (!y0.empty()) std::memcpy(v + 2 * dim, y0.data(), (size_t)dim * 16);
    *k_out = m;
    return QBH_OK;
}

// ---- row shards: every rank keeps a reference-format checkpoint of ITS slice in dir/shard<r>of<P>/ ----
// The ranks meet (through the communicator's own all-reduce) between the two phases of an update and before a resume, so that
// they always continue from the same step: <= 16 ranks (one slot of qbh_comm.d_scal per rank).
namespace {

std::string shard_dir(const qbh_csr *A, const char *dir)
{
    if (!A->has_comm || A->comm.nranks <= 1) return std::string(dir);
    return P(dir, "shard" + std::to_string(A->comm.rank) + "of" + std::to_string(A->comm.nranks));
}

// every rank's number, on every rank
int rank_gather(qbh_csr *A, double mine, std::vector<double> &all)
{
    const int np = A->has_comm ? A->comm.nranks : 1;
    all.assign((size_t)np, mine);
    if (np <= 1) return QBH_OK;
    if (np > 16) {
        qbh::set_error("checkpoints of row shards: at most 16 ranks");
        return QBH_EUNSUPP;
    }
    double v[16] = {0};
    v[A->comm.rank] = mine;
    QBH_HIP(hipMemcpyAsync(A->comm.d_scal, v, sizeof(v), hipMemcpyHostToDevice, A->stream));
    if (A->comm.allreduce_sum(A->comm.ctx, 0, 16) != 0) {
        qbh::set_error("allreduce_sum hook failed (checkpoint agreement)");
        return QBH_ECOMM;
    }
    QBH_HIP(hipMemcpyAsync(v, A->comm.d_scal, sizeof(v), hipMemcpyDeviceToHost, A->stream));
    QBH_HIP(hipStreamSynchronize(A->stream));
    for (int q = 0; q < np; ++q) all[(size_t)q] = v[q];
    return QBH_OK;
}

// the rank's verdict on a phase, agreed: QBH_OK only when every rank succeeded
int agree_ok(qbh_csr *A, int my_rc, const char *what)
{
    std::vector<double> all;
    QBH_TRY(rank_gather(A, my_rc == QBH_OK ? 0.0 : 1.0, all));
    for (double f : all)
        if (f != 0.0) {
            if (my_rc == QBH_OK) qbh::set_error("%s failed on a peer rank", what);
            return my_rc != QBH_OK ? my_rc : QBH_ECOMM;
        }
    return QBH_OK;
}

bool file_is(const std::string &f, int64_t n, int elem)
{
    std::error_code ec;
    return fs::exists(f, ec) && fs::file_size(f, ec) == (uint64_t)(sizeof(int64_t) + (uint64_t)n * (uint64_t)elem + sizeof(uint32_t));
}

// Before a resume on shards: a rank whose update was interrupted AFTER its new data was complete may finish it only when every
// rank is at that step (pending-and-complete, or already committed); otherwise everybody falls back to the step before.
// pending_marker: lczs_updt.Qckpt1 / CG_updt.Qckpt1; complete(m): the rank's new files for step m are all there (sizes);
// committed(): the step the directory holds without markers (-1: none).
template <typename Complete, typename Committed>
int settle_pending(qbh_csr *A, const std::string &d, const char *mk1_name, const char *mk2_name, Complete complete, Committed committed)
{
    if (!A->has_comm || A->comm.nranks <= 1) return QBH_OK;
    std::error_code ec;
    const std::string mk1 = P(d, mk1_name), mk2 = P(d, mk2_name);
    int64_t cand = -1;
    double state = 0.0;                              // 0 no pending update, 1 pending and complete, -1 pending and incomplete
    if (fs::exists(mk1, ec) && fs::file_size(mk1, ec) == sizeof(int64_t)) {
        std::ifstream f(mk1, std::ios::in | std::ios::binary);
        f.read(reinterpret_cast<char *>(&cand), sizeof(int64_t));
        state = (fs::exists(mk2, ec) || complete(cand)) ? 1.0 : -1.0;
    } else {
        cand = committed();
    }
    std::vector<double> st, cs;
    QBH_TRY(rank_gather(A, state, st));
    QBH_TRY(rank_gather(A, (double)cand, cs));
    bool all_there = true;
    double top = -1.0;
    for (size_t q = 0; q < st.size(); ++q) {
        all_there = all_there && st[q] >= 0.0;
        top = std::max(top, cs[q]);
    }
    for (size_t q = 0; q < st.size(); ++q) all_there = all_there && cs[q] == top;
    if (state == 1.0 && !fs::exists(mk2, ec)) {
        if (all_there) {
            if (!write_pod(mk2, &cand, sizeof(int64_t))) return QBH_EINVAL;      // the init below finishes the clean-up
        }
        // else: the marker stays alone and the init below rewinds, as on every other rank
    } else if (state == 1.0 && fs::exists(mk2, ec) && !all_there) {
        qbh::set_error("checkpoint of row shards: this rank committed step %lld but a peer does not hold it", (long long)cand);
        return QBH_EINVAL;
    }
    return QBH_OK;
}

}  // namespace

// lanczos(0, maxit - 1, ...) of the reference with enable_ckpt = true: resume from `dir` if it holds a usable step,
// otherwise start from v (v[0] normalised, phi0 at 2*dim for sr_val1); a checkpoint is committed every `every` steps
// and at the end.  max_steps > 0 stops after that many new steps (an "interrupted" run for tests and time-sliced
// jobs).  *converged reports whether the stop rule fired.  v and hessenberg are host arrays as in qbh_lanczos.
// A row shard under a communicator (collective call): every rank checkpoints its slice in dir/shard<r>of<P>/ with the same
// file names and protocol, and the ranks agree between the phases (above).
extern "C" int qbh_lanczos_ckpt(const qbh_csr *Ac, int64_t maxit, int64_t *m_out, qbh_z *v_host, double *hessenberg,
                                const char *purpose, int64_t every, int64_t max_steps, const char *dir, int *converged,
                                qbh_solver_info *info_out)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !m_out || !v_host || !hessenberg || !purpose || !dir || maxit < 3 || every < 1) return QBH_EINVAL;
    const std::string pur(purpose);
    const int nvec = pur.find("val1") != std::string::npos ? 3 : 2;
    const int64_t n = A->nrows;
    if (A->nrows != A->ncols && !A->has_comm) {
        qbh::set_error("qbh_lanczos_ckpt: a row shard needs its communicator (every rank checkpoints its own slice)");
        return QBH_EUNSUPP;
    }
    const std::string d = shard_dir(A, dir);
    const bool val0 = pur.find("val0") != std::string::npos;
    {
        std::error_code ec;
        fs::create_directories(d, ec);
    }
    QBH_TRY(settle_pending(
        A, d, "lczs_updt.Qckpt1", "lczs_updt.Qckpt2",
        [&](int64_t m) {
            return m > 0 && file_is(P(d, "HessenbergA.dat.new"), m, 8) && file_is(P(d, "HessenbergB.dat.new"), m + 1, 8) &&
                   file_is(P(d, "lanczosV" + std::to_string(m - 1) + ".dat"), n, 16) && file_is(P(d, "lanczosV" + std::to_string(m) + ".dat"), n, 16) &&
                   fs::exists(P(d, "lczs_mlns.dat.new")) && (val0 || file_is(P(d, "lanczosY0.dat.new"), n, 16));
        },
        [&]() -> int64_t {
            const std::vector<int64_t> ks = lanczos_vec_indices(d);
            if (ks.empty()) return -1;
            int64_t m = ks[0];
            for (size_t i = 1; i < ks.size() && ks[i] == m + 1; ++i) m = ks[i];
            return m;
        }));
    int cnt = 0;
    double accuracy = 0.0, t0 = 0.0, t1 = 0.0;
    int64_t k = 0;
    QBH_TRY(qbh_ckpt_lanczos_init(d.c_str(), &k, maxit, n, &cnt, &accuracy, &t0, &t1, v_host, hessenberg, purpose));
    {   // every rank from the same step, or every rank from scratch
        std::vector<double> ks;
        QBH_TRY(rank_gather(A, (double)k, ks));
        for (double q : ks)
            if (q != (double)k) k = 0;
        std::vector<double> k2;
        QBH_TRY(rank_gather(A, (double)k, k2));
        for (double q : k2)
            if (q == 0.0) k = 0;
    }
    if (k == 0) {                                  // from scratch: nothing of an earlier run may survive
        std::error_code ec;
        for (auto &e : fs::directory_iterator(d, ec)) {
            const std::string nm = e.path().filename().string();
            if (nm.compare(0, 8, "lanczosV") == 0 || nm.compare(0, 8, "lanczosY") == 0 || nm.compare(0, 10, "Hessenberg") == 0 ||
                nm.compare(0, 5, "lczs_") == 0)
                rm(e.path().string());
        }
    }
    qbh_z *d_v = nullptr;
    QBH_TRY(qbh_vec_alloc(&d_v, (int64_t)nvec * n));
    int rc = qbh_vec_upload(A, d_v, v_host, (int64_t)nvec * n);
    qbh_solver_info info{};
    info.resume = k > 0 ? 1 : 0;
    info.cnt_accuE0 = cnt;
    info.accuracy = accuracy;
    info.theta0_prev = t0;
    info.theta1_prev = t1;
    int64_t m = k, done = 0;
    bool conv = false;
    // per-iteration log rows (src/lanczos.cc:102-128) of all chunks, in the caller's buffer
    qbh_lanczos_row *log_base = info_out ? info_out->log : nullptr;
    const int64_t log_cap = info_out ? info_out->log_cap : 0;
    int64_t log_total = 0;
    while (rc == QBH_OK && m < maxit - 1) {
        int64_t np = std::min<int64_t>(every, maxit - 1 - m);
        if (max_steps > 0) np = std::min<int64_t>(np, max_steps - done);
        if (np <= 0) break;
        int64_t m_new = m;
        info.log = (log_base && log_total < log_cap) ? log_base + log_total : nullptr;
        info.log_cap = info.log ? log_cap - log_total : 0;
        rc = qbh_lanczos_dev(A, m, np, maxit, &m_new, d_v, hessenberg, purpose, &info);
        if (rc != QBH_OK) break;
        log_total += info.log ? std::min<int64_t>(info.log_len, info.log_cap) : 0;
        info.resume = 1;                            // the bookkeeping returned in info seeds the next chunk
        done += m_new - m;
        const bool early = m_new < m + np;
        m = m_new;
        rc = qbh_vec_download(A, v_host, d_v, (int64_t)nvec * n);
        if (rc != QBH_OK) break;
        // phase 1 on every rank, the ranks meet, phase 2 on every rank (one rank: the two phases back to back)
        rc = lanczos_update_write(d, m, maxit, n, (int)info.cnt_accuE0, info.accuracy, info.theta0_prev, info.theta1_prev, v_host, hessenberg, pur);
        rc = agree_ok(A, rc, "writing the checkpoint");
        if (rc != QBH_OK) break;
        rc = lanczos_update_commit(d, m, pur);
        if (early || (info.cnt_accuE0 > 15 && info.accuracy < QBH_LANCZOS_PRECISION)) {
            conv = true;
            break;
        }
    }
    (void)qbh_vec_free(d_v);
    if (rc != QBH_OK) return rc;
    *m_out = m;
    if (converged) *converged = conv ? 1 : 0;
    if (info_out) {
        *info_out = info;
        info_out->log = log_base;
        info_out->log_cap = log_cap;
        info_out->log_len = log_total;
    }
    return QBH_OK;
}

// ------------------------------------------------------------------- CG checkpoints (src/ckpt.cc:344-517) ----
// Files CG_V<m>.dat, CG_R<m>.dat, CG_P<m>.dat (vec_disk_write format) and the two markers CG_updt.Qckpt1 / .Qckpt2: the first is
// written before the new vectors, the second after them ("before / after this point, have to use old / new data"), then the
// vectors of the step before and both markers go.
namespace {

std::vector<int64_t> cg_indices(const std::string &dir)
{
    std::vector<int64_t> ks;
    std::error_code ec;
    for (auto &e : fs::directory_iterator(dir, ec)) {
        const std::string n = e.path().filename().string();
        if (n.size() > 8 && n.compare(0, 4, "CG_V") == 0 && n.compare(n.size() - 4, 4, ".dat") == 0) {
            const std::string mid = n.substr(4, n.size() - 8);
            if (!mid.empty() && mid.find_first_not_of("0123456789") == std::string::npos) ks.push_back(std::stoll(mid));
        }
    }
    std::sort(ks.begin(), ks.end());
    return ks;
}
std::string cgf(const std::string &d, char which, int64_t m) { return P(d, std::string("CG_") + which + std::to_string(m) + ".dat"); }

int cg_update_write(const std::string &d, int64_t m, int64_t dim, const qbh_z *v, const qbh_z *r, const qbh_z *p)
{
    std::error_code ec;
    if (fs::exists(d, ec) && !fs::is_directory(d, ec)) fs::remove_all(d, ec);
    fs::create_directories(d, ec);
    rm(P(d, "CG_updt.Qckpt1"));
    rm(P(d, "CG_updt.Qckpt2"));
    if (!write_pod(P(d, "CG_updt.Qckpt1"), &m, sizeof(int64_t))) return QBH_EINVAL;
    QBH_TRY(qbh_vec_disk_write(cgf(d, 'V', m).c_str(), dim, 16, v));
    QBH_TRY(qbh_vec_disk_write(cgf(d, 'R', m).c_str(), dim, 16, r));
    QBH_TRY(qbh_vec_disk_write(cgf(d, 'P', m).c_str(), dim, 16, p));
    return QBH_OK;
}
int cg_update_commit(const std::string &d, int64_t m)
{
    if (!write_pod(P(d, "CG_updt.Qckpt2"), &m, sizeof(int64_t))) return QBH_EINVAL;     // src/ckpt.cc:467: fs::copy of the first marker
    for (int64_t k : cg_indices(d))                        // the reference removes step m - 1; updates `every` steps apart: whatever is older
        if (k != m)
            for (char w : {'V', 'R', 'P'}) rm(cgf(d, w, k));
    rm(P(d, "CG_updt.Qckpt1"));
    rm(P(d, "CG_updt.Qckpt2"));
    return QBH_OK;
}

}  // namespace

extern "C" int qbh_ckpt_cg_update(const char *dir, int64_t m, int64_t dim, const qbh_z *v, const qbh_z *r, const qbh_z *p)
{
    if (!dir || !v || !r || !p || m < 0 || dim <= 0) return QBH_EINVAL;
    QBH_TRY(cg_update_write(dir, m, dim, v, r, p));
    return cg_update_commit(dir, m);
}

// ckpt_CG_init (src/ckpt.cc:344-433): finishes or rewinds an interrupted update, finds the step on disk and loads v, r, p.
// *m_out = 0: nothing usable.
extern "C" int qbh_ckpt_cg_init(const char *dir, int64_t *m_out, int64_t maxit, int64_t dim, qbh_z *v, qbh_z *r, qbh_z *p)
{
    if (!dir || !m_out || !v || !r || !p || dim <= 0) return QBH_EINVAL;
    *m_out = 0;
    const std::string d(dir);
    std::error_code ec;
    if (!fs::is_directory(d, ec)) return QBH_OK;
    const std::string mk1 = P(d, "CG_updt.Qckpt1"), mk2 = P(d, "CG_updt.Qckpt2");
    if (fs::exists(mk1, ec) && fs::file_size(mk1, ec) == sizeof(int64_t)) {
        int64_t k = 0;
        {
            std::ifstream f(mk1, std::ios::in | std::ios::binary);
            f.read(reinterpret_cast<char *>(&k), sizeof(int64_t));
        }
        if (fs::exists(mk2, ec)) {                         // :382-392 new data complete: the older vectors go
            for (int64_t kk : cg_indices(d))
                if (kk != k)
                    for (char w : {'V', 'R', 'P'}) rm(cgf(d, w, kk));
        } else {                                           // :393-405 torn update: its files go, the step before stays
            for (char w : {'V', 'R', 'P'}) rm(cgf(d, w, k));
        }
        rm(mk1);
        rm(mk2);
    } else {
        rm(mk1);
        rm(mk2);
    }
    const std::vector<int64_t> ks = cg_indices(d);         // :410-413: the first step present (after the clean-up: the only one)
    if (ks.empty()) return QBH_OK;
    const int64_t m = ks.back();
    if (m <= 0 || m >= maxit) return QBH_OK;
    std::vector<qbh_z> tv((size_t)dim), tr((size_t)dim), tp((size_t)dim);
    // where the reference asserts on an unreadable file, nothing is loaded and the caller starts from scratch
    if (qbh_vec_disk_read(cgf(d, 'V', m).c_str(), dim, 16, tv.data()) != 0) return QBH_OK;
    if (qbh_vec_disk_read(cgf(d, 'R', m).c_str(), dim, 16, tr.data()) != 0) return QBH_OK;
    if (qbh_vec_disk_read(cgf(d, 'P', m).c_str(), dim, 16, tp.data()) != 0) return QBH_OK;
    std::memcpy(v, tv.data(), (size_t)dim * 16);
    std::memcpy(r, tr.data(), (size_t)dim * 16);
    std::memcpy(p, tp.data(), (size_t)dim * 16);
    *m_out = m;
    return QBH_OK;
}

// ckpt_CG_clean (src/ckpt.cc:480-517)
extern "C" int qbh_ckpt_cg_clean(const char *dir)
{
    if (!dir) return QBH_EINVAL;
    const std::string d(dir);
    for (int64_t k : cg_indices(d))
        for (char w : {'V', 'R', 'P'}) rm(cgf(d, w, k));
    return QBH_OK;
}

// eigenvec_CG of the reference with enable_ckpt = true (src/lanczos.cc:281-341): resume from `dir` when it holds a step,
// otherwise start from v (m = 0); a checkpoint every `every` CG steps and at the end; max_steps > 0 stops after that many new
// steps.  *converged: the loop ended by its own rule (residual below lanczos_precision with |v| = 1).  The residual history
// (log_CG.txt, src/lanczos.cc:308-311,334-337) is returned in info->cg_resid[1..m] for the steps made by THIS call.
// Row shards (collective): dir/shard<r>of<P>/ as for qbh_lanczos_ckpt.
extern "C" int qbh_eigenvec_cg_ckpt(const qbh_csr *Ac, int64_t maxit, int64_t *m_out, double E0, double *accu_out, qbh_z *v_host, qbh_z *r_host,
                                    qbh_z *p_host, qbh_z *pp_host, int64_t every, int64_t max_steps, const char *dir, int *converged,
                                    qbh_solver_info *info_out)
{
    qbh_csr *A = const_cast<qbh_csr *>(Ac);
    if (!A || !m_out || !accu_out || !v_host || !r_host || !p_host || !pp_host || !dir || maxit < 2 || every < 1) return QBH_EINVAL;
    if (A->nrows != A->ncols && !A->has_comm) {
        qbh::set_error("qbh_eigenvec_cg_ckpt: a row shard needs its communicator");
        return QBH_EUNSUPP;
    }
    const int64_t n = A->nrows;
    const std::string d = shard_dir(A, dir);
    {
        std::error_code ec;
        fs::create_directories(d, ec);
    }
    QBH_TRY(settle_pending(
        A, d, "CG_updt.Qckpt1", "CG_updt.Qckpt2",
        [&](int64_t m) { return file_is(cgf(d, 'V', m), n, 16) && file_is(cgf(d, 'R', m), n, 16) && file_is(cgf(d, 'P', m), n, 16); },
        [&]() -> int64_t {
            const std::vector<int64_t> ks = cg_indices(d);
            return ks.empty() ? -1 : ks.back();
        }));
    int64_t m = 0;
    QBH_TRY(qbh_ckpt_cg_init(d.c_str(), &m, maxit, n, v_host, r_host, p_host));
    {
        std::vector<double> ms;
        QBH_TRY(rank_gather(A, (double)m, ms));
        for (double q : ms)
            if (q != (double)m) m = 0;
        std::vector<double> m2;
        QBH_TRY(rank_gather(A, (double)m, m2));
        for (double q : m2)
            if (q == 0.0) m = 0;
    }
    if (m == 0) QBH_TRY(qbh_ckpt_cg_clean(d.c_str()));
    qbh_z *dv = nullptr;
    QBH_TRY(qbh_vec_alloc(&dv, 4 * n));
    qbh_z *hv[4] = {v_host, r_host, p_host, pp_host};
    int rc = QBH_OK;
    for (int i = 0; i < 3 && rc == QBH_OK; ++i) rc = qbh_vec_upload(A, dv + (size_t)i * (size_t)n, hv[i], n);
    double accu = 0.0;
    int64_t done = 0;
    bool conv = false;
    std::vector<double> resid((size_t)maxit + 2, 0.0);
    while (rc == QBH_OK && m < maxit) {
        int64_t stop = std::min<int64_t>(m + every, maxit);
        if (max_steps > 0) stop = std::min<int64_t>(stop, m + (max_steps - done));
        if (stop <= m) break;
        qbh_solver_info ci{};
        ci.cg_resid = resid.data();
        int64_t m_new = m;
        rc = qbh_eigenvec_cg_dev(A, stop, &m_new, E0, &accu, dv, dv + n, dv + 2 * n, dv + 3 * n, &ci);
        if (rc != QBH_OK) break;
        if (info_out && info_out->cg_resid)
            for (int64_t j = m + 1; j <= m_new; ++j) info_out->cg_resid[j] = resid[(size_t)j];
        if (info_out) info_out->n_matvec += ci.n_matvec;
        const bool ended = m_new < stop || (m_new == m);        // the loop left by its own rule
        done += m_new - m;
        m = m_new;
        for (int i = 0; i < 3 && rc == QBH_OK; ++i) rc = qbh_vec_download(A, hv[i], dv + (size_t)i * (size_t)n, n);
        if (rc != QBH_OK) break;
        if (m > 0) {
            rc = cg_update_write(d, m, n, v_host, r_host, p_host);
            rc = agree_ok(A, rc, "writing the CG checkpoint");
            if (rc != QBH_OK) break;
            rc = cg_update_commit(d, m);
        }
        if (ended) {
            conv = true;
            break;
        }
    }
    if (rc == QBH_OK) rc = qbh_vec_download(A, pp_host, dv + 3 * (size_t)n, n);
    (void)qbh_vec_free(dv);
    if (rc != QBH_OK) return rc;
    *m_out = m;
    *accu_out = accu;
    if (converged) *converged = conv ? 1 : 0;
    return QBH_OK;
}
