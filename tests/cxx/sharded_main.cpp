// One rank of a row-sharded C++ host run -- what the reference's host program becomes with N GPUs: every process reads
// the SAME host CSR (the unchanged model::generate_Ham_sparse_full output), takes its nnz-balanced row block
// (qbh_balanced_row_cuts), builds only that block on its GPU (qbh_csr_create_rows) and joins the native RCCL
// communicator (qbh_comm_create_rccl); then the unchanged solver call: lanczos("sr_val0") + hess_eigen, eigenvec_CG.
// usage: sharded_main csr.bin rank nranks uid_file [uniform] [key=value ...]   (rank 0 writes the ncclUniqueId to uid_file, the
//        others wait; "uniform": equal row blocks -> ncclAllGather instead of the send/recv all-gather-v of nnz-balanced cuts)
//        kron=S      the index is major * S + minor (qbh_opts.kron_minor, kron_split = 2): row blocks of whole major indices, every
//                    shard split in place, the ranks exchange the tiled copies of their blocks
//        parts=K     qbh_opts.gather_parts;  unsplit=R  rank R keeps its shard unsplit (the ranks must fall back together)
//        plain=1     complex128 values and vectors (value_dict = 0, real_fast_path = 0)
//        pipeline=0  qbh_opts.lanczos_pipeline = 0 (one host synchronisation per Lanczos step)
//        sparse=0    qbh_opts.sparse_gather = 0 (every rank's whole block travels to everybody)
//        realwire=0  qbh_opts.real_wire = 0 (split shards: 16-byte elements on the links even for a real solve)
//        ckpt=DIR every=K maxsteps=N   the Lanczos run through qbh_lanczos_ckpt (collective: every rank checkpoints its slice in
//                    DIR/shard<r>of<P>/), stopped after N new steps when N > 0; then only the Lanczos part is dumped
//        dump=PREFIX rank r writes PREFIX.r.bin: m, mcg, E0, a_j / b_j (2 m doubles), its slice of the eigenvector
#include <chrono>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "qbhip.h"

typedef std::complex<double> cplx;

static void must(int rc, const char *what)
{
    if (rc != QBH_OK) {
        std::printf("ERR %s: %s (%s)\n", what, qbh_strerror(rc), qbh_last_error());
        std::exit(3);
    }
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int rank = std::atoi(argv[2]), nranks = std::atoi(argv[3]);
    std::ifstream f(argv[1], std::ios::binary);
    int64_t dim = 0, nnz = 0, sym = 0;
    f.read((char *)&dim, 8); f.read((char *)&nnz, 8); f.read((char *)&sym, 8);
    std::vector<int64_t> ia(dim + 1), ja(nnz);
    std::vector<cplx> val(nnz);
    f.read((char *)ia.data(), 8 * (dim + 1)); f.read((char *)ja.data(), 8 * nnz); f.read((char *)val.data(), 16 * nnz);
    if (!f) return 2;
    // the communicator id: rank 0 creates it, the others read it from the file
    char uid[128];
    if (rank == 0) {
        must(qbh_rccl_unique_id(uid), "qbh_rccl_unique_id");
        std::ofstream o(std::string(argv[4]) + ".tmp", std::ios::binary);
        o.write(uid, 128);
        o.close();
        std::rename((std::string(argv[4]) + ".tmp").c_str(), argv[4]);
    } else {
        for (int t = 0; t < 600; ++t) {
            std::ifstream i(argv[4], std::ios::binary);
            if (i && i.read(uid, 128)) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    std::vector<int64_t> cuts(nranks + 1);
    bool uniform = false;
    int64_t kron = 0;
    int parts = 0, unsplit = -1, plain = 0, realwire = 1, pipeline = 1, sparse = 1, det = 0;
    std::string dump, ckdir;
    long long every = 10, maxsteps = 0;
    for (int i = 5; i < argc; ++i) {
        const std::string a(argv[i]);
        if (a == "uniform") uniform = dim % nranks == 0;
        else if (a.rfind("kron=", 0) == 0) kron = std::atoll(a.c_str() + 5);
        else if (a.rfind("parts=", 0) == 0) parts = std::atoi(a.c_str() + 6);
        else if (a.rfind("unsplit=", 0) == 0) unsplit = std::atoi(a.c_str() + 8);
        else if (a.rfind("plain=", 0) == 0) plain = std::atoi(a.c_str() + 6);
        else if (a.rfind("realwire=", 0) == 0) realwire = std::atoi(a.c_str() + 9);
        else if (a.rfind("pipeline=", 0) == 0) pipeline = std::atoi(a.c_str() + 9);
        else if (a.rfind("sparse=", 0) == 0) sparse = std::atoi(a.c_str() + 7);
        else if (a.rfind("det=", 0) == 0) det = std::atoi(a.c_str() + 4);
        else if (a.rfind("ckpt=", 0) == 0) ckdir = a.substr(5);
        else if (a.rfind("every=", 0) == 0) every = std::atoll(a.c_str() + 6);
        else if (a.rfind("maxsteps=", 0) == 0) maxsteps = std::atoll(a.c_str() + 9);
        else if (a.rfind("dump=", 0) == 0) dump = a.substr(5);
        else return 2;
    }
    if (kron > 0) {                                  // whole major indices per rank, as even as they come
        const int64_t nu = dim / kron;
        for (int q = 0; q <= nranks; ++q) cuts[q] = (nu * q / nranks) * kron;
        uniform = uniform && nu % nranks == 0;
    } else if (uniform) {
        for (int q = 0; q <= nranks; ++q) cuts[q] = dim / nranks * q;
    } else {
        must(qbh_balanced_row_cuts(dim, nnz, (int)sym, ia.data(), ja.data(), nranks, cuts.data()), "qbh_balanced_row_cuts");
    }
    qbh_opts opts;
    qbh_opts_default(&opts);
    opts.device = rank % (qbh_device_count() > 0 ? qbh_device_count() : 1);          // one process per GPU
    if (plain) opts.value_dict = 0, opts.real_fast_path = 0;
    if (kron > 0) {
        opts.kron_minor = kron;
        opts.kron_split = rank == unsplit ? 0 : 2;
    }
    opts.gather_parts = parts;
    opts.real_wire = realwire;
    opts.lanczos_pipeline = pipeline;
    opts.sparse_gather = sparse;
    opts.deterministic = det;                      // 1: static walks (their slot arithmetic needs grids that are multiples of 8)
    qbh_csr *A = nullptr;
    must(qbh_csr_create_rows(&A, dim, nnz, (int)sym, ia.data(), ja.data(), reinterpret_cast<const qbh_z *>(val.data()), cuts[rank],
                             cuts[rank + 1], &opts), "qbh_csr_create_rows");
    must(qbh_comm_create_rccl(A, uid, rank, nranks, uniform ? nullptr : cuts.data()), "qbh_comm_create_rccl");
    const int64_t n = cuts[rank + 1] - cuts[rank], maxit = 1000;
    qbh_z *d_v = nullptr;
    must(qbh_vec_alloc(&d_v, 4 * n), "qbh_vec_alloc");
    must(qbh_vec_randomize(A, d_v, 1), "qbh_vec_randomize");                         // this rank's slice of the global start vector
    std::vector<double> hess(2 * maxit, 0.0), ritz(maxit), s((size_t)maxit * maxit);
    int64_t m = 0;
    if (!ckdir.empty()) {
        // the checkpointed run (host vectors, as lanczos() of the reference with enable_ckpt): interrupted after maxsteps, or to the end
        std::vector<cplx> hv(2 * n);
        must(qbh_vec_download(A, reinterpret_cast<qbh_z *>(hv.data()), d_v, 2 * n), "qbh_vec_download");
        int conv = 0;
        must(qbh_lanczos_ckpt(A, maxit, &m, reinterpret_cast<qbh_z *>(hv.data()), hess.data(), "sr_val0", every, maxsteps, ckdir.c_str(), &conv, nullptr),
             "qbh_lanczos_ckpt");
        double E0c = 0.0;
        if (m > 2) {
            must(qbh_hess_eigen(hess.data(), maxit, m, "sr", ritz.data(), s.data()), "qbh_hess_eigen");
            E0c = ritz[0];
        }
        if (!dump.empty()) {
            std::ofstream o(dump + "." + std::to_string(rank) + ".bin", std::ios::binary);
            const int64_t head[2] = {m, (int64_t)conv};
            o.write((const char *)head, 16);
            o.write((const char *)&E0c, 8);
            o.write((const char *)(hess.data() + maxit), 8 * m);
            o.write((const char *)hess.data(), 8 * m);
        }
        std::printf("OK %d %d %lld %lld %lld %.17g %d 0 1 kron 0 parts 0 cols16 0 wire 0 need 1 sparse 0\n", rank, nranks, (long long)cuts[rank], (long long)cuts[rank + 1],
                    (long long)m, E0c, conv);
        qbh_vec_free(d_v);
        must(qbh_comm_destroy(A), "qbh_comm_destroy");
        qbh_csr_destroy(A);
        return 0;
    }
    must(qbh_lanczos_dev(A, 0, maxit - 1, maxit, &m, d_v, hess.data(), "sr_val0", nullptr), "qbh_lanczos_dev");
    must(qbh_hess_eigen(hess.data(), maxit, m, "sr", ritz.data(), s.data()), "qbh_hess_eigen");
    const double E0 = ritz[0];
    must(qbh_vec_randomize(A, d_v + 2 * n, 1), "qbh_vec_randomize");
    int64_t mcg = 0;
    double accu = 0.0;
    must(qbh_eigenvec_cg_dev(A, maxit, &mcg, E0, &accu, d_v + 2 * n, d_v, d_v + n, d_v + 3 * n, nullptr), "qbh_eigenvec_cg_dev");
    double nrm = 0.0;
    must(qbh_nrm2_dev(A, d_v + 2 * n, &nrm), "qbh_nrm2_dev");                         // global norm of the eigenvector
    qbh_csr_info inf;
    must(qbh_csr_get_info(A, &inf), "qbh_csr_get_info");
    if (!dump.empty()) {
        std::vector<cplx> vec(n);
        must(qbh_vec_download(A, reinterpret_cast<qbh_z *>(vec.data()), d_v + 2 * n, n), "qbh_vec_download");
        std::ofstream o(dump + "." + std::to_string(rank) + ".bin", std::ios::binary);
        const int64_t head[2] = {m, mcg};
        o.write((const char *)head, 16);
        o.write((const char *)&E0, 8);
        o.write((const char *)(hess.data() + maxit), 8 * m);       // a_j
        o.write((const char *)hess.data(), 8 * m);                 // b_j
        o.write((const char *)vec.data(), 16 * n);
    }
    std::printf("OK %d %d %lld %lld %lld %.17g %lld %.17g %.15g kron %lld parts %d cols16 %d wire %d need %.4f sparse %d\n", rank, nranks, (long long)cuts[rank],
                (long long)cuts[rank + 1], (long long)m, E0, (long long)mcg, accu, nrm, (long long)inf.kron_minor, inf.gather_parts, inf.kron_cols16,
                inf.wire_element_bytes, inf.gather_needed_frac, inf.gather_sparse);
    qbh_vec_free(d_v);
    must(qbh_comm_destroy(A), "qbh_comm_destroy");
    qbh_csr_destroy(A);
    return 0;
}
