// One rank of a row-sharded C++ host run -- what the reference's host program becomes with N GPUs: every process reads
// the SAME host CSR (the unchanged model::generate_Ham_sparse_full output), takes its nnz-balanced row block
// (qbh_balanced_row_cuts), builds only that block on its GPU (qbh_csr_create_rows) and joins the native RCCL
// communicator (qbh_comm_create_rccl); then the unchanged solver call: lanczos("sr_val0") + hess_eigen, eigenvec_CG.
// usage: sharded_main csr.bin rank nranks uid_file [uniform]   (rank 0 writes the ncclUniqueId to uid_file, the others wait;
//        "uniform": equal row blocks -> ncclAllGather instead of the send/recv all-gather-v of nnz-balanced cuts)
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
    const bool uniform = argc > 5 && std::string(argv[5]) == "uniform" && dim % nranks == 0;
    if (uniform) {
        for (int q = 0; q <= nranks; ++q) cuts[q] = dim / nranks * q;
    } else {
        must(qbh_balanced_row_cuts(dim, nnz, (int)sym, ia.data(), ja.data(), nranks, cuts.data()), "qbh_balanced_row_cuts");
    }
    qbh_opts opts;
    qbh_opts_default(&opts);
    opts.device = rank % (qbh_device_count() > 0 ? qbh_device_count() : 1);          // one process per GPU
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
    must(qbh_lanczos_dev(A, 0, maxit - 1, maxit, &m, d_v, hess.data(), "sr_val0", nullptr), "qbh_lanczos_dev");
    must(qbh_hess_eigen(hess.data(), maxit, m, "sr", ritz.data(), s.data()), "qbh_hess_eigen");
    const double E0 = ritz[0];
    must(qbh_vec_randomize(A, d_v + 2 * n, 1), "qbh_vec_randomize");
    int64_t mcg = 0;
    double accu = 0.0;
    must(qbh_eigenvec_cg_dev(A, maxit, &mcg, E0, &accu, d_v + 2 * n, d_v, d_v + n, d_v + 3 * n, nullptr), "qbh_eigenvec_cg_dev");
    double nrm = 0.0;
    must(qbh_nrm2_dev(A, d_v + 2 * n, &nrm), "qbh_nrm2_dev");                         // global norm of the eigenvector
    std::printf("OK %d %d %lld %lld %lld %.17g %lld %.3e %.15g\n", rank, nranks, (long long)cuts[rank], (long long)cuts[rank + 1],
                (long long)m, E0, (long long)mcg, accu, nrm);
    qbh_vec_free(d_v);
    must(qbh_comm_destroy(A), "qbh_comm_destroy");
    qbh_csr_destroy(A);
    return 0;
}
