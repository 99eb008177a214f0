// C++ host program over include/qbhip_qbasis.hpp -- the way the reference's C++ host code would
// drive the engine: load a CSR (binary dump written by the test), build qbhip::csr_mat, run
// MultMv, lanczos("sr_val0") + hess_eigen, eigenvec_CG.  Prints results for the Python test.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "qbhip_qbasis.hpp"

using qbhip::cplx;
using qbhip::qint;

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    qint dim = 0, nnz = 0, sym = 0;
    f.read((char *)&dim, 8); f.read((char *)&nnz, 8); f.read((char *)&sym, 8);
    qint *ia = new qint[dim + 1];
    qint *ja = new qint[nnz];
    cplx *val = new cplx[nnz];
    f.read((char *)ia, 8 * (dim + 1)); f.read((char *)ja, 8 * nnz); f.read((char *)val, 16 * nnz);
    std::vector<cplx> x(dim);
    f.read((char *)x.data(), 16 * dim);
    if (!f) { std::cerr << "short read\n"; return 2; }
    try {
        qbhip::csr_mat H(dim, nnz, sym != 0, val, ja, ia);
        std::vector<cplx> y(dim, cplx(1.0, -1.0));
        H.MultMv(x.data(), y.data());
        double sr = 0, si = 0;
        for (auto &e : y) { sr += e.real(); si += e.imag(); }
        H.MultMv2(x.data(), y.data());
        double sr2 = 0;
        for (auto &e : y) sr2 += e.real();
        const qint maxit = 1000;
        std::vector<cplx> v(4 * dim, cplx(0.0));
        for (qint j = 0; j < dim; j++) v[j] = x[j];
        std::vector<double> hess(2 * maxit, 0.0), ritz, s;
        qint m = 0;
        qbhip::lanczos(0, maxit - 1, maxit, m, dim, H, v.data(), hess.data(), "sr_val0");
        qbhip::hess_eigen(hess.data(), maxit, m, "sr", ritz, s);
        const double E0 = ritz[0];
        for (qint j = 0; j < dim; j++) v[2 * dim + j] = x[j];
        double accu = 0;
        qint mcg = 0;
        qbhip::eigenvec_CG(dim, maxit, mcg, H, cplx(E0), accu, v.data() + 2 * dim, v.data(), v.data() + dim, v.data() + 3 * dim);
        qbhip::csr_mat H2(H);                    // deep copy keeps working
        std::vector<cplx> y2(dim);
        H2.MultMv(x.data(), y2.data());
        double d = 0;
        for (qint j = 0; j < dim; j++) d += std::abs(2.0 * y2[j] - y[j]);
        std::printf("OK %.17g %.17g %.17g %lld %.17g %lld %.3e %.3e\n", sr, si, sr2, (long long)m, E0, (long long)mcg, accu, d);
        // the four-stage driver and the device-resident IRAM through the C++ mirror
        qbhip::E0_result r = qbhip::locate_E0_lanczos(H, 2, 1);
        std::vector<double> ev(3);
        std::vector<cplx> evec(3 * dim);
        qint nconv = 0;
        qbhip::iram(dim, H, nullptr, 3, 10, 300, "sr", nconv, ev.data(), evec.data());
        double ov = 0;                       // |<phi0(CG) | phi0(IRAM)>|
        cplx acc(0.0);
        for (qint j = 0; j < dim; j++) acc += std::conj(r.eigenvecs[j]) * evec[j];
        ov = std::abs(acc);
        std::printf("DRV %.17g %.17g %lld %lld %lld %.17g %.17g %.17g %lld %.6e\n", r.E0, r.E1, (long long)r.steps_E0,
                    (long long)r.steps_V0, (long long)r.nconv, ev[0], ev[1], ev[2], (long long)nconv, ov);
    } catch (const std::exception &e) {
        std::printf("EXC %s\n", e.what());
        return 3;
    }
    return 0;
}
