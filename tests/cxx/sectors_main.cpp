// C++ host program over include/qbhip_qbasis.hpp that follows the reference's own example
// examples/trans_symmetric/latt_square/square_Fermi_Hubbard.cc: the 4x2 torus with 4+4 electrons, t = 1, U = 1.1, all eight
// momentum sectors, locate_E0_lanczos in each, and the reference's asserted energies (:146-153).
#include <cmath>
#include <complex>
#include <cstdio>
#include <iostream>
#include <vector>

#include "qbhip_qbasis.hpp"

using qbhip::cplx;

int main()
{
    const int Lx = 4, Ly = 2, n = Lx * Ly;
    const double t = 1.0, U = 1.1;
    auto site = [&](int x, int y) { return ((x % Lx) + Lx) % Lx + Lx * (((y % Ly) + Ly) % Ly); };
    std::vector<std::pair<int, int>> bonds;                 // the example's loop: (x+1, y) and (x, y+1) from every site
    for (int x = 0; x < Lx; x++)
        for (int y = 0; y < Ly; y++) {
            bonds.push_back({site(x, y), site(x + 1, y)});
            bonds.push_back({site(x, y), site(x, y + 1)});
        }
    std::vector<int32_t> perms;
    std::vector<std::pair<int, int>> shifts;
    for (int tx = 0; tx < Lx; tx++)
        for (int ty = 0; ty < Ly; ty++) {
            for (int s = 0; s < n; s++) perms.push_back(site(s % Lx + tx, s / Lx + ty));
            shifts.push_back({tx, ty});
        }
    const double want[8] = {-14.07605866, -10.50470669, -12.16861094, -12.19847764, -10.54300366, -14.03137587, -12.16861094, -12.19847764};
    try {
        int idx = 0;
        double worst = 0.0;
        for (int m = 0; m < Lx; m++)
            for (int nn = 0; nn < Ly; nn++, idx++) {
                std::vector<cplx> chars;
                for (auto &sh : shifts) chars.push_back(std::exp(cplx(0.0, -2.0 * M_PI * (m * sh.first / (double)Lx + nn * sh.second / (double)Ly))));
                qbhip::csr_mat H = qbhip::hubbard_sector(n, 4, 4, bonds, t, U, perms, chars);
                auto res = qbhip::locate_E0_lanczos(H, 1, 1);
                worst = std::max(worst, std::fabs(res.E0 - want[idx]));
                // the same sector without the stored matrix
                qbhip::csr_mat M = qbhip::hubbard_sector(n, 4, 4, bonds, t, U, perms, chars, nullptr, true);
                auto resm = qbhip::locate_E0_lanczos(M, 1, 1);
                worst = std::max(worst, std::fabs(resm.E0 - want[idx]));
                std::printf("k=(%d,%d) dim %lld E0 %.10f (reference %.8f)\n", m, nn, (long long)H.dimension(), res.E0, want[idx]);
            }
        std::printf("OK %.3e\n", worst);
        return worst < 1e-8 ? 0 : 1;
    } catch (const std::exception &e) {
        std::cout << "EXCEPTION " << e.what() << std::endl;
        return 3;
    }
}
