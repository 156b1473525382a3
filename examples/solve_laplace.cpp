// A complete C++ program on the drop-in classes, no Python: 7-point Laplacian on an nx x ny x nz grid, B = p random columns,
// solved for X = V T V' with A X + X A' + B B' = 0 on both HIP back ends of the solver template.
//
//   hipcc -O2 -std=c++17 -Iinclude -Irails_amd/include examples/solve_laplace.cpp -Lrails_amd/lib -lrails_hip -Wl,-rpath,'$ORIGIN/../rails_amd/lib' -o solve_laplace
//   ./solve_laplace [nx ny nz p]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "rails/HipSolverOps.hpp"      // rails::HipSolver       = Solver<HipOperatorWrapper, HipMultiVectorWrapper, HostDenseMatrix>
#include "rails/SubspaceSolverOps.hpp" // rails::SubspaceSolver  = Solver<SubspaceOperator, SubspaceMultiVector, HostDenseMatrix>

struct ParameterList { // anything with get(name, default): Teuchos::ParameterList in the reference (src/LyapunovSolver.hpp:27-36)
    std::map<std::string, double> p;
    template <typename T>
    T get(std::string const &name, T def)
    {
        auto it = p.find(name);
        return it == p.end() ? def : (T)it->second;
    }
};

// || A X + X A' + B B' ||_F / || B B' ||_F for X = V T V' (host, dense: small problems only)
static double residual(int m, std::vector<int64_t> const &rp, std::vector<int32_t> const &ci, std::vector<double> const &va, std::vector<double> const &B, int p,
                       std::vector<double> const &V, int k, rails::HostDenseMatrix const &T)
{
    std::vector<double> X((size_t)m * m, 0.0), VT((size_t)m * k, 0.0), R((size_t)m * m, 0.0);
    for (int j = 0; j < k; ++j)
        for (int l = 0; l < k; ++l)
            for (int i = 0; i < m; ++i) VT[i + (size_t)j * m] += V[i + (size_t)l * m] * T(l, j);
    for (int j = 0; j < m; ++j)
        for (int l = 0; l < k; ++l)
            for (int i = 0; i < m; ++i) X[i + (size_t)j * m] += VT[i + (size_t)l * m] * V[j + (size_t)l * m];
    double bb = 0.0, rr = 0.0;
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int l = 0; l < p; ++l) s += B[i + (size_t)l * m] * B[j + (size_t)l * m];
            R[i + (size_t)j * m] = s;
            bb += s * s;
        }
    for (int i = 0; i < m; ++i)
        for (int64_t q = rp[i]; q < rp[i + 1]; ++q)
            for (int j = 0; j < m; ++j) {
                R[i + (size_t)j * m] += va[q] * X[ci[q] + (size_t)j * m]; // A X
                R[j + (size_t)i * m] += va[q] * X[j + (size_t)ci[q] * m]; // X A'
            }
    for (size_t i = 0; i < R.size(); ++i) rr += R[i] * R[i];
    return std::sqrt(rr / bb);
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 12, ny = argc > 2 ? atoi(argv[2]) : 10, nz = argc > 3 ? atoi(argv[3]) : 8, p = argc > 4 ? atoi(argv[4]) : 4;
    const int m = nx * ny * nz;
    std::vector<int64_t> rp(1, 0);
    std::vector<int32_t> ci;
    std::vector<double> va;
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                const int r = x + nx * (y + ny * z);
                auto add = [&](int c, double v) {
                    ci.push_back(c);
                    va.push_back(v);
                };
                if (z > 0) add(r - nx * ny, 1.0);
                if (y > 0) add(r - nx, 1.0);
                if (x > 0) add(r - 1, 1.0);
                add(r, -6.0);
                if (x < nx - 1) add(r + 1, 1.0);
                if (y < ny - 1) add(r + nx, 1.0);
                if (z < nz - 1) add(r + nx * ny, 1.0);
                rp.push_back((int64_t)ci.size());
            }
    rails_ctx *ctx = nullptr;
    if (rails_ctx_create(0, nullptr, &ctx) != RAILS_OK) {
        std::printf("no gfx950 device: %s\n", rails_last_error());
        return 2;
    }
    rails::set_default_context(ctx);
    int failures = 0;
    {
        rails::HipOperatorWrapper A(ctx, m, m, rp.data(), ci.data(), va.data());
        rails::HipMultiVectorWrapper B(m, p, ctx);
        rails_ctx_set_seed(ctx, 7, 0);
        B.random();
        std::vector<double> Bh((size_t)m * p);
        B.to_host(Bh.data(), m);
        ParameterList params;
        const bool small = m <= 2000; // small grids: tight tolerance and an explicit residual check; larger ones: the solver's defaults
        params.p = {{"Restart size", small ? 60.0 : 160.0}, {"Reduced size", small ? 30.0 : 80.0}, {"Expand size", (double)p},
                    {"Lanczos iterations", (double)p + 2}, {"Tolerance", small ? 1e-6 : 1e-3}};
        for (int backend = 0; backend < 2; ++backend) {
            rails_ctx_set_seed(ctx, 11, 0);
            rails::HostDenseMatrix T;
            std::vector<double> Vh;
            int ret, k, trips;
            auto t0 = std::chrono::steady_clock::now();
            if (backend == 0) { // direct back end: device panels
                rails::HipSolver solver(A, B, A);
                solver.set_parameters(params);
                solver.set_verbose(false);
                rails::HipMultiVectorWrapper V(m, 1, ctx);
                ret = solver.solve(V, T);
                k = V.N();
                trips = solver.trips();
                Vh.resize((size_t)m * k);
                V.to_host(Vh.data(), m);
            } else { // coordinates in one orthonormal device basis
                auto basis = std::make_shared<rails::SubspaceBasis>(ctx, m, m, 2 * 180 + p + 64);
                rails::SubspaceMultiVector Bc = rails::SubspaceMultiVector::Absorb(basis, B);
                rails::SubspaceOperator Ac(A, basis);
                rails::SubspaceSolver solver(Ac, Bc, Ac);
                solver.set_parameters(params);
                solver.set_verbose(false);
                rails::SubspaceMultiVector V(basis, 1);
                ret = solver.solve(V, T);
                k = V.N();
                trips = solver.trips();
                Vh.resize((size_t)m * k);
                V.to_host(Vh.data(), m);
            }
            double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            double rel = m <= 2000 ? residual(m, rp, ci, va, Bh, p, Vh, k, T) : -1.0;
            std::printf("%s back end: return %d, %d iterations, V is %d x %d, %.3f s, ||A X + X A' + B B'||_F / ||B B'||_F = %.2e\n",
                        backend ? "coordinate-space" : "direct", ret, trips, m, k, sec, rel);
            if (ret != 0 || (rel >= 0 && !(rel < 1e-4))) failures++;
        }
    }
    rails_ctx_destroy(ctx);
    std::printf(failures ? "FAILED\n" : "OK\n");
    return failures ? 1 : 0;
}
