// TEST SCAFFOLDING: runs rails::Solver<CpuDense, CpuDense, CpuDense> (the product's solver template on a
// plain CPU backend) on a problem read from binary files and writes V and T back; tests/test_host_logic.py
// compares the result with the oracle.  usage: driver A.bin B.bin n p seed out_prefix [name=value ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "CpuDense.hpp"
#include "rails/LyapunovSolver.hpp"

using cpu::CpuDense;

struct ParameterList {
    std::map<std::string, double> p;
    template <typename T>
    T get(std::string const &name, T def)
    {
        auto it = p.find(name);
        return it == p.end() ? def : (T)it->second;
    }
};

static void read(const char *path, double *dst, size_t n)
{
    FILE *f = fopen(path, "rb");
    if (!f || fread(dst, sizeof(double), n, f) != n) {
        fprintf(stderr, "cannot read %s\n", path);
        exit(2);
    }
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    int n = atoi(argv[3]), p = atoi(argv[4]);
    cpu::rng().seed = strtoull(argv[5], nullptr, 10);
    cpu::rng().stream = 0;
    std::string prefix = argv[6];
    ParameterList params;
    int v0cols = 0;
    std::string v0file;
    for (int i = 7; i < argc; ++i) {
        std::string s(argv[i]);
        size_t eq = s.find('=');
        std::string key = s.substr(0, eq), val = s.substr(eq + 1);
        if (key == "V0")
            v0file = val;
        else if (key == "V0cols")
            v0cols = atoi(val.c_str());
        else
            params.p[key] = atof(val.c_str());
    }
    CpuDense A(n, n), B(n, p);
    read(argv[1], (double *)A, (size_t)n * n);
    read(argv[2], (double *)B, (size_t)n * p);
    rails::Solver<CpuDense, CpuDense, CpuDense> solver(A, B, A);
    int prc = solver.set_parameters(params);
    solver.set_verbose(false);
    CpuDense V(n, std::max(1, v0cols)), T;
    if (v0cols > 0) {
        read(v0file.c_str(), (double *)V, (size_t)n * v0cols);
        // warm start: the caller's V is orthonormal; mark it so (the watermark travels with the object)
        V.orthogonalize();
    }
    int rc = prc ? 100 + prc : solver.solve(V, T);
    FILE *f = fopen((prefix + ".txt").c_str(), "w");
    fprintf(f, "%d %d %d\n", rc, solver.trips(), prc ? 0 : V.N());
    for (double r : solver.residual_history()) fprintf(f, "%.17g\n", r);
    fclose(f);
    if (!prc) {
        int k = V.N();
        f = fopen((prefix + ".V").c_str(), "wb");
        for (int j = 0; j < k; ++j) fwrite(&V(0, j), sizeof(double), n, f);
        fclose(f);
        f = fopen((prefix + ".T").c_str(), "wb");
        for (int j = 0; j < k; ++j) fwrite(&T(0, j), sizeof(double), k, f);
        fclose(f);
    }
    return 0;
}
