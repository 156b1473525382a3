// ============================================================================
// ref_harness.cpp -- C-ABI shim around the UNMODIFIED reference sources
// (compiled where they lie under /root/reference by oracle/Makefile, output
// only into oracle/_ref/).  TEST INFRASTRUCTURE: used to validate the CPU
// restatement (rails_oracle.cpp) and to generate tests/golden/*.npz.
//
// Only the parts of the reference that build from its own sources plus a
// BLAS/LAPACK that exists in this image are used: StlWrapper / StlVector /
// LapackWrapper / Timer, and the header-only Solver members resid_lanczos and
// compute_restart_vectors.  Solver::solve / dense_solve need SLICOT's
// sb03md_, which this image lacks (no Fortran compiler either): they are NOT
// instantiated here and no stand-in is written for them.
// ============================================================================
#include <cstring>
#include <iostream>
#include <vector>

#include "src/LyapunovSolver.hpp"
#include "src/StlWrapper.hpp"

using RAILS::StlWrapper;
typedef RAILS::Solver<StlWrapper, StlWrapper, StlWrapper> RefSolver;

namespace {

StlWrapper make(int m, int n, const double *data, int ld)
{
    StlWrapper w(m, n);
    double *p = (double *)w;
    int lw = w.LDA();
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) p[i + (size_t)j * lw] = data[i + (size_t)j * ld];
    return w;
}

void out(StlWrapper const &w, double *data, int ld)
{
    double *p = (double *)w;
    int lw = w.LDA();
    for (int j = 0; j < w.N(); ++j)
        for (int i = 0; i < w.M(); ++i) data[i + (size_t)j * ld] = p[i + (size_t)j * lw];
}

} // namespace

extern "C" {

void ref_srand(unsigned s) { std::srand(s); }

// StlWrapper::random (src/StlWrapper.cpp:414-423)
void ref_random(int m, int n, double *X)
{
    StlWrapper w(m, n);
    w.random();
    out(w, X, m);
}

// StlWrapper::dot (src/StlWrapper.cpp:394-412): C = X^T Y
void ref_dot(int m, int a, int b, const double *X, const double *Y, double *C)
{
    StlWrapper x = make(m, a, X, m), y = make(m, b, Y, m);
    StlWrapper c = x.dot(y);
    out(c, C, a);
}

// StlWrapper::operator* (src/StlWrapper.cpp:168-187): Y = X * C
void ref_mult(int m, int k, int r, const double *X, const double *C, double *Y)
{
    StlWrapper x = make(m, k, X, m), c = make(k, r, C, k);
    StlWrapper y = x * c;
    out(y, Y, m);
}

// transposed multiply: Y = X^T * C  (the B_.transpose() * W call, src/LyapunovSolver.hpp:150)
void ref_mult_t(int m, int k, int r, const double *X, const double *C, double *Y)
{
    StlWrapper x = make(m, k, X, m), c = make(m, r, C, m);
    StlWrapper y = x.transpose() * c;
    out(y, Y, k);
}

// StlWrapper::norm (src/StlWrapper.cpp:265-289) and norm_inf (:291-303)
double ref_norm(int m, int n, const double *X)
{
    StlWrapper x = make(m, n, X, m);
    return x.norm();
}
double ref_norm_inf(int m, int n, const double *X)
{
    StlWrapper x = make(m, n, X, m);
    return x.norm_inf();
}

// orthogonalize n1 columns, then push_back n2 more and orthogonalize again (the solver's
// usage, src/LyapunovSolver.hpp:338-340, with the watermark of src/StlWrapper.cpp:305-321)
void ref_orthogonalize(int m, int n1, const double *V1, int n2, const double *V2, double *Out)
{
    StlWrapper v(m, n1 + n2);
    v.resize(n1);
    {
        double *p = (double *)v;
        int lw = v.LDA();
        for (int j = 0; j < n1; ++j)
            for (int i = 0; i < m; ++i) p[i + (size_t)j * lw] = V1[i + (size_t)j * m];
    }
    v.orthogonalize();
    if (n2 > 0) {
        StlWrapper w = make(m, n2, V2, m);
        v.push_back(w);
        v.orthogonalize();
    }
    out(v, Out, m);
}

// StlWrapper::eigs (src/StlWrapper.cpp:433-479): all eigenpairs, ascending
int ref_eigs(int n, const double *A, double *Vout, double *d)
{
    StlWrapper a = make(n, n, A, n);
    StlWrapper v, dd;
    int info = a.eigs(v, dd);
    out(v, Vout, n);
    for (int i = 0; i < n; ++i) d[i] = dd(i, 0);
    return info;
}

// find_largest_eigenvalues (src/StlTools.hpp:12-30)
void ref_find_largest(const double *vals, int n, int N, int *idx)
{
    StlWrapper d = make(n, 1, vals, n);
    std::vector<int> indices;
    RAILS::find_largest_eigenvalues(d, indices, N);
    for (int i = 0; i < N; ++i) idx[i] = indices[i];
}

// Solver::resid_lanczos (src/LyapunovSolver.hpp:367-447).  The start vector comes from
// StlWrapper::random, i.e. from std::rand(): call ref_srand first.
// H out is (max_iter+1)^2 zero-padded col-major; returns the number of Lanczos steps.
int ref_resid_lanczos(int m, int k, int p, const double *AV, const double *V, const double *T, const double *B,
                      int max_iter, double *H, double *evals, double *evecs)
{
    StlWrapper A(1, 1);
    A = 0.0;
    StlWrapper b = make(m, p, B, m);
    RefSolver solver(A, b, A);
    StlWrapper av = make(m, k, AV, m), v = make(m, k, V, m), t = make(k, k, T, k);
    StlWrapper h(max_iter + 1, max_iter + 1), ev(max_iter, 1), evec;
    solver.resid_lanczos(av, v, t, h, evec, ev, max_iter);
    int steps = h.M();
    memset(H, 0, sizeof(double) * (size_t)(max_iter + 1) * (max_iter + 1));
    {
        double *ph = (double *)h;
        int lh = h.LDA();
        for (int j = 0; j < steps; ++j)
            for (int i = 0; i < steps; ++i) H[i + (size_t)j * (max_iter + 1)] = ph[i + (size_t)j * lh];
    }
    for (int i = 0; i < steps; ++i) evals[i] = ev(i, 0);
    out(evec, evecs, m);
    return steps;
}

// Solver::compute_restart_vectors (src/LyapunovSolver.hpp:449-482)
int ref_compute_restart_vectors(int k, const double *T, int num, double tol, double *X)
{
    StlWrapper A(1, 1);
    A = 0.0;
    RefSolver solver(A, A, A);
    StlWrapper t = make(k, k, T, k);
    StlWrapper x;
    solver.compute_restart_vectors(x, t, num, tol);
    out(x, X, k);
    return x.N();
}

} // extern "C"
