// ============================================================================
// rails_oracle.cpp -- CPU ORACLE for the RAILS inner loop.  TEST INFRASTRUCTURE.
//
// This file is a plain restatement, for the CPU, of the algorithm on the hot
// path of the reference (Sbte/RAILS, StlWrapper path).  It exists ONLY as the
// checker for the HIP product: only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load it.  The product (rails_amd/) never
// includes, links or calls anything in this directory.
//
// Parity pin: see oracle/README.md.  In short: every op and resid_lanczos /
// compute_restart_vectors are checked against the reference's own Stl sources
// compiled unmodified into oracle/_ref (tests/golden/*.npz were generated from
// that build); the projected Lyapunov solve (SLICOT sb03md is a third-party
// dependency that is absent here, no pinned version, cmake/FindSLICOT.cmake:29)
// is a Bartels-Stewart restatement pinned by the reference's known-answer
// tests (test/SlicotWrapper_test.cpp:7-38, test/LyapunovSolverEpetra_test.cpp
// :19-48,103-106,174-177) and by residual checks of test/LyapunovSolver_test.cpp.
// The generalized (M != I) path has no C++ reference (only MATLAB, not
// runnable here): "parity unpinned" for that path, residual-checked only.
//
// Storage is column-major with a leading dimension, like the reference's
// StlWrapper (src/StlVector.cpp:47-50).  Each function cites the reference
// lines it follows.
// ============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <utility>
#include <vector>
#include <dlfcn.h>
#include <omp.h>

// ----------------------------------------------------------------------------
// LAPACK (host, run-time resolved: the GPU box has no system LAPACK; the image
// carries scipy's OpenBLAS and MKL).  Only dsyev / dgees / dtrsyl are needed.
// ----------------------------------------------------------------------------
namespace {

typedef void (*dsyev_t)(const char *, const char *, const int *, double *, const int *, double *,
                        double *, const int *, int *);
typedef int (*select_t)(const double *, const double *);
typedef void (*dgees_t)(const char *, const char *, select_t, const int *, double *, const int *,
                        int *, double *, double *, double *, const int *, double *, const int *,
                        int *, int *);
typedef void (*dtrsyl_t)(const char *, const char *, const int *, const int *, const int *,
                         const double *, const int *, const double *, const int *, double *,
                         const int *, double *, int *);
typedef void (*dpotrf_t)(const char *, const int *, double *, const int *, int *);
typedef void (*setthreads_t)(int);

struct Lapack {
    void *h = nullptr;
    dsyev_t dsyev = nullptr;
    dgees_t dgees = nullptr;
    dtrsyl_t dtrsyl = nullptr;
    dpotrf_t dpotrf = nullptr;
    std::string path;
};
Lapack g_lapack;

void *sym2(void *h, const char *name)
{
    std::string s = std::string("scipy_") + name;
    void *p = dlsym(h, s.c_str());
    if (!p) p = dlsym(h, name);
    return p;
}

int lapack_try(const char *path)
{
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    Lapack L;
    L.h = h;
    L.dsyev = (dsyev_t)sym2(h, "dsyev_");
    L.dgees = (dgees_t)sym2(h, "dgees_");
    L.dtrsyl = (dtrsyl_t)sym2(h, "dtrsyl_");
    L.dpotrf = (dpotrf_t)sym2(h, "dpotrf_");
    if (!L.dsyev || !L.dgees || !L.dtrsyl || !L.dpotrf) {
        dlclose(h);
        return -2;
    }
    // The oracle parallelises over rows itself; keep the small dense LAPACK calls serial
    // and bit-reproducible.
    setthreads_t st = (setthreads_t)dlsym(h, "scipy_openblas_set_num_threads");
    if (!st) st = (setthreads_t)dlsym(h, "openblas_set_num_threads");
    if (st) st(1);
    L.path = path;
    g_lapack = L;
    return 0;
}

} // namespace

extern "C" int orc_lapack_init(const char *path)
{
    if (g_lapack.h) return 0;
    if (path && *path && lapack_try(path) == 0) return 0;
    const char *env = getenv("RAILS_LAPACK_LIB");
    if (env && *env && lapack_try(env) == 0) return 0;
    static const char *cands[] = {
        "/usr/local/lib/python3.10/dist-packages/scipy.libs/libscipy_openblas-68440149.so",
        "libopenblas.so.0", "libopenblas.so", "liblapack.so.3", "liblapack.so",
        "/opt/conda/lib/libmkl_rt.so", nullptr};
    for (int i = 0; cands[i]; ++i)
        if (lapack_try(cands[i]) == 0) return 0;
    fprintf(stderr, "rails_oracle: no LAPACK found (set RAILS_LAPACK_LIB)\n");
    return -1;
}

extern "C" const char *orc_lapack_path() { return g_lapack.path.c_str(); }

namespace {

// ----------------------------------------------------------------------------
// Dense kernels (column-major).  These replace the DGEMM calls of
// src/StlWrapper.cpp:181 and :407; summation over the long (row) dimension is
// done in fixed chunks so results do not depend on the thread count.
// ----------------------------------------------------------------------------
const int ROWCHUNK = 2048;

// Row-partitioned runs (world_size > 1 CPU tests): every reduction over rows is summed over the ranks through
// this hook, and the operator apply fetches ghost rows through the halo hook -- the same two exchanges the
// HIP library makes (include/rails_hip.h: rails_allreduce_fn, rails_halo_fn).
typedef int (*orc_allreduce_fn)(double *buf, size_t n);
typedef int (*orc_halo_fn)(const double *send, double *recv, int ncols);
orc_allreduce_fn g_allreduce = nullptr;
orc_halo_fn g_halo = nullptr;
std::vector<int64_t> g_send_rows;
int64_t g_n_ghost = 0;
int64_t g_m_global = 0;

// C (a x b, ldc) = X^T Y ; X is m x a (ldx), Y is m x b (ldy).  src/StlWrapper.cpp:394-412
void gemm_tn(int m, int a, int b, const double *X, int ldx, const double *Y, int ldy, double *C,
             int ldc)
{
    if (a <= 0 || b <= 0) return;
    int nch = (m + ROWCHUNK - 1) / ROWCHUNK;
    if (nch < 1) nch = 1;
    std::vector<double> part((size_t)nch * a * b, 0.0);
#pragma omp parallel for schedule(static)
    for (int c = 0; c < nch; ++c) {
        int r0 = c * ROWCHUNK, r1 = std::min(m, r0 + ROWCHUNK);
        double *P = &part[(size_t)c * a * b];
        for (int j = 0; j < b; ++j) {
            const double *y = Y + (size_t)j * ldy;
            for (int i = 0; i < a; ++i) {
                const double *x = X + (size_t)i * ldx;
                double s = 0.0;
                for (int r = r0; r < r1; ++r) s += x[r] * y[r];
                P[i + (size_t)j * a] = s;
            }
        }
    }
    std::vector<double> packed((size_t)a * b);
    for (int j = 0; j < b; ++j)
        for (int i = 0; i < a; ++i) {
            double s = 0.0;
            for (int c = 0; c < nch; ++c) s += part[(size_t)c * a * b + i + (size_t)j * a];
            packed[i + (size_t)j * a] = s;
        }
    if (g_allreduce) g_allreduce(packed.data(), packed.size());
    for (int j = 0; j < b; ++j)
        for (int i = 0; i < a; ++i) C[i + (size_t)j * ldc] = packed[i + (size_t)j * a];
}

// Y (m x r, ldy) = beta*Y + alpha * X (m x k, ldx) * C (k x r, ldc).  src/StlWrapper.cpp:168-187
void gemm_nn(int m, int k, int r, double alpha, const double *X, int ldx, const double *C, int ldc,
             double beta, double *Y, int ldy)
{
    if (r <= 0) return;
    int nch = (m + ROWCHUNK - 1) / ROWCHUNK;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < nch; ++c) {
        int r0 = c * ROWCHUNK, r1 = std::min(m, r0 + ROWCHUNK);
        for (int j = 0; j < r; ++j) {
            double *y = Y + (size_t)j * ldy;
            if (beta == 0.0)
                for (int i = r0; i < r1; ++i) y[i] = 0.0;
            else if (beta != 1.0)
                for (int i = r0; i < r1; ++i) y[i] *= beta;
            for (int l = 0; l < k; ++l) {
                double w = alpha * C[l + (size_t)j * ldc];
                const double *x = X + (size_t)l * ldx;
                for (int i = r0; i < r1; ++i) y[i] += w * x[i];
            }
        }
    }
}

// small dense helpers (no threading)
void small_gemm(char ta, char tb, int M, int N, int K, const double *A, int lda, const double *B,
                int ldb, double *C, int ldc)
{
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            double s = 0.0;
            for (int l = 0; l < K; ++l) {
                double a = (ta == 'N') ? A[i + (size_t)l * lda] : A[l + (size_t)i * lda];
                double b = (tb == 'N') ? B[l + (size_t)j * ldb] : B[j + (size_t)l * ldb];
                s += a * b;
            }
            C[i + (size_t)j * ldc] = s;
        }
}

// symmetric eigen-decomposition, ascending.  src/LapackWrapper.cpp:20-39 (DSYEV 'V','U')
int sym_eig(int n, double *a, int lda, double *w)
{
    if (n <= 0) return 0;
    int info = 0, lwork = -1;
    double wq = 0.0;
    g_lapack.dsyev("V", "U", &n, a, &lda, w, &wq, &lwork, &info);
    lwork = (int)wq;
    std::vector<double> work(std::max(1, lwork));
    g_lapack.dsyev("V", "U", &n, a, &lda, w, work.data(), &lwork, &info);
    return info;
}

// 2-norm of an m x n block: sqrt(max |eig(X^T X)|).  src/StlWrapper.cpp:265-289
double norm2(int m, int n, const double *X, int ldx)
{
    if (n <= 0) return 0.0;
    std::vector<double> G((size_t)n * n), w(n);
    gemm_tn(m, n, n, X, ldx, X, ldx, G.data(), n);
    sym_eig(n, G.data(), n, w.data());
    double mx = 0.0;
    for (int i = 0; i < n; ++i) mx = std::max(mx, std::sqrt(std::fabs(w[i])));
    return mx;
}

void scal(int m, double s, double *x)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) x[i] *= s;
}

// Column-wise CGS2 with pre/post normalisation.  src/StlWrapper.cpp:305-321
void orthogonalize(int m, double *V, int ldv, int from, int n)
{
    std::vector<double> c(std::max(1, n));
    for (int i = from; i < n; ++i) {
        double *v = V + (size_t)i * ldv;
        scal(m, 1.0 / norm2(m, 1, v, ldv), v); // v /= v.norm()  (/= is *= 1/x, :139-143)
        if (i) {
            for (int k = 0; k < 2; ++k) {
                gemm_tn(m, i, 1, V, ldv, v, ldv, c.data(), i);        // V.dot(v)
                gemm_nn(m, i, 1, -1.0, V, ldv, c.data(), i, 1.0, v, ldv); // v -= V * (..)
            }
        }
        scal(m, 1.0 / norm2(m, 1, v, ldv), v);
    }
}

// indices of the N largest |values| (std::sort, not stable).  src/StlTools.hpp:12-30
bool eig_sorter(std::pair<int, double> const &a, std::pair<int, double> const &b)
{
    return std::abs(a.second) > std::abs(b.second);
}
void find_largest(const double *vals, int n, int N, std::vector<int> &idx)
{
    std::vector<std::pair<int, double>> iv;
    for (int i = 0; i < n; ++i) iv.push_back(std::pair<int, double>(i, vals[i]));
    std::sort(iv.begin(), iv.end(), eig_sorter);
    for (int i = 0; i < N; ++i) idx.push_back(iv[i].first);
}

// ----------------------------------------------------------------------------
// Random fills.  mode 0 = the reference's generator (src/StlWrapper.cpp:414-423:
// default_random_engine seeded by std::rand(), uniform_real(-1,1), row index
// outer / column inner); mode 1 = the counter-based generator shared with the
// HIP product (value depends only on (seed, stream, global row, column)).
// ----------------------------------------------------------------------------
inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline double counter_uniform(uint64_t seed, uint64_t stream, uint64_t row, uint64_t col)
{
    uint64_t h = splitmix64(seed ^ splitmix64(stream * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
    h = splitmix64(h ^ splitmix64(row * 0x9E3779B97F4A7C15ull + col * 0xC2B2AE3D27D4EB4Full + 1));
    // 53 random bits -> [0,1) -> (-1,1)
    double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    return 2.0 * u - 1.0;
}

struct Rng {
    int mode;
    uint64_t seed;
    uint64_t stream;
    long row0; // global index of local row 0 (row-partitioned runs)
};

void random_fill(Rng &rng, int m, int n, double *X, int ldx)
{
    if (rng.mode == 0) {
        std::default_random_engine generator(std::rand());
        std::uniform_real_distribution<double> distribution(-1, 1);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) X[i + (size_t)j * ldx] = distribution(generator);
    } else {
        uint64_t s = rng.stream++;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j)
                X[i + (size_t)j * ldx] = counter_uniform(rng.seed, s, (uint64_t)(rng.row0 + i), j);
    }
}

// ----------------------------------------------------------------------------
// Operators: dense column-major (the Stl Matrix role, src/StlWrapper.cpp:168-187)
// or CSR (the role Epetra_CrsMatrix plays in src/Epetra_OperatorWrapper.cpp:75-91).
// ----------------------------------------------------------------------------
struct Op {
    int m = 0;
    const double *dense = nullptr; // column-major m x m
    int ldd = 0;
    const int64_t *rowptr = nullptr;
    const int32_t *col = nullptr;
    const double *val = nullptr;
    bool identity = false;
};

void csr_spmm(int m, const int64_t *rp, const int32_t *ci, const double *va, int nc,
              const double *X, int ldx, double *Y, int ldy)
{
#pragma omp parallel for schedule(static, 256)
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < nc; ++j) {
            const double *x = X + (size_t)j * ldx;
            double s = 0.0;
            for (int64_t p = rp[i]; p < rp[i + 1]; ++p) s += va[p] * x[ci[p]];
            Y[i + (size_t)j * ldy] = s;
        }
    }
}

void op_apply(const Op &A, int nc, const double *X, int ldx, double *Y, int ldy)
{
    if (A.identity) {
        for (int j = 0; j < nc; ++j)
            memcpy(Y + (size_t)j * ldy, X + (size_t)j * ldx, sizeof(double) * A.m);
    } else if (A.dense)
        gemm_nn(A.m, A.m, nc, 1.0, A.dense, A.ldd, X, ldx, 0.0, Y, ldy);
    else if (g_halo && !A.identity) {
        // [local rows ; ghost rows] of X, ghosts fetched from their owners (packed row-major per destination)
        const int64_t ns = (int64_t)g_send_rows.size(), ng = g_n_ghost;
        std::vector<double> send((size_t)std::max<int64_t>(ns, 1) * nc), recv((size_t)std::max<int64_t>(ng, 1) * nc);
        for (int64_t i = 0; i < ns; ++i)
            for (int j = 0; j < nc; ++j) send[(size_t)i * nc + j] = X[g_send_rows[i] + (size_t)j * ldx];
        g_halo(send.data(), recv.data(), nc);
        const int64_t me = A.m + ng;
        std::vector<double> Xe((size_t)me * nc);
        for (int j = 0; j < nc; ++j) {
            memcpy(&Xe[(size_t)j * me], X + (size_t)j * ldx, sizeof(double) * A.m);
            for (int64_t i = 0; i < ng; ++i) Xe[(size_t)j * me + A.m + i] = recv[(size_t)i * nc + j];
        }
        csr_spmm(A.m, A.rowptr, A.col, A.val, nc, Xe.data(), (int)me, Y, ldy);
    } else
        csr_spmm(A.m, A.rowptr, A.col, A.val, nc, X, ldx, Y, ldy);
}

} // namespace

// ----------------------------------------------------------------------------
// Projected Lyapunov solve: the role of SLICOT SB03MD as called at
// src/SlicotWrapper.cpp:38-41 with (dico,job,fact,trans) = ('C','X','N','T'):
// solves  A*X + X*A^T = scale*C  for symmetric C, X overwrites C.
// Bartels-Stewart: A = U S U^T (dgees), F = U^T C U, S*Y + Y*S^T = scale*F
// (dtrsyl 'N','T',+1), X = U Y U^T.  trans='N' solves A^T X + X A = scale*C.
// info = n+1 when dtrsyl reports perturbed (near-singular) diagonal blocks,
// which is the SLICOT convention the caller tolerates (src/LyapunovSolver.hpp:361).
// ----------------------------------------------------------------------------
extern "C" int orc_sb03md(char trans, int n, double *A, int lda, double *X, int ldx, double *scale)
{
    if (n < 1) return -1; // src/SlicotWrapper.cpp:12-16
    if (orc_lapack_init(nullptr)) return -100;
    std::vector<double> U((size_t)n * n), wr(n), wi(n), F((size_t)n * n), tmp((size_t)n * n);
    int sdim = 0, info = 0, lwork = -1;
    double wq = 0;
    g_lapack.dgees("V", "N", nullptr, &n, A, &lda, &sdim, wr.data(), wi.data(), U.data(), &n, &wq,
                   &lwork, nullptr, &info);
    lwork = (int)wq;
    std::vector<double> work(std::max(1, lwork));
    g_lapack.dgees("V", "N", nullptr, &n, A, &lda, &sdim, wr.data(), wi.data(), U.data(), &n,
                   work.data(), &lwork, nullptr, &info);
    if (info) return info;
    // F = U^T C U
    small_gemm('T', 'N', n, n, n, U.data(), n, X, ldx, tmp.data(), n);
    small_gemm('N', 'N', n, n, n, tmp.data(), n, U.data(), n, F.data(), n);
    int isgn = 1, tinfo = 0;
    const char *ta = (trans == 'T' || trans == 't') ? "N" : "T";
    const char *tb = (trans == 'T' || trans == 't') ? "T" : "N";
    g_lapack.dtrsyl(ta, tb, &isgn, &n, &n, A, &lda, A, &lda, F.data(), &n, scale, &tinfo);
    // X = U Y U^T
    small_gemm('N', 'N', n, n, n, U.data(), n, F.data(), n, tmp.data(), n);
    small_gemm('N', 'T', n, n, n, tmp.data(), n, U.data(), n, F.data(), n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) X[i + (size_t)j * ldx] = F[i + (size_t)j * n];
    return tinfo == 1 ? n + 1 : (tinfo < 0 ? tinfo : 0);
}

// dense_solve: solve A*X + X*A^T + B = 0.  src/LyapunovSolver.hpp:348-365
extern "C" int orc_dense_solve(int n, const double *A, int lda, const double *B, int ldb, double *X,
                               int ldx)
{
    std::vector<double> Ac((size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            Ac[i + (size_t)j * n] = A[i + (size_t)j * lda];
            X[i + (size_t)j * ldx] = B[i + (size_t)j * ldb];
        }
    double scale = 1.0;
    int info = orc_sb03md('T', n, Ac.data(), n, X, ldx, &scale);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) X[i + (size_t)j * ldx] *= -1.0;
    return info;
}

extern "C" int orc_dsyev(int n, double *a, int lda, double *w)
{
    if (orc_lapack_init(nullptr)) return -100;
    return sym_eig(n, a, lda, w);
}

namespace {

struct Params {
    int max_iter, expand_size, lanczos_iterations, restart_size, reduced_size, restart_iterations;
    double tol, restart_tolerance;
    int minimize_solution_space, restart_from_solution;
};

// ----------------------------------------------------------------------------
// resid_lanczos: src/LyapunovSolver.hpp:367-447.  Lanczos on the implicit
// R = AV*T*MV^T + MV*T*AV^T + B*B^T (MV == V in the reference's C++; the MV
// argument carries the generalized form of matlab/RAILSsolver.m:392).
// Returns the number of steps done; H is (max_iter+1)^2 col-major; Q m x (steps).
// ----------------------------------------------------------------------------
int resid_lanczos(int m, int k, const double *AV, int ldav, const double *V, int ldv,
                  const double *T, int ldt, const double *B, int ldb, int p, int max_iter, Rng &rng,
                  std::vector<double> &Q, std::vector<double> &H, std::vector<double> &evals,
                  std::vector<double> &evecs)
{
    int ldh = max_iter + 1;
    Q.assign((size_t)m * (max_iter + 1), 0.0);
    H.assign((size_t)ldh * ldh, 0.0);
    random_fill(rng, m, 1, Q.data(), m);
    scal(m, 1.0 / norm2(m, 1, Q.data(), m), Q.data());

    std::vector<double> Y(std::max(1, p)), Z(std::max(1, k)), Z2(std::max(1, k));
    double alpha = 0.0, beta = 0.0;
    int iter = 0;
    for (int i = 0; i < max_iter; ++i) {
        double *q = &Q[(size_t)iter * m];
        double *qn = &Q[(size_t)(iter + 1) * m];
        // Y = B^T q ; q+ = B Y                                   (:389-390)
        gemm_tn(m, p, 1, B, ldb, q, m, Y.data(), p);
        gemm_nn(m, p, 1, 1.0, B, ldb, Y.data(), p, 0.0, qn, m);
        // Z = T (V^T q); q+ += AV Z                               (:394-396)
        gemm_tn(m, k, 1, V, ldv, q, m, Z.data(), k);
        small_gemm('N', 'N', k, 1, k, T, ldt, Z.data(), k, Z2.data(), k);
        gemm_nn(m, k, 1, 1.0, AV, ldav, Z2.data(), k, 1.0, qn, m);
        // Z = T (AV^T q); q+ += V Z                               (:400-402)
        gemm_tn(m, k, 1, AV, ldav, q, m, Z.data(), k);
        small_gemm('N', 'N', k, 1, k, T, ldt, Z.data(), k, Z2.data(), k);
        gemm_nn(m, k, 1, 1.0, V, ldv, Z2.data(), k, 1.0, qn, m);
        // alpha = q+ . q                                          (:406-407)
        gemm_tn(m, 1, 1, qn, m, q, m, &alpha, 1);
        H[iter + (size_t)iter * ldh] = alpha;
        // q+ -= alpha q ; q+ -= beta q-                           (:411-413)
        {
            double ma = -alpha;
            gemm_nn(m, 1, 1, 1.0, q, m, &ma, 1, 1.0, qn, m);
            if (iter > 0) {
                double mb = -beta;
                gemm_nn(m, 1, 1, 1.0, &Q[(size_t)(iter - 1) * m], m, &mb, 1, 1.0, qn, m);
            }
        }
        beta = norm2(m, 1, qn, m); //                              (:417)
        if (beta < 1e-14) {        //                              (:419-426)
            iter++;
            break;
        }
        H[(iter + 1) + (size_t)iter * ldh] = beta;
        H[iter + (size_t)(iter + 1) * ldh] = beta;
        scal(m, 1.0 / beta, qn); //                                (:431)
        iter++;
    }
    // eigs of the leading iter x iter block, eigenvectors = Q v   (:437-443)
    std::vector<double> v((size_t)iter * iter);
    for (int j = 0; j < iter; ++j)
        for (int i2 = 0; i2 < iter; ++i2) v[i2 + (size_t)j * iter] = H[i2 + (size_t)j * ldh];
    evals.assign(iter, 0.0);
    sym_eig(iter, v.data(), iter, evals.data());
    evecs.assign((size_t)m * std::max(1, iter), 0.0);
    gemm_nn(m, iter, iter, 1.0, Q.data(), m, v.data(), iter, 0.0, evecs.data(), m);
    return iter;
}

// compute_restart_vectors: src/LyapunovSolver.hpp:449-482.  X is T.N() x idx.
int compute_restart_vectors(int k, const double *T, int ldt, int num, double tol,
                            std::vector<double> &X)
{
    std::vector<double> ev((size_t)k * k), w(k);
    for (int j = 0; j < k; ++j)
        for (int i = 0; i < k; ++i) ev[i + (size_t)j * k] = T[i + (size_t)j * ldt];
    sym_eig(k, ev.data(), k, w.data());
    num = (num > 0 ? num : k);
    std::vector<int> idx;
    find_largest(w.data(), k, num, idx);
    X.assign((size_t)k * num, 0.0);
    int kept = 0;
    for (int i = 0; i < num; ++i)
        if (std::abs(w[idx[i]]) > tol) {
            for (int j = 0; j < k; ++j) X[j + (size_t)i * k] = ev[j + (size_t)idx[i] * k];
            kept++;
        }
    X.resize((size_t)k * kept);
    return kept;
}

// X^T (S X) for a symmetric-size small matrix.  src/LyapunovSolver.hpp:286,293
void project_small(int k, int r, const double *X, std::vector<double> &S, int &lds)
{
    std::vector<double> t((size_t)k * r), o((size_t)r * r);
    small_gemm('N', 'N', k, r, k, S.data(), lds, X, k, t.data(), k);
    small_gemm('T', 'N', r, r, k, X, k, t.data(), k, o.data(), r);
    for (int j = 0; j < r; ++j)
        for (int i = 0; i < r; ++i) S[i + (size_t)j * lds] = o[i + (size_t)j * r];
}

} // namespace

// ----------------------------------------------------------------------------
// C ABI used by tests/, smoke() and bench.py's cpu_baseline.
// ----------------------------------------------------------------------------
static std::vector<double> g_trip_seconds;
static double g_solve_t0 = 0.0;

extern "C" {

struct orc_params {
    int max_iter;
    double tol;
    int expand_size;
    int lanczos_iterations;
    int restart_size;
    int reduced_size;
    int restart_iterations;
    double restart_tolerance;
    int minimize_solution_space;
    int restart_from_solution;
    int rng_mode;            // 0 = reference generator (std::rand seeded), 1 = counter-based
    unsigned long long seed; // counter mode seed
    unsigned long long stream0; // counter mode: first stream id
    long row0;               // global index of the first local row (counter mode)
    int verbose;
    int max_trips;           // >0: stop after this many loop trips (bounded CPU baseline)
};

void orc_default_params(orc_params *p)
{
    // src/LyapunovSolver.hpp:27-36
    p->max_iter = 1000;
    p->tol = 1e-3;
    p->expand_size = 3;
    p->lanczos_iterations = 10;
    p->restart_size = -1;
    p->reduced_size = -1;
    p->restart_iterations = 20;
    p->restart_tolerance = 1e-3 * 1e-3;
    p->minimize_solution_space = 1;
    p->restart_from_solution = 0;
    p->rng_mode = 1;
    p->seed = 1;
    p->stream0 = 0;
    p->row0 = 0;
    p->verbose = 0;
    p->max_trips = 0;
}

void orc_srand(unsigned s) { std::srand(s); }

// row-partitioned mode: hooks + ghost plan (null hooks switch it off)
void orc_set_partition(orc_allreduce_fn ar, orc_halo_fn halo, const int64_t *send_rows, int64_t n_send, int64_t n_ghost, int64_t m_global)
{
    g_allreduce = ar;
    g_halo = halo;
    g_send_rows.assign(send_rows, send_rows + (send_rows ? n_send : 0));
    g_n_ghost = n_ghost;
    g_m_global = m_global;
}
// seconds since the start of the last orc_solve at which each of its trips had its residual estimate (timing of single trips)
int orc_trip_seconds(double *out, int cap)
{
    int n = (int)g_trip_seconds.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = g_trip_seconds[i];
    return n;
}
int orc_num_threads() { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }

void orc_random(int mode, unsigned long long seed, unsigned long long stream, long row0, int m, int n,
                double *X, int ldx)
{
    Rng r{mode, seed, stream, row0};
    random_fill(r, m, n, X, ldx);
}

void orc_dot(int m, int a, int b, const double *X, int ldx, const double *Y, int ldy, double *C, int ldc)
{
    gemm_tn(m, a, b, X, ldx, Y, ldy, C, ldc);
}
void orc_panel_gemm(int m, int k, int r, double alpha, const double *X, int ldx, const double *C, int ldc,
                    double beta, double *Y, int ldy)
{
    gemm_nn(m, k, r, alpha, X, ldx, C, ldc, beta, Y, ldy);
}
double orc_norm2(int m, int n, const double *X, int ldx)
{
    orc_lapack_init(nullptr);
    return norm2(m, n, X, ldx);
}
void orc_orthogonalize(int m, double *V, int ldv, int from, int n)
{
    orc_lapack_init(nullptr);
    orthogonalize(m, V, ldv, from, n);
}
void orc_csr_spmm(int m, const int64_t *rp, const int32_t *ci, const double *va, int nc, const double *X,
                  int ldx, double *Y, int ldy)
{
    csr_spmm(m, rp, ci, va, nc, X, ldx, Y, ldy);
}
// operator apply in the current partition mode (ghost rows fetched through the halo hook)
void orc_op_apply(int m, const int64_t *rp, const int32_t *ci, const double *va, int nc, const double *X, int ldx, double *Y, int ldy)
{
    Op A;
    A.m = m;
    A.rowptr = rp;
    A.col = ci;
    A.val = va;
    op_apply(A, nc, X, ldx, Y, ldy);
}
void orc_find_largest(const double *vals, int n, int N, int *out)
{
    std::vector<int> idx;
    find_largest(vals, n, N, idx);
    for (int i = 0; i < N; ++i) out[i] = idx[i];
}

// resid_lanczos entry.  B m x p col-major.  Outputs: H (max_iter+1)^2 (ld = max_iter+1), evals[steps],
// evecs m x steps (ld m), Qout m x steps (may be null).  Returns steps.
int orc_resid_lanczos(int m, int k, const double *AV, int ldav, const double *V, int ldv, const double *T,
                      int ldt, const double *B, int ldb, int p, int max_iter, int rng_mode,
                      unsigned long long seed, unsigned long long stream, long row0, double *H,
                      double *evals, double *evecs, double *Qout)
{
    if (orc_lapack_init(nullptr)) return -100;
    Rng rng{rng_mode, seed, stream, row0};
    std::vector<double> Q, Hh, ev, evec;
    int it = resid_lanczos(m, k, AV, ldav, V, ldv, T, ldt, B, ldb, p, max_iter, rng, Q, Hh, ev, evec);
    memcpy(H, Hh.data(), sizeof(double) * Hh.size());
    for (int i = 0; i < it; ++i) evals[i] = ev[i];
    memcpy(evecs, evec.data(), sizeof(double) * (size_t)m * it);
    if (Qout) memcpy(Qout, Q.data(), sizeof(double) * (size_t)m * it);
    return it;
}

int orc_compute_restart_vectors(int k, const double *T, int ldt, int num, double tol, double *X)
{
    if (orc_lapack_init(nullptr)) return -100;
    std::vector<double> Xv;
    int kept = compute_restart_vectors(k, T, ldt, num, tol, Xv);
    memcpy(X, Xv.data(), sizeof(double) * Xv.size());
    return kept;
}

// ----------------------------------------------------------------------------
// solve: src/LyapunovSolver.hpp:100-346, standard form (M ignored, as in the
// reference's C++: M_ is stored at :26 and never read).  With Mop != null the
// generalized form of matlab/RAILSsolver.m:368-395,499-504 is used instead
// (MV, VMV, T from the Cholesky-reduced standard equation) -- parity unpinned.
//
// A is dense (Adense != null) or CSR.  B col-major m x p.  V col-major, ldv >= m,
// capacity vcap columns; *k_io = columns on entry (warm start) / exit.
// T col-major ldt >= vcap.  res_hist[i] = Lanczos estimate of trip i.
// Returns the reference's code: 0 converged, -1 not converged, 1 loop exhausted;
// 2 = stopped by max_trips.
// ----------------------------------------------------------------------------
int orc_solve(int m, const double *Adense, int lda, const int64_t *rp, const int32_t *ci, const double *va,
              const int64_t *mrp, const int32_t *mci, const double *mva, const double *B, int ldb, int p,
              const orc_params *prm, double *V, int ldv, int vcap, int *k_io, double *T, int ldt,
              double *res_hist, int hist_cap, int *trips_out)
{
    if (orc_lapack_init(nullptr)) return -100;
    g_trip_seconds.clear();
    g_solve_t0 = omp_get_wtime();
    Op A;
    A.m = m;
    A.dense = Adense;
    A.ldd = lda;
    A.rowptr = rp;
    A.col = ci;
    A.val = va;
    bool generalized = (mrp != nullptr);
    Op Mo;
    Mo.m = m;
    Mo.rowptr = mrp;
    Mo.col = mci;
    Mo.val = mva;
    Rng rng{prm->rng_mode, prm->seed, prm->stream0, prm->row0};

    const int n = g_m_global > 0 ? (int)g_m_global : m;
    int max_size = std::max(*k_io, std::min(prm->restart_size > 0 ? prm->restart_size : 100, n)); // :106
    int kV = *k_io;
    if (!prm->restart_from_solution) { // :108-115
        kV = 1;
        random_fill(rng, m, 1, V, ldv);
        orthogonalize(m, V, ldv, 0, 1);
    }
    // capacity bookkeeping: the caller gave vcap columns; the reference grows by 100 (:311-332)
    auto need_cap = [&](int cols) {
        if (cols > vcap) {
            fprintf(stderr, "rails_oracle: V capacity %d too small (need %d)\n", vcap, cols);
            return false;
        }
        return true;
    };
    if (!need_cap(max_size)) return -101;

    int cap = vcap;
    std::vector<double> AV((size_t)m * cap), MV;
    if (generalized) MV.assign((size_t)m * cap, 0.0);
    std::vector<double> BV((size_t)std::max(1, p) * cap);
    int ldS = cap;
    std::vector<double> VAV((size_t)cap * cap, 0.0), VBV((size_t)cap * cap, 0.0), VMV;
    if (generalized) VMV.assign((size_t)cap * cap, 0.0);
    int nAV = 0;
    // W = V (:123): a window [w0, w0+wn) of V's columns
    int w0 = 0, wn = kV;

    bool converged_previously = false;
    int previous_restart = 0;
    double r0 = norm2(m, p, B, ldb); // :134
    int ret = 1, trips = 0;

    std::vector<double> Q, H, evals, evecs;
    for (int iter = 0; iter < prm->max_iter; ++iter) {
        int N_V = kV;
        if (wn) { // :141-207
            double *W = V + (size_t)w0 * ldv;
            double *AW = &AV[(size_t)nAV * m];
            op_apply(A, wn, W, ldv, AW, m);                      // :146  AW = A*W (written in AV's tail)
            double *BW = &BV[(size_t)nAV * p];
            gemm_tn(m, p, wn, B, ldb, W, ldv, BW, p);            // :150  BW = B^T W
            double *MW = nullptr;
            if (generalized) {
                MW = &MV[(size_t)nAV * m];
                op_apply(Mo, wn, W, ldv, MW, m);                 // RAILSsolver.m:368-373
            }
            int s = nAV + wn;
            if (nAV > 0) { // :171-184
                std::vector<double> WAV((size_t)wn * nAV), WBV((size_t)wn * nAV);
                gemm_tn(m, wn, nAV, W, ldv, AV.data(), m, WAV.data(), wn);
                small_gemm('T', 'N', wn, nAV, p, BW, p, BV.data(), p, WBV.data(), wn);
                for (int i = 0; i < wn; ++i)
                    for (int j = 0; j < nAV; ++j) {
                        VAV[(i + nAV) + (size_t)j * ldS] = WAV[i + (size_t)j * wn];
                        VBV[(i + nAV) + (size_t)j * ldS] = WBV[i + (size_t)j * wn];
                        VBV[j + (size_t)(i + nAV) * ldS] = WBV[i + (size_t)j * wn];
                    }
                if (generalized) {
                    std::vector<double> WMV((size_t)wn * nAV);
                    gemm_tn(m, wn, nAV, W, ldv, MV.data(), m, WMV.data(), wn);
                    for (int i = 0; i < wn; ++i)
                        for (int j = 0; j < nAV; ++j)
                            VMV[(i + nAV) + (size_t)j * ldS] = WMV[i + (size_t)j * wn];
                }
            }
            { // :187-192  V^T AW
                std::vector<double> VAW((size_t)s * wn);
                gemm_tn(m, s, wn, V, ldv, AW, m, VAW.data(), s);
                for (int i = 0; i < s; ++i)
                    for (int j = 0; j < wn; ++j) VAV[i + (size_t)(j + nAV) * ldS] = VAW[i + (size_t)j * s];
                if (generalized) {
                    gemm_tn(m, s, wn, V, ldv, MW, m, VAW.data(), s);
                    for (int i = 0; i < s; ++i)
                        for (int j = 0; j < wn; ++j)
                            VMV[i + (size_t)(j + nAV) * ldS] = VAW[i + (size_t)j * s];
                }
            }
            { // :195-200
                std::vector<double> WBW((size_t)wn * wn);
                small_gemm('T', 'N', wn, wn, p, BW, p, BW, p, WBW.data(), wn);
                for (int i = 0; i < wn; ++i)
                    for (int j = 0; j < wn; ++j)
                        VBV[(i + nAV) + (size_t)(j + nAV) * ldS] = WBW[i + (size_t)j * wn];
            }
            nAV = s; // push_back :203-204 (AW, BW already in place)
        }

        // dense_solve(VAV, VBV, T) :209
        int k = nAV;
        if (!generalized) {
            orc_dense_solve(k, VAV.data(), ldS, VBV.data(), ldS, T, ldt);
        } else {
            // T = lyap(VAV, VBV, [], VMV): VAV T VMV^T + VMV T VAV^T + VBV = 0
            // (matlab/RAILSsolver.m:382, lyap.c:125-133).  Reduce with VMV = L L^T:
            // (L^-1 VAV L^-T) Tt + Tt (..)^T + L^-1 VBV L^-T = 0,  T = L^-T Tt L^-1.
            std::vector<double> L((size_t)k * k), Ai((size_t)k * k), Bi((size_t)k * k);
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i)
                    L[i + (size_t)j * k] = 0.5 * (VMV[i + (size_t)j * ldS] + VMV[j + (size_t)i * ldS]);
            int info = 0;
            g_lapack.dpotrf("L", &k, L.data(), &k, &info);
            double sgn = 1.0;
            if (info) { // negative definite M: the equation is invariant under (A, M) -> (-A, -M)
                sgn = -1.0;
                for (int j = 0; j < k; ++j)
                    for (int i = 0; i < k; ++i)
                        L[i + (size_t)j * k] = -0.5 * (VMV[i + (size_t)j * ldS] + VMV[j + (size_t)i * ldS]);
                g_lapack.dpotrf("L", &k, L.data(), &k, &info);
            }
            if (info) fprintf(stderr, "rails_oracle: VMV neither positive nor negative definite (dpotrf info %d)\n", info);
            auto lsolve_left = [&](std::vector<double> &X) { // X <- L^-1 X
                for (int j = 0; j < k; ++j)
                    for (int i = 0; i < k; ++i) {
                        double s = X[i + (size_t)j * k];
                        for (int l = 0; l < i; ++l) s -= L[i + (size_t)l * k] * X[l + (size_t)j * k];
                        X[i + (size_t)j * k] = s / L[i + (size_t)i * k];
                    }
            };
            auto lsolve_right = [&](std::vector<double> &X) { // X <- X L^-T
                for (int i = 0; i < k; ++i)
                    for (int j = 0; j < k; ++j) {
                        double s = X[i + (size_t)j * k];
                        for (int l = 0; l < j; ++l) s -= X[i + (size_t)l * k] * L[j + (size_t)l * k];
                        X[i + (size_t)j * k] = s / L[j + (size_t)j * k];
                    }
            };
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) {
                    Ai[i + (size_t)j * k] = sgn * VAV[i + (size_t)j * ldS];
                    Bi[i + (size_t)j * k] = VBV[i + (size_t)j * ldS];
                }
            lsolve_left(Ai);
            lsolve_right(Ai);
            lsolve_left(Bi);
            lsolve_right(Bi);
            std::vector<double> Tt((size_t)k * k);
            orc_dense_solve(k, Ai.data(), k, Bi.data(), k, Tt.data(), k);
            // T = L^-T Tt L^-1
            auto ltsolve_left = [&](std::vector<double> &X) { // X <- L^-T X
                for (int j = 0; j < k; ++j)
                    for (int i = k - 1; i >= 0; --i) {
                        double s = X[i + (size_t)j * k];
                        for (int l = i + 1; l < k; ++l) s -= L[l + (size_t)i * k] * X[l + (size_t)j * k];
                        X[i + (size_t)j * k] = s / L[i + (size_t)i * k];
                    }
            };
            auto lisolve_right = [&](std::vector<double> &X) { // X <- X L^-1
                for (int i = 0; i < k; ++i)
                    for (int j = k - 1; j >= 0; --j) {
                        double s = X[i + (size_t)j * k];
                        for (int l = j + 1; l < k; ++l) s -= X[i + (size_t)l * k] * L[l + (size_t)j * k];
                        X[i + (size_t)j * k] = s / L[j + (size_t)j * k];
                    }
            };
            ltsolve_left(Tt);
            lisolve_right(Tt);
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) T[i + (size_t)j * ldt] = Tt[i + (size_t)j * k];
        }

        // resid_lanczos :211-215
        int L = prm->lanczos_iterations;
        int steps = resid_lanczos(m, k, AV.data(), m, generalized ? MV.data() : V, generalized ? m : ldv, T,
                                  ldt, B, ldb, p, L, rng, Q, H, evals, evecs);
        double res = 0.0; // norm_inf of the eigenvalue column :217
        for (int i = 0; i < steps; ++i) res = std::max(res, std::fabs(evals[i]));
        if (trips < hist_cap) res_hist[trips] = res;
        trips++;
        if (prm->verbose)
            printf("Iteration %d. Estimate Lanczos, absolute: %.6e, relative: %.6e (V.N=%d)\n", iter + 1, res,
                   std::abs(res) / r0 / r0, kV);

        g_trip_seconds.push_back(omp_get_wtime() - g_solve_t0); // bench.py's cpu_baseline: when each trip's estimate was ready

        bool converged = std::abs(res) < prm->tol * r0 * r0; // :223
        if (converged || iter + 1 >= prm->max_iter || kV >= n) { // :224-242
            if (converged && prm->minimize_solution_space && !converged_previously)
                converged_previously = true;
            else {
                ret = converged ? 0 : -1;
                break;
            }
        }
        if (prm->max_trips > 0 && trips >= prm->max_trips) {
            ret = 2;
            break;
        }

        // restart :245-304
        if ((prm->restart_size > 0 && kV >= prm->restart_size) ||
            (prm->restart_iterations > 0 && iter - previous_restart >= prm->restart_iterations) || converged) {
            std::vector<double> X;
            int r = compute_restart_vectors(k, T, ldt, std::min(prm->reduced_size, kV), prm->restart_tolerance, X);
            std::vector<double> tmp((size_t)m * std::max(1, r));
            gemm_nn(m, kV, r, 1.0, V, ldv, X.data(), k, 0.0, tmp.data(), m); // V = V*X :265
            for (int j = 0; j < r; ++j) memcpy(V + (size_t)j * ldv, &tmp[(size_t)j * m], sizeof(double) * m);
            kV = r;
            if (prm->verbose) printf("Restarted with %d vectors\n", r);
            wn = 0; // W.resize(0) :284
            project_small(k, r, X.data(), VAV, ldS); // :286-288
            gemm_nn(m, k, r, 1.0, AV.data(), m, X.data(), k, 0.0, tmp.data(), m); // :290
            memcpy(AV.data(), tmp.data(), sizeof(double) * (size_t)m * r);
            project_small(k, r, X.data(), VBV, ldS); // :293-295
            {
                std::vector<double> t2((size_t)p * std::max(1, r));
                small_gemm('N', 'N', p, r, k, BV.data(), p, X.data(), k, t2.data(), p); // :297
                memcpy(BV.data(), t2.data(), sizeof(double) * (size_t)p * r);
            }
            if (generalized) { // RAILSsolver.m:499-504
                project_small(k, r, X.data(), VMV, ldS);
                gemm_nn(m, k, r, 1.0, MV.data(), m, X.data(), k, 0.0, tmp.data(), m);
                memcpy(MV.data(), tmp.data(), sizeof(double) * (size_t)m * r);
            }
            nAV = r;
            previous_restart = iter;
            continue;
        }

        // expand :306-342
        int expand = std::min(std::min(prm->expand_size, steps), (prm->restart_size > 0 ? prm->restart_size : n) - kV);
        if (kV + expand > max_size) max_size += 100; // :311-332 (capacity only)
        if (!need_cap(kV + expand)) {
            ret = -101;
            break;
        }
        std::vector<int> idx;
        find_largest(evals.data(), steps, expand, idx);
        for (int i = 0; i < expand; ++i)
            memcpy(V + (size_t)(kV + i) * ldv, &evecs[(size_t)idx[i] * m], sizeof(double) * m);
        kV += expand;
        orthogonalize(m, V, ldv, N_V, kV); // watermark: only the new columns :340
        w0 = N_V;
        wn = expand; // :342
    }
    *k_io = kV;
    if (trips_out) *trips_out = trips;
    return ret;
}


// ----------------------------------------------------------------------------
// Interpreter of the sweep-SpMM schedule (rails_amd/csrc/sweep_plan.h): executes, on the CPU and in the order the HIP
// kernel does, the program of every (part, phase, chunk, wave) and checks the two things the kernel relies on -- every
// ring row a trip reads holds the X row the schedule meant (it arrived, and is not the segment being refilled), and every
// row of Y is written exactly once.  Y = A X then follows from the schedule alone, so comparing it with orc_csr_spmm
// pins the planner without a GPU.  Replaces nothing in the reference (`A_ * W`, src/LyapunovSolver.hpp:146, is the
// product being scheduled).  X, Y column-major.  Returns 0, or -1 ring row not readable, -2 row written twice,
// -3 row never written, -4 trips left in a program's stream.
// ----------------------------------------------------------------------------
int orc_sweep_interpret(const int64_t *iinfo, const int64_t *part_row0, const int64_t *sweep0, const int32_t *nsteps,
                        const int64_t *hdr_off, const int64_t *batch_off, const int64_t *flush_off, const uint32_t *codes,
                        const double *vals, const uint16_t *offs, const int32_t *flush_rows, int64_t m, int64_t ncols, int n_chunks,
                        const double *X, int64_t ldx, double *Y, int64_t ldy)
{
    const int W = (int)iinfo[0], G = (int)iinfo[1], SEG = (int)iinfo[2], NSEG = (int)iinfo[3], parts = (int)iinfo[4], P = (int)iinfo[5];
    const int CPS = (int)iinfo[6]; // entries per (program, step) record
    const int AHEAD = (int)iinfo[12]; // segments being refilled at any time
    const int SLOTS = (int)iinfo[11]; // rows of a group in one wave (16: a slot is a quad of lanes)
    const int ENTRY_TRIPS = iinfo[13] == 2 ? 2 : 4; // trips per schedule entry (a whole unit of the kernel's code, or half of one)
    std::vector<int> written((size_t)m * n_chunks, 0);
    int rc = 0;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int x = 0; x < parts; ++x)
        for (int q = 0; q < n_chunks; ++q) {
            std::vector<double> acc((size_t)G * SLOTS * 16);
            for (int ph = 0; ph < P; ++ph)
                for (int w = 0; w < W; ++w) {
                    const int64_t prog = ((int64_t)x * P + ph) * W + w;
                    std::fill(acc.begin(), acc.end(), 0.0);
                    int64_t trip = 0, fl = 0;
                    for (int k = 0; k < nsteps[x]; ++k)
                        for (int ci = 1; ci <= (int)(codes[hdr_off[prog] + (int64_t)k * CPS] & 0xff); ++ci) {
                            const uint32_t *rec = codes + hdr_off[prog] + (int64_t)k * CPS;
                            const int n = (int)(rec[0] & 0xff);
                            const uint32_t code = rec[ci];
                            const int g = (int)(code & 0xff) / 8;
                            const int T = (code & 0x200) ? 0 : ENTRY_TRIPS;
                            if (g >= G || ci >= CPS || (code & 7)) {
#pragma omp atomic write
                                rc = -1;
                                continue;
                            }
                            // the flags the kernel branches on: last entry of the step; the next entry (for the header: the first) has no trips
                            const bool last = ci == n, next_none = ci < n && (rec[ci + 1] & 0x200);
                            if (((code & 0x400) != 0) != last || ((code & 0x800) != 0) != next_none ||
                                (ci == 1 && ((rec[0] & 0x800) != 0) != ((code & 0x200) != 0)) || ((code & 0x200) && !(code & 0x100))) {
#pragma omp atomic write
                                rc = -5;
                                continue;
                            }
                            for (int t = 0; t < T; ++t, ++trip) {
                                const int64_t b = batch_off[prog] + trip / 16;
                                const int tt = (int)(trip % 16), unit = tt / 4, ql = tt % 4;
                                for (int s = 0; s < SLOTS; ++s) {
                                    const size_t lane = (size_t)s * 4 + ql;
                                    const double v = vals[(size_t)b * 256 + (size_t)(unit / 2) * 128 + lane * 2 + (unit % 2)];
                                    const int o = offs[(size_t)b * 256 + lane * 4 + unit];
                                    const int seg = o / SEG;
                                    const int back = ((k - seg) % NSEG + NSEG) % NSEG; // steps since that segment was filled
                                    const int kk = k - back;
                                    if (seg >= NSEG || kk < 0 || back > NSEG - 1 - AHEAD) {
#pragma omp atomic write
                                        rc = -1;
                                        continue;
                                    }
                                    int64_t xrow = sweep0[x] + (int64_t)kk * SEG + o % SEG;
                                    xrow = xrow < 0 ? 0 : (xrow >= ncols ? ncols - 1 : xrow);
                                    double *a = &acc[((size_t)g * SLOTS + s) * 16];
                                    for (int c = 0; c < 16; ++c) a[c] += v * X[xrow + (int64_t)(q * 16 + c) * ldx];
                                }
                            }
                            if (code & 0x100) {
                                const int64_t row0 = flush_rows[flush_off[prog] + fl++];
                                if ((int64_t)(code >> 12) * 16 + part_row0[x] != row0) { // the kernel takes the rows from the entry
#pragma omp atomic write
                                    rc = -5;
                                }
                                for (int s = 0; s < SLOTS; ++s) {
                                    const int64_t row = row0 + s;
                                    double *a = &acc[((size_t)g * SLOTS + s) * 16];
                                    if (row < part_row0[x + 1]) {
                                        for (int c = 0; c < 16; ++c) Y[row + (int64_t)(q * 16 + c) * ldy] = a[c];
#pragma omp atomic
                                        written[(size_t)row * n_chunks + q]++;
                                    }
                                    for (int c = 0; c < 16; ++c) a[c] = 0.0;
                                }
                            }
                        }
                    // the wave's stream must be used up exactly: the next program starts at the next batch boundary
                    const int64_t nprog = (int64_t)parts * P * W;
                    if (prog + 1 < nprog && batch_off[prog + 1] != batch_off[prog] + (trip + 15) / 16) {
#pragma omp atomic write
                        rc = -4;
                    }
                }
        }
    if (rc) return rc;
    for (size_t i = 0; i < written.size(); ++i) {
        if (written[i] > 1) return -2;
        if (written[i] < 1) return -3;
    }
    return 0;
}

} // extern "C"
