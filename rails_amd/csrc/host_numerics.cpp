// host_numerics.cpp -- the small dense solves that stay on the host (north star): thin C ABI with
// the argument meaning of the reference's shims
//   rails_sb03md  <- RAILS::sb03md      (src/SlicotWrapper.hpp:14-16, src/SlicotWrapper.cpp:8-49)
//   rails_dsyev   <- LapackWrapper::DSYEV  (src/LapackWrapper.cpp:20-39)
//   rails_dsteqr  <- LapackWrapper::DSTEQR (src/LapackWrapper.cpp:12-18)
// LAPACK is resolved at run time with dlopen (the GPU box carries no system LAPACK; this image
// ships scipy's OpenBLAS and MKL).  SLICOT is not available anywhere in the image, so the
// continuous Lyapunov solve is a Bartels-Stewart implementation on dgees + dtrsyl.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <utility>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rails_hip.h"

void rails_set_error(const char *fmt, ...);

namespace {

extern "C" {
typedef int (*lp_select2)(const double *, const double *);
typedef void (*lp_dsyev)(const char *, const char *, const int *, double *, const int *, double *, double *, const int *, int *);
typedef void (*lp_dsyevd)(const char *, const char *, const int *, double *, const int *, double *, double *, const int *, int *, const int *, int *);
typedef void (*lp_dsteqr)(const char *, const int *, double *, double *, double *, const int *, double *, int *);
typedef void (*lp_dgees)(const char *, const char *, lp_select2, const int *, double *, const int *, int *, double *, double *,
                         double *, const int *, double *, const int *, int *, int *);
typedef void (*lp_dtrsyl)(const char *, const char *, const int *, const int *, const int *, const double *, const int *,
                          const double *, const int *, double *, const int *, double *, int *);
typedef void (*lp_dtrsyl3)(const char *, const char *, const int *, const int *, const int *, const double *, const int *,
                           const double *, const int *, double *, const int *, double *, int *, const int *, double *, const int *, int *);
typedef void (*lp_dpotrf)(const char *, const int *, double *, const int *, int *);
typedef void (*lp_dgeqp3)(const int *, const int *, double *, const int *, int *, double *, double *, const int *, int *);
typedef void (*lp_dorgqr)(const int *, const int *, const int *, double *, const int *, const double *, double *, const int *, int *);
typedef void (*lp_dpstrf)(const char *, const int *, double *, const int *, int *, int *, const double *, double *, int *);
typedef void (*lp_dgemm)(const char *, const char *, const int *, const int *, const int *, const double *, const double *,
                         const int *, const double *, const int *, const double *, double *, const int *);
typedef void (*lp_dtrsm)(const char *, const char *, const char *, const char *, const int *, const int *, const double *, const double *,
                         const int *, double *, const int *);
typedef void (*lp_dsyrk)(const char *, const char *, const int *, const int *, const double *, const double *, const int *, const double *,
                         double *, const int *);
typedef void (*lp_dgetrf)(const int *, const int *, double *, const int *, int *, int *);
typedef void (*lp_dgetrs)(const char *, const int *, const int *, const double *, const int *, const int *, double *, const int *, int *);
typedef void (*lp_dlaswp)(const int *, double *, const int *, const int *, const int *, const int *, const int *);
typedef void (*lp_dgetri)(const int *, double *, const int *, const int *, double *, const int *, int *);
}

struct HostLapack {
    void *handle = nullptr;
    std::string path;
    lp_dsyev dsyev = nullptr;
    lp_dsyevd dsyevd = nullptr; // optional
    lp_dsteqr dsteqr = nullptr;
    lp_dgees dgees = nullptr;
    lp_dtrsyl dtrsyl = nullptr;
    lp_dtrsyl3 dtrsyl3 = nullptr; // optional: blocked (level-3) triangular Sylvester solver of LAPACK >= 3.11
    lp_dpotrf dpotrf = nullptr;
    lp_dpstrf dpstrf = nullptr; // optional
    lp_dgeqp3 dgeqp3 = nullptr; // optional (rails_range_basis)
    lp_dorgqr dorgqr = nullptr;
    lp_dgemm dgemm = nullptr;
    lp_dtrsm dtrsm = nullptr; // optional: the generalized projected solve falls back to loops
    lp_dsyrk dsyrk = nullptr;   // optional: X = Z Z' of the factored ADI route
    lp_dgetrf dgetrf = nullptr; // optional: the squared-Smith fast path of rails_sb03md
    lp_dgetrs dgetrs = nullptr;
    lp_dlaswp dlaswp = nullptr; // optional: the bordered LU update of the factored ADI route
    lp_dgetri dgetri = nullptr; // optional: explicit (M - p I)^-1 of the factored ADI route
} g_lp;
std::mutex g_lp_mutex;

void *lookup(void *h, const char *base)
{
    static const char *prefixes[] = {"scipy_", "", nullptr};
    for (int i = 0; prefixes[i]; ++i) {
        std::string s = std::string(prefixes[i]) + base;
        if (void *p = dlsym(h, s.c_str())) return p;
    }
    return nullptr;
}

bool try_open(const std::string &path)
{
    if (path.empty()) return false;
    // OpenBLAS starts its worker threads when the library is loaded -- one per visible CPU (up to its build limit), each
    // spinning for a while before it goes to sleep.  On a box whose CPU quota (cgroup) is far below its CPU count those ~60
    // spinning threads use up the quota of a scheduling period in a few milliseconds and the WHOLE process is throttled for the
    // rest of it: one 40-90 ms trip some 100 ms after the first host LAPACK call of a solve (BENCH_r01: "slowest 42 ms").
    // openblas_set_num_threads() after the fact does not stop them, so the count is given through the environment for the
    // duration of the dlopen (and restored: other BLAS users of the process keep their own setting).
    // setenv is not safe against getenv in other threads of the process, so the window is kept as small as it can be: nothing is touched
    // when the library is in the process already (its threads exist: RTLD_NOLOAD probe), a variable the application has set itself is
    // left alone (its choice stands -- `OPENBLAS_NUM_THREADS=1 python bench.py` never gets here), and what is set is set once, under
    // g_lp_mutex, before the dlopen, and taken back right after.  An application with threads of its own that cannot tolerate even
    // that calls rails_host_lapack_init (or creates its first context) before it starts them, or exports the variable.
    int want_threads = 1;
    if (const char *e = getenv("RAILS_LAPACK_THREADS")) want_threads = atoi(e) > 0 ? atoi(e) : 1;
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) {
        static const char *thread_vars[] = {"OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", nullptr};
        bool mine[2] = {false, false};
        for (int i = 0; thread_vars[i]; ++i)
            if (!getenv(thread_vars[i])) {
                mine[i] = true;
                setenv(thread_vars[i], std::to_string(want_threads).c_str(), 1);
            }
        h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        for (int i = 0; thread_vars[i]; ++i)
            if (mine[i]) unsetenv(thread_vars[i]);
    }
    if (!h) return false;
    HostLapack L;
    L.handle = h;
    L.path = path;
    L.dsyev = (lp_dsyev)lookup(h, "dsyev_");
    L.dsyevd = (lp_dsyevd)lookup(h, "dsyevd_");
    L.dsteqr = (lp_dsteqr)lookup(h, "dsteqr_");
    L.dgees = (lp_dgees)lookup(h, "dgees_");
    L.dtrsyl = (lp_dtrsyl)lookup(h, "dtrsyl_");
    L.dtrsyl3 = (lp_dtrsyl3)lookup(h, "dtrsyl3_");
    L.dpotrf = (lp_dpotrf)lookup(h, "dpotrf_");
    L.dgemm = (lp_dgemm)lookup(h, "dgemm_");
    L.dtrsm = (lp_dtrsm)lookup(h, "dtrsm_");
    L.dsyrk = (lp_dsyrk)lookup(h, "dsyrk_");
    L.dgetrf = (lp_dgetrf)lookup(h, "dgetrf_");
    L.dgetrs = (lp_dgetrs)lookup(h, "dgetrs_");
    L.dlaswp = (lp_dlaswp)lookup(h, "dlaswp_");
    L.dgetri = (lp_dgetri)lookup(h, "dgetri_");
    L.dpstrf = (lp_dpstrf)lookup(h, "dpstrf_");
    L.dgeqp3 = (lp_dgeqp3)lookup(h, "dgeqp3_");
    L.dorgqr = (lp_dorgqr)lookup(h, "dorgqr_");
    if (!L.dsyev || !L.dsteqr || !L.dgees || !L.dtrsyl || !L.dpotrf || !L.dgemm) {
        dlclose(h);
        return false;
    }
    // The projected matrices are a few hundred rows: threaded BLAS only adds fork/join latency to the thousands of
    // small calls inside dgees/dtrsyl.  RAILS_LAPACK_THREADS overrides (default 1).
    typedef void (*setthreads_t)(int);
    setthreads_t st = (setthreads_t)dlsym(h, "scipy_openblas_set_num_threads");
    if (!st) st = (setthreads_t)dlsym(h, "openblas_set_num_threads");
    if (!st) st = (setthreads_t)dlsym(h, "MKL_Set_Num_Threads");
    int nt = 1;
    if (const char *e = getenv("RAILS_LAPACK_THREADS")) nt = atoi(e) > 0 ? atoi(e) : 1;
    if (st) st(nt);
    g_lp = L;
    return true;
}

void gemm(char ta, char tb, int m, int n, int k, const double *A, int lda, const double *B, int ldb, double *C, int ldc)
{
    const double one = 1.0, zero = 0.0;
    g_lp.dgemm(&ta, &tb, &m, &n, &k, &one, A, &lda, B, &ldb, &zero, C, &ldc);
}

} // namespace

extern "C" int rails_host_lapack_init(const char *path)
{
    std::lock_guard<std::mutex> lock(g_lp_mutex);
    if (g_lp.handle) return RAILS_OK;
    if (path && *path && try_open(path)) return RAILS_OK;
    if (const char *env = getenv("RAILS_LAPACK_LIB"))
        if (try_open(env)) return RAILS_OK;
    // scipy's bundled OpenBLAS (hashed file name): scan the usual site-packages locations
    static const char *dirs[] = {"/usr/local/lib/python3.10/dist-packages/scipy.libs", "/usr/lib/python3/dist-packages/scipy.libs",
                                 nullptr};
    for (int i = 0; dirs[i]; ++i) {
        std::string cmd = std::string("ls ") + dirs[i] + "/libscipy_openblas*.so 2>/dev/null";
        if (FILE *f = popen(cmd.c_str(), "r")) {
            char buf[1024];
            std::vector<std::string> found;
            while (fgets(buf, sizeof(buf), f)) {
                std::string s(buf);
                while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
                found.push_back(s);
            }
            pclose(f);
            for (auto &s : found)
                if (try_open(s)) return RAILS_OK;
        }
    }
    static const char *cands[] = {"libopenblas.so.0", "libopenblas.so", "liblapack.so.3", "liblapack.so", "/opt/conda/lib/libmkl_rt.so",
                                  "libmkl_rt.so", nullptr};
    for (int i = 0; cands[i]; ++i)
        if (try_open(cands[i])) return RAILS_OK;
    rails_set_error("no host LAPACK found: set RAILS_LAPACK_LIB to a library exporting dsyev_/dgees_/dtrsyl_/dpotrf_/dgemm_");
    return RAILS_ELAPACK;
}

extern "C" const char *rails_host_lapack_path(void) { return g_lp.path.c_str(); }

// LAPACK's eigen-solvers may not return (or may index out of bounds) on NaN / Inf input: refuse it with an error code instead.
static bool all_finite(int rows, int cols, const double *a, int lda)
{
    for (int j = 0; j < cols; ++j)
        for (int i = 0; i < rows; ++i)
            if (!std::isfinite(a[i + (size_t)j * lda])) return false;
    return true;
}

extern "C" void rails_dsyev(char jobz, char uplo, int n, double *a, int lda, double *w, int *info)
{
    if (rails_host_lapack_init(nullptr) != RAILS_OK) {
        *info = -100;
        return;
    }
    if (n <= 0) {
        *info = 0;
        return;
    }
    if (!all_finite(n, n, a, lda)) {
        fprintf(stderr, "rails_dsyev: the matrix holds NaN or Inf entries\n");
        *info = n + 1;
        return;
    }
    if (g_lp.dsyevd && n >= 64 && (jobz == 'V' || jobz == 'v')) { // divide and conquer: 2-3x faster at the restart sizes (eig(T), k = 200)
        int lwork = -1, liwork = -1, iq = 0;
        double wq = 0.0;
        g_lp.dsyevd(&jobz, &uplo, &n, a, &lda, w, &wq, &lwork, &iq, &liwork, info);
        if (*info == 0) {
            lwork = (int)wq + 1;
            liwork = iq + 1;
            std::vector<double> work((size_t)lwork);
            std::vector<int> iwork((size_t)liwork);
            g_lp.dsyevd(&jobz, &uplo, &n, a, &lda, w, work.data(), &lwork, iwork.data(), &liwork, info);
            return;
        }
    }
    int lwork = -1;
    double wq = 0.0;
    g_lp.dsyev(&jobz, &uplo, &n, a, &lda, w, &wq, &lwork, info); // workspace query, as the reference does
    if (*info) return;
    lwork = (int)wq;
    std::vector<double> work((size_t)std::max(1, lwork));
    g_lp.dsyev(&jobz, &uplo, &n, a, &lda, w, work.data(), &lwork, info);
}

extern "C" void rails_dsteqr(char compz, int n, double *d, double *e, double *z, int ldz, double *work, int *info)
{
    if (rails_host_lapack_init(nullptr) != RAILS_OK) {
        *info = -100;
        return;
    }
    g_lp.dsteqr(&compz, &n, d, e, z, &ldz, work, info);
}

// B <- alpha * op(A)^-1 B (side 'L') or alpha * B op(A)^-1 (side 'R'), A triangular: BLAS DTRSM, or plain loops when the library has none
extern "C" void rails_dtrsm(char side, char uplo, char transa, char diag, int m, int n, double alpha, const double *A, int lda, double *B, int ldb)
{
    if (m <= 0 || n <= 0) return;
    const char *force_loops = getenv("RAILS_DTRSM_LOOPS"); // tests: exercise the fallback
    if (!(force_loops && atoi(force_loops) != 0) && rails_host_lapack_init(nullptr) == RAILS_OK && g_lp.dtrsm) {
        g_lp.dtrsm(&side, &uplo, &transa, &diag, &m, &n, &alpha, A, &lda, B, &ldb);
        return;
    }
    const bool left = side == 'L' || side == 'l', upper = uplo == 'U' || uplo == 'u', trans = !(transa == 'N' || transa == 'n'),
               unit = diag == 'U' || diag == 'u';
    const int na = left ? m : n;
    auto a = [&](int i, int j) { return trans ? A[j + (size_t)i * lda] : A[i + (size_t)j * lda]; }; // op(A)(i, j)
    const bool op_upper = upper != trans;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) B[i + (size_t)j * ldb] *= alpha;
    if (left) { // solve op(A) X = B column by column
        for (int j = 0; j < n; ++j) {
            double *b = B + (size_t)j * ldb;
            if (op_upper)
                for (int i = na - 1; i >= 0; --i) {
                    double s = b[i];
                    for (int l = i + 1; l < na; ++l) s -= a(i, l) * b[l];
                    b[i] = unit ? s : s / a(i, i);
                }
            else
                for (int i = 0; i < na; ++i) {
                    double s = b[i];
                    for (int l = 0; l < i; ++l) s -= a(i, l) * b[l];
                    b[i] = unit ? s : s / a(i, i);
                }
        }
    } else { // solve X op(A) = B row by row
        for (int i = 0; i < m; ++i) {
            if (op_upper)
                for (int j = 0; j < na; ++j) {
                    double s = B[i + (size_t)j * ldb];
                    for (int l = 0; l < j; ++l) s -= B[i + (size_t)l * ldb] * a(l, j);
                    B[i + (size_t)j * ldb] = unit ? s : s / a(j, j);
                }
            else
                for (int j = na - 1; j >= 0; --j) {
                    double s = B[i + (size_t)j * ldb];
                    for (int l = j + 1; l < na; ++l) s -= B[i + (size_t)l * ldb] * a(l, j);
                    B[i + (size_t)j * ldb] = unit ? s : s / a(j, j);
                }
        }
    }
}

extern "C" void rails_dgemm(char ta, char tb, int m, int n, int k, double alpha, const double *A, int lda, const double *B, int ldb,
                            double beta, double *C, int ldc)
{
    if (rails_host_lapack_init(nullptr) != RAILS_OK) return;
    g_lp.dgemm(&ta, &tb, &m, &n, &k, &alpha, A, &lda, B, &ldb, &beta, C, &ldc);
}

extern "C" void rails_dpotrf(char uplo, int n, double *a, int lda, int *info)
{
    if (rails_host_lapack_init(nullptr) != RAILS_OK) {
        *info = -100;
        return;
    }
    g_lp.dpotrf(&uplo, &n, a, &lda, info);
}

// Cholesky with complete pivoting of a positive semi-definite matrix (LAPACK dpstrf): P' A P = R' R, stops at the first
// pivot <= tol and returns the rank found.  piv is 0-based here.  Falls back to an unblocked outer-product form when the
// host LAPACK does not export dpstrf.
extern "C" void rails_dpstrf(char uplo, int n, double *a, int lda, int *piv, int *rank, double tol, int *info)
{
    if (rails_host_lapack_init(nullptr) != RAILS_OK) {
        *info = -100;
        return;
    }
    if (g_lp.dpstrf) {
        std::vector<double> work((size_t)2 * (n > 0 ? n : 1));
        g_lp.dpstrf(&uplo, &n, a, &lda, piv, rank, &tol, work.data(), info);
        for (int i = 0; i < n; ++i) piv[i] -= 1;
        return;
    }
    // upper form: a(i, j), i <= j
    *info = 0;
    if (uplo != 'U' && uplo != 'u') {
        *info = -1;
        return;
    }
    for (int i = 0; i < n; ++i) piv[i] = i;
    int r = 0;
    for (; r < n; ++r) {
        int q = r;
        for (int j = r + 1; j < n; ++j)
            if (a[j + (size_t)j * lda] > a[q + (size_t)q * lda]) q = j;
        if (!(a[q + (size_t)q * lda] > tol)) break;
        if (q != r) { // symmetric swap of r and q in the upper triangle
            std::swap(piv[r], piv[q]);
            std::swap(a[r + (size_t)r * lda], a[q + (size_t)q * lda]);
            for (int i = 0; i < r; ++i) std::swap(a[i + (size_t)r * lda], a[i + (size_t)q * lda]);
            for (int j = q + 1; j < n; ++j) std::swap(a[r + (size_t)j * lda], a[q + (size_t)j * lda]);
            for (int i = r + 1; i < q; ++i) std::swap(a[r + (size_t)i * lda], a[i + (size_t)q * lda]);
        }
        double d = std::sqrt(a[r + (size_t)r * lda]);
        a[r + (size_t)r * lda] = d;
        for (int j = r + 1; j < n; ++j) a[r + (size_t)j * lda] /= d;
        for (int j = r + 1; j < n; ++j) {
            double f = a[r + (size_t)j * lda];
            for (int i = r + 1; i <= j; ++i) a[i + (size_t)j * lda] -= a[r + (size_t)i * lda] * f;
        }
    }
    *rank = r;
    *info = r < n ? 1 : 0;
}

// Orthonormal basis of the column space of A (m x n, column-major, overwritten): Householder QR with column pivoting (dgeqp3),
// numerical rank = pivots with |r_ii| > tol * |r_11|, Q (m x rank) formed with dorgqr into q (ldq >= m).  Used by the
// coordinate-space backend (rails/SubspaceWrappers.hpp) to compress its basis after a restart.
extern "C" void rails_range_basis(int m, int n, double *a, int lda, double tol, double *q, int ldq, int *rank, int *info)
{
    *rank = 0;
    if (rails_host_lapack_init(nullptr) != RAILS_OK || !g_lp.dgeqp3 || !g_lp.dorgqr) {
        *info = -100;
        return;
    }
    if (m <= 0 || n <= 0) {
        *info = 0;
        return;
    }
    const int kmin = m < n ? m : n;
    std::vector<int> jpvt((size_t)n, 0);
    std::vector<double> tau((size_t)kmin);
    double wq = 0.0;
    int lwork = -1;
    g_lp.dgeqp3(&m, &n, a, &lda, jpvt.data(), tau.data(), &wq, &lwork, info);
    if (*info != 0) return;
    lwork = (int)wq + 1;
    std::vector<double> work((size_t)lwork);
    g_lp.dgeqp3(&m, &n, a, &lda, jpvt.data(), tau.data(), work.data(), &lwork, info);
    if (*info != 0) return;
    const double r11 = std::fabs(a[0]);
    int r = 0;
    while (r < kmin && std::fabs(a[r + (size_t)r * lda]) > tol * r11 && r11 > 0.0) ++r;
    *rank = r;
    if (r == 0) return;
    // the reflectors live below the diagonal of a's first r columns: copy them to q and expand
    for (int j = 0; j < r; ++j)
        for (int i = 0; i < m; ++i) q[i + (size_t)j * ldq] = a[i + (size_t)j * lda];
    lwork = -1;
    g_lp.dorgqr(&m, &r, &r, q, &ldq, tau.data(), &wq, &lwork, info);
    if (*info != 0) return;
    lwork = (int)wq + 1;
    work.resize((size_t)lwork);
    g_lp.dorgqr(&m, &r, &r, q, &ldq, tau.data(), work.data(), &lwork, info);
}

static int &sb03md_smith_pause()
{
    static thread_local int pause = 0; // calls of this thread that skip the Smith attempt after one that did not apply
    return pause;
}
static int &sb03md_factored_pause()
{
    static thread_local int pause = 0; // the factored (ADI) form is tried first; where it gives up it rests on its own count
    return pause;
}
extern "C" void rails_sb03md_set_pause(int calls) { sb03md_smith_pause() = sb03md_factored_pause() = calls > 0 ? calls : 0; }
static std::atomic<long> g_sb03md_smith{0}, g_sb03md_schur{0};
extern "C" void rails_sb03md_counts(long *smith, long *schur)
{
    if (smith) *smith = g_sb03md_smith.load();
    if (schur) *schur = g_sb03md_schur.load();
}

// Squared Smith iteration for M X + X M' = C (M stable, n >= 32): with p > 0, S = M - p I,
//     X = Md X Md' + Cd,   Md = S^-1 (M + p I) = I + 2 p S^-1,   Cd = -2 p S^-1 C S^-T,
// and X = sum_j Md^j Cd Md'^j is summed by squaring: Y <- Y + Ak Y Ak', Ak <- Ak^2.  With p = -trace(M)/n the spectral radius of
// Md is small whenever the spectrum of M is clustered relative to its distance from the imaginary axis -- the projections V'AV
// of diagonally dominant operators (the benchmark's: rho ~ 0.1-0.25, 4-6 squarings).  All level-3 BLAS: ~40 n^3 flops at GEMM
// speed against the ~25 n^3 flops of the Hessenberg QR sweeps at a fifth of it (2.5 vs 6.9 ms at n = 160).  The result is
// VERIFIED -- ||M X + X M' - C||_F <= 2e-15 (2 ||M|| ||X|| + ||C||), the level Bartels-Stewart reaches -- and anything else
// (slow convergence, an unstable M, a failed check) returns false: the caller then runs Bartels-Stewart as before.
static bool smith_lyapunov(bool tr, int n, const double *A, int lda, double *X, int ldx)
{
    if (!g_lp.dgetrf || !g_lp.dgetrs) return false;
    const size_t nn = (size_t)n * n;
    std::vector<double> M(nn), S(nn), Ak(nn), Y(nn), T1(nn), T2(nn), C(nn);
    double trace = 0.0;
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) {
            M[i + (size_t)j * n] = tr ? A[i + (size_t)j * lda] : A[j + (size_t)i * lda];
            C[i + (size_t)j * n] = 0.5 * (X[i + (size_t)j * ldx] + X[j + (size_t)i * ldx]);
        }
        trace += A[j + (size_t)j * lda];
    }
    const double p = -trace / n;
    if (!(p > 0.0) || !std::isfinite(p)) return false;
    S = M;
    for (int j = 0; j < n; ++j) S[j + (size_t)j * n] -= p;
    std::vector<int> ipiv(n);
    int info = 0;
    g_lp.dgetrf(&n, &n, S.data(), &n, ipiv.data(), &info);
    if (info != 0) return false;
    std::fill(Ak.begin(), Ak.end(), 0.0);
    for (int j = 0; j < n; ++j) Ak[j + (size_t)j * n] = 1.0;
    const char N = 'N';
    g_lp.dgetrs(&N, &n, &n, S.data(), &n, ipiv.data(), Ak.data(), &n, &info); // Ak = S^-1
    if (info != 0) return false;
    // Y0 = Cd = -2p S^-1 C S^-T
    gemm('N', 'N', n, n, n, Ak.data(), n, C.data(), n, T1.data(), n);
    gemm('N', 'T', n, n, n, T1.data(), n, Ak.data(), n, Y.data(), n);
    for (size_t q = 0; q < nn; ++q) Y[q] *= -2.0 * p;
    for (size_t q = 0; q < nn; ++q) Ak[q] *= 2.0 * p; // Ak = Md = I + 2p S^-1
    for (int j = 0; j < n; ++j) Ak[j + (size_t)j * n] += 1.0;
    auto fro = [&](std::vector<double> const &Z) {
        double s2 = 0.0;
        for (size_t q = 0; q < nn; ++q) s2 += Z[q] * Z[q];
        return std::sqrt(s2);
    };
    bool converged = false;
    for (int k = 0; k < 10; ++k) {
        gemm('N', 'N', n, n, n, Ak.data(), n, Y.data(), n, T1.data(), n);
        gemm('N', 'T', n, n, n, T1.data(), n, Ak.data(), n, T2.data(), n);
        const double inc = fro(T2), ny = fro(Y);
        if (!std::isfinite(inc) || inc > 10.0 * ny) return false; // not a contraction: M is not stable (enough)
        for (size_t q = 0; q < nn; ++q) Y[q] += T2[q];
        if (inc <= 1e-8 * ny) { // the ratio squares from one step to the next: what is left is below 1e-16 (the check below decides)
            converged = true;
            break;
        }
        if (k >= 5 && inc > 1e-2 * ny) return false; // the ratio squares per step: would need more squarings than Bartels-Stewart costs
        gemm('N', 'N', n, n, n, Ak.data(), n, Ak.data(), n, T1.data(), n);
        Ak.swap(T1);
    }
    if (!converged) return false;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < j; ++i) {
            const double v = 0.5 * (Y[i + (size_t)j * n] + Y[j + (size_t)i * n]);
            Y[i + (size_t)j * n] = v;
            Y[j + (size_t)i * n] = v;
        }
    // verification: R = M Y + Y M' - C
    gemm('N', 'N', n, n, n, M.data(), n, Y.data(), n, T1.data(), n);
    double r2 = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const double r = T1[i + (size_t)j * n] + T1[j + (size_t)i * n] - C[i + (size_t)j * n]; // Y M' = (M Y)' for symmetric Y
            r2 += r * r;
        }
    if (!(std::sqrt(r2) <= 2e-15 * (2.0 * fro(M) * fro(Y) + fro(C)))) return false;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) X[i + (size_t)j * ldx] = Y[i + (size_t)j * n];
    return true;
}

// What the factored ADI route keeps from its last successful call (per thread): inside a restart cycle the solver's projected matrix
// grows by BORDERING -- V gets new orthonormal columns, V'AV keeps its leading block (src/LyapunovSolver.hpp:146-160) -- so the next
// call's matrix has this one as its leading block.  Then the shifts are kept (the extent of the spectrum moves by a few percent per
// trip: the estimate by power / inverse iteration, an LU of M and twelve solves, is skipped) and the LU factors of M - p_i I are
// EXTENDED instead of recomputed: with P11 A11 = L11 U11 from before,
//     U12 = L11^-1 P11 A12,   L21 = A21 U11^-1,   P22 (A22 - L21 U12) = L22 U22,   rows of L21 swapped as P22 says,
// which is the factorisation dgetrf would produce if its pivot search in the first n1 columns stopped at row n1: (n - n1) / n of the
// triangular-solve work instead of a factorisation (n = 200, 16 new rows: 0.06 ms for two shifts instead of 0.26 + 0.19 for the bounds).
// M - p I with p > 0 and M stable is far from singular, and the result is judged by the residual fence either way; a call that does not
// converge with inherited shifts is repeated from scratch.
struct AdiCache {
    int n = 0, L = 0;
    bool tr = false;
    double a = 0.0, b = 0.0;
    std::vector<double> M, shift;
    std::vector<std::vector<double>> LUs;
    std::vector<std::vector<int>> ips;
    std::vector<std::vector<double>> Inv; // explicit (M - p_i I)^-1 (n >= 64: see adi_lyapunov_lowrank_once)
};
static AdiCache &adi_cache()
{
    static thread_local AdiCache c;
    return c;
}
static std::atomic<long> g_adi_extended{0}, g_adi_fresh{0};
extern "C" void rails_sb03md_adi_counts(long *extended, long *fresh)
{
    if (extended) *extended = g_adi_extended.load();
    if (fresh) *fresh = g_adi_fresh.load();
}

// LU (LAPACK's packed form, pivots 1-based) of the n x n matrix S whose leading n1 x n1 block has the factors (LU1, ip1)
static bool extend_lu(int n1, int n, const std::vector<double> &LU1, const std::vector<int> &ip1, std::vector<double> &S, std::vector<int> &ip)
{
    const int nb = n - n1, one = 1;
    const double d_one = 1.0, d_mone = -1.0;
    for (int j = 0; j < n1; ++j) memcpy(&S[(size_t)j * n], &LU1[(size_t)j * n1], sizeof(double) * n1);
    ip.assign(n, 0);
    for (int j = 0; j < n1; ++j) ip[j] = ip1[j];
    if (nb == 0) return true;
    double *A12 = &S[(size_t)n1 * n], *A21 = &S[n1], *A22 = &S[n1 + (size_t)n1 * n];
    g_lp.dlaswp(&nb, A12, &n, &one, &n1, ip1.data(), &one);
    g_lp.dtrsm("L", "L", "N", "U", &n1, &nb, &d_one, S.data(), &n, A12, &n);
    g_lp.dtrsm("R", "U", "N", "N", &nb, &n1, &d_one, S.data(), &n, A21, &n);
    g_lp.dgemm("N", "N", &nb, &nb, &n1, &d_mone, A21, &n, A12, &n, &d_one, A22, &n);
    std::vector<int> ip2(nb);
    int info = 0;
    g_lp.dgetrf(&nb, &nb, A22, &n, ip2.data(), &info);
    if (info != 0) return false;
    g_lp.dlaswp(&n1, A21, &n, &one, &nb, ip2.data(), &one);
    for (int j = 0; j < nb; ++j) ip[n1 + j] = ip2[j] + n1;
    return true;
}

// In place: S <- S^-1 (n x n), through dgetrf + dgetri (or n solves with the identity)
static bool invert_dense(int n, std::vector<double> &S)
{
    std::vector<int> ip(n);
    int info = 0;
    g_lp.dgetrf(&n, &n, S.data(), &n, ip.data(), &info);
    if (info != 0) return false;
    if (g_lp.dgetri) {
        double wq = 0.0;
        int lw = -1;
        g_lp.dgetri(&n, S.data(), &n, ip.data(), &wq, &lw, &info);
        lw = info == 0 ? std::max(n, (int)wq) : 64 * n;
        std::vector<double> work((size_t)lw);
        g_lp.dgetri(&n, S.data(), &n, ip.data(), work.data(), &lw, &info);
        return info == 0;
    }
    std::vector<double> I((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) I[j + (size_t)j * n] = 1.0;
    const char N = 'N';
    g_lp.dgetrs(&N, &n, &n, S.data(), &n, ip.data(), I.data(), &n, &info);
    S.swap(I);
    return info == 0;
}

// S (n x n, holding the matrix on entry) <- S^-1, given the inverse J1 of its leading n1 x n1 block: with E = J1 A12, F = A21 J1,
// C = (A22 - A21 E)^-1:  S^-1 = [J1 + E C F, -E C; -C F, C].  Four thin products and an (n - n1)-square inverse: (n - n1) / n of an inversion.
static bool extend_inverse(int n1, int n, const std::vector<double> &J1, std::vector<double> &S)
{
    const int nb = n - n1;
    if (nb == 0) {
        S = J1;
        return true;
    }
    const double d_one = 1.0, d_mone = -1.0, d_zero = 0.0;
    double *A12 = &S[(size_t)n1 * n], *A21 = &S[n1], *A22 = &S[n1 + (size_t)n1 * n];
    std::vector<double> E((size_t)n1 * nb), F((size_t)nb * n1), Cm((size_t)nb * nb), G((size_t)nb * n1);
    g_lp.dgemm("N", "N", &n1, &nb, &n1, &d_one, J1.data(), &n1, A12, &n, &d_zero, E.data(), &n1);
    g_lp.dgemm("N", "N", &nb, &n1, &n1, &d_one, A21, &n, J1.data(), &n1, &d_zero, F.data(), &nb);
    for (int j = 0; j < nb; ++j)
        for (int i = 0; i < nb; ++i) Cm[i + (size_t)j * nb] = A22[i + (size_t)j * n];
    g_lp.dgemm("N", "N", &nb, &nb, &n1, &d_mone, A21, &n, E.data(), &n1, &d_one, Cm.data(), &nb);
    if (!invert_dense(nb, Cm)) return false;
    g_lp.dgemm("N", "N", &nb, &n1, &nb, &d_one, Cm.data(), &nb, F.data(), &nb, &d_zero, G.data(), &nb); // G = C F
    // leading block: J1 + E G
    for (int j = 0; j < n1; ++j) memcpy(&S[(size_t)j * n], &J1[(size_t)j * n1], sizeof(double) * n1);
    g_lp.dgemm("N", "N", &n1, &n1, &nb, &d_one, E.data(), &n1, G.data(), &nb, &d_one, S.data(), &n);
    // -E C, -G, C
    g_lp.dgemm("N", "N", &n1, &nb, &nb, &d_mone, E.data(), &n1, Cm.data(), &nb, &d_zero, A12, &n);
    for (int j = 0; j < n1; ++j)
        for (int i = 0; i < nb; ++i) A21[i + (size_t)j * n] = -G[i + (size_t)j * nb];
    for (int j = 0; j < nb; ++j)
        for (int i = 0; i < nb; ++i) A22[i + (size_t)j * n] = Cm[i + (size_t)j * nb];
    return true;
}

// Factored ADI for a right-hand side of low rank -- the solver's is +-(V'B)(V'B)' with p = 16 columns.  C = sign * F F' (pivoted
// Cholesky, rank r) and, for shifts p_1, p_2, ... > 0 (Li / White's low-rank ADI),
//     Z_1 = sqrt(2 p_1) (M - p_1 I)^-1 F,   Z_j = sqrt(p_j / p_j-1) [I + (p_j + p_j-1) (M - p_j I)^-1] Z_j-1,   X = -sign * sum_j Z_j Z_j'.
// The shifts are L points spaced logarithmically over [a, b], the extent of M's spectrum along the real axis (a from a few inverse
// iterations with an LU of M, b from a few power iterations), used cyclically: every mode is damped by the shifts near it, so the
// terms shrink by a roughly constant factor per cycle however far the spectrum is spread (the squared Smith iteration above has ONE
// shift and pays for spread spectra with squarings of n x n matrices).  Cost: L + 1 LU factorisations, one solve with r right-hand
// sides per term, X = Z Z', the residual check -- ~12 n^3 flops at n = 200, r = 16 against ~35 n^3.  All terms are positive
// semi-definite: nothing cancels.  Same fences: verified residual, early exit when the terms do not shrink, false = not applicable.
static bool adi_lyapunov_lowrank_once(bool tr, int n, const double *A, int lda, double *X, int ldx, bool *not_applicable, bool from_scratch, bool *inherited)
{
    *inherited = false;
    *not_applicable = false;
    if (!g_lp.dgetrf || !g_lp.dgetrs) return false;
    static const bool trace_lr = getenv("RAILS_SB03MD_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_mark = t_begin, t_factor = 0, t_bounds = 0, t_lu = 0, t_terms = 0, t_form = 0;
    auto lap = [&](double &acc) {
        const double t = now();
        acc += t - t_mark;
        t_mark = t;
    };
    const size_t nn = (size_t)n * n;
    std::vector<double> C(nn), W(nn);
    double trc = 0.0, dmax = 0.0;
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) C[i + (size_t)j * n] = 0.5 * (X[i + (size_t)j * ldx] + X[j + (size_t)i * ldx]);
        trc += C[j + (size_t)j * n];
        dmax = std::max(dmax, std::fabs(C[j + (size_t)j * n]));
    }
    if (!(dmax > 0.0) || !std::isfinite(dmax)) return false;
    const double sign = trc >= 0.0 ? 1.0 : -1.0;
    for (size_t q = 0; q < nn; ++q) W[q] = sign * C[q];
    std::vector<int> piv(n);
    int rank = 0, info = 0;
    rails_dpstrf('U', n, W.data(), n, piv.data(), &rank, 1e-15 * dmax, &info);
    if (info < 0 || rank <= 0 || rank > n / 3) { // not (semi-)definite of low rank: the dense form decides, and nothing is held against this route
        *not_applicable = true;
        return false;
    }
    std::vector<double> F((size_t)n * rank, 0.0); // F(piv[j], i) = R(i, j)
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < rank && i <= j; ++i) F[piv[j] + (size_t)i * n] = W[i + (size_t)j * n];
    std::vector<double> M(nn);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) M[i + (size_t)j * n] = tr ? A[i + (size_t)j * lda] : A[j + (size_t)i * lda];
    lap(t_factor);
    const char N = 'N';
    const int one = 1;
    // a bordered extension of the last matrix this thread solved for?  (see AdiCache)
    static const bool allow_inherit = !(getenv("RAILS_SB03MD_ADI_INHERIT") && atoi(getenv("RAILS_SB03MD_ADI_INHERIT")) == 0);
    AdiCache &cache = adi_cache();
    // From n = 64 on the step operators S_i = sqrt(p_i / p_i-1) [I + (p_i + p_i-1) (M - p_i I)^-1] are formed explicitly and a term is ONE
    // product S_i Z with 16 columns: a pair of triangular solves with 16 right-hand sides runs at a third of that rate, and there are
    // 20-odd terms (n = 177: 0.77 -> 0.23 ms).  An inversion costs more than a factorisation (0.3-0.4 ms per shift at n = 200), but inside a
    // restart cycle the inverse is EXTENDED like the matrix (extend_inverse: four thin products); the residual fence judges the result.
    static const bool allow_explicit = !(getenv("RAILS_SB03MD_ADI_INVERSE") && atoi(getenv("RAILS_SB03MD_ADI_INVERSE")) == 0);
    const bool explicit_steps = allow_explicit && n >= 64;
    bool inherit = allow_inherit && !from_scratch && g_lp.dlaswp && g_lp.dtrsm && cache.n >= 32 && cache.tr == tr && n >= cache.n && n - cache.n <= 64 && cache.L >= 1 &&
                   (explicit_steps ? (int)cache.Inv.size() == cache.L : (int)cache.LUs.size() == cache.L);
    for (int j = 0; inherit && j < cache.n; ++j) inherit = memcmp(&M[(size_t)j * n], &cache.M[(size_t)j * cache.n], sizeof(double) * cache.n) == 0;
    double a = 0.0, b = 0.0;
    int L = 1;
    std::vector<double> shift;
    std::vector<std::vector<double>> LUs, Inv;
    std::vector<std::vector<int>> ips;
    // extent of the spectrum along the real axis: b ~ largest modulus (power iteration), a ~ smallest (inverse iteration with the factors
    // of M - p I: the eigenvalue nearest p > 0 is the one of smallest modulus when the spectrum is near the negative real axis), six steps
    // each from a fixed start vector
    std::vector<double> v(n), u(n);
    auto nrm = [&](std::vector<double> const &z) {
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) s2 += z[i] * z[i];
        return std::sqrt(s2);
    };
    auto extent = [&](const std::vector<double> &LUp, const std::vector<int> &ipp, const std::vector<double> *InvP, double pshift, double *a_out, double *b_out) {
        for (int i = 0; i < n; ++i) v[i] = 1.0 + 0.37 * std::sin(1.0 + 2.3 * i);
        double bmax = 0.0, amin = 0.0;
        for (int it = 0; it < 6; ++it) {
            const double nv = nrm(v);
            for (int i = 0; i < n; ++i) v[i] /= nv;
            gemm('N', 'N', n, 1, n, M.data(), n, v.data(), n, u.data(), n);
            bmax = std::max(bmax, nrm(u));
            v = u;
        }
        for (int i = 0; i < n; ++i) v[i] = 1.0 + 0.37 * std::sin(1.0 + 2.3 * i);
        for (int it = 0; it < 6; ++it) {
            const double nv = nrm(v);
            for (int i = 0; i < n; ++i) v[i] /= nv;
            if (InvP) {
                gemm('N', 'N', n, 1, n, InvP->data(), n, v.data(), n, u.data(), n);
                v = u;
            } else {
                int inf = 0;
                g_lp.dgetrs(&N, &n, &one, LUp.data(), &n, ipp.data(), v.data(), &n, &inf);
                if (inf != 0) return false;
            }
            amin = nrm(v); // -> 1 / |lambda - p|min
        }
        if (!(amin > 0.0) || !(bmax > 0.0) || !std::isfinite(amin) || !std::isfinite(bmax)) return false;
        const double lmin = 1.0 / amin - pshift;
        if (!(lmin > 0.0)) return false;
        *a_out = 0.8 * lmin, *b_out = 1.2 * bmax;
        if (!(*a_out < *b_out)) *a_out = 0.5 * *b_out;
        return true;
    };
    if (inherit) {
        a = cache.a, b = cache.b, L = cache.L, shift = cache.shift;
        LUs.assign(explicit_steps ? 0 : L, std::vector<double>());
        ips.assign(explicit_steps ? 0 : L, std::vector<int>());
        Inv.assign(explicit_steps ? L : 0, std::vector<double>());
        for (int i = 0; i < L && inherit; ++i) {
            std::vector<double> &S = explicit_steps ? Inv[i] : LUs[i];
            S = M;
            for (int j = 0; j < n; ++j) S[j + (size_t)j * n] -= shift[i];
            inherit = explicit_steps ? extend_inverse(cache.n, n, cache.Inv[i], S) : extend_lu(cache.n, n, cache.LUs[i], cache.ips[i], S, ips[i]);
        }
        lap(t_lu);
        // do the inherited shifts still cover the spectrum?  (after a restart it widens quickly as the space grows: shifts chosen for
        // [9.5, 19] took 32 terms, then did not converge at all, on a matrix whose spectrum had reached [1.3, 20])
        double a_now = 0.0, b_now = 0.0;
        if (inherit && !((explicit_steps ? extent(Inv[0], std::vector<int>(), &Inv[0], shift[0], &a_now, &b_now) : extent(LUs[0], ips[0], nullptr, shift[0], &a_now, &b_now)) &&
                         a_now >= 0.8 * a && b_now <= 1.15 * b))
            inherit = false;
        lap(t_bounds);
    }
    if (!inherit) {
    std::vector<double> LU0 = M;
    std::vector<int> ip0(n);
    g_lp.dgetrf(&n, &n, LU0.data(), &n, ip0.data(), &info);
    if (info != 0) return false;
    if (!extent(LU0, ip0, nullptr, 0.0, &a, &b)) return false;
    lap(t_bounds);
    if (b / a > 1e5) return false;
    // Wachspress' optimal real ADI parameters for [a, b]: with L = 2^s of them one sweep damps every mode by
    // rho_L ~ 4 exp(-pi^2 L / ln(4 b/a)); the 2L parameters of [a, b] follow from the L parameters x of [sqrt(ab), (a+b)/2] as
    // x +- sqrt(x^2 - ab).  A factorisation costs about four 16-column solves, so L is the power of two with the smallest
    // 4 L + (terms needed for 1e-8.5 in Z, i.e. 1e-17 in X).
    const double kappa = b / a, pi = 3.14159265358979323846;
    double best_cost = 1e300;
    for (int cand = 1; cand <= 8; cand *= 2) {
        double rho = cand == 1 ? (std::sqrt(kappa) - 1.0) / (std::sqrt(kappa) + 1.0) : std::min(0.999, 4.0 * std::exp(-pi * pi * cand / std::log(4.0 * kappa)));
        double sweeps = std::ceil(std::log(3e-9) / std::log(std::max(rho, 1e-300)));
        double cost = 4.0 * cand + cand * std::max(1.0, sweeps);
        if (cost < best_cost) best_cost = cost, L = cand;
    }
    shift.assign(1, std::sqrt(a * b));
    {
        // intervals down the recursion, then the parameters back up
        std::vector<std::pair<double, double>> iv(1, std::make_pair(a, b));
        for (int l = L; l > 1; l /= 2) iv.push_back(std::make_pair(std::sqrt(iv.back().first * iv.back().second), 0.5 * (iv.back().first + iv.back().second)));
        shift.assign(1, std::sqrt(iv.back().first * iv.back().second));
        for (int lev = (int)iv.size() - 2; lev >= 0; --lev) {
            const double ab = iv[lev].first * iv[lev].second;
            std::vector<double> up;
            for (double x : shift) {
                const double d = std::sqrt(std::max(0.0, x * x - ab));
                up.push_back(x + d);
                up.push_back(x - d);
            }
            shift.swap(up);
        }
        std::sort(shift.begin(), shift.end());
    }
    LUs.assign(explicit_steps ? 0 : L, std::vector<double>());
    ips.assign(explicit_steps ? 0 : L, std::vector<int>(n));
    Inv.assign(explicit_steps ? L : 0, std::vector<double>());
    for (int i = 0; i < L; ++i) {
        std::vector<double> &S = explicit_steps ? Inv[i] : LUs[i];
        S = M;
        for (int j = 0; j < n; ++j) S[j + (size_t)j * n] -= shift[i];
        if (explicit_steps) {
            if (!invert_dense(n, S)) return false;
        } else {
            g_lp.dgetrf(&n, &n, S.data(), &n, ips[i].data(), &info);
            if (info != 0) return false;
        }
    }
    lap(t_lu);
    }
    (inherit ? g_adi_extended : g_adi_fresh)++;
    *inherited = inherit;
    lap(t_lu);
    const size_t blk = (size_t)n * rank;
    const int max_terms = std::max(64, 16 * L), max_cols = 8 * n;
    std::vector<double> Z, T(blk);
    Z.reserve(blk * 24);
    auto norm2 = [&](const double *z) {
        double s2 = 0.0;
        for (size_t q = 0; q < blk; ++q) s2 += z[q] * z[q];
        return s2;
    };
    // the step operators of the explicit form (the inverses themselves are what the next call extends: they are kept as they are)
    std::vector<std::vector<double>> Sm(explicit_steps ? L : 0);
    for (int i = 0; i < (int)Sm.size(); ++i) {
        const double pj = shift[i], pp = shift[(i + L - 1) % L];
        const double f = std::sqrt(pj / pp), fg = f * (pj + pp);
        Sm[i].resize(nn);
        for (size_t q = 0; q < nn; ++q) Sm[i][q] = fg * Inv[i][q];
        for (int j = 0; j < n; ++j) Sm[i][j + (size_t)j * n] += f;
    }
    if (explicit_steps)
        gemm('N', 'N', n, rank, n, Inv[0].data(), n, F.data(), n, T.data(), n);
    else {
        T = F;
        g_lp.dgetrs(&N, &n, &rank, LUs[0].data(), &n, ips[0].data(), T.data(), &n, &info);
        if (info != 0) return false;
    }
    {
        const double f = std::sqrt(2.0 * shift[0]);
        for (size_t q = 0; q < blk; ++q) T[q] *= f;
    }
    Z.insert(Z.end(), T.begin(), T.end());
    double total = norm2(T.data()), cycle_start = total;
    bool converged = false;
    for (int j = 1; j < max_terms; ++j) {
        const int cur = j % L, prv = (j - 1) % L;
        const double pj = shift[cur], pp = shift[prv];
        const double *zp = Z.data() + (size_t)(j - 1) * blk;
        if (explicit_steps)
            gemm('N', 'N', n, rank, n, Sm[cur].data(), n, zp, n, T.data(), n);
        else {
            std::copy(zp, zp + blk, T.begin());
            g_lp.dgetrs(&N, &n, &rank, LUs[cur].data(), &n, ips[cur].data(), T.data(), &n, &info);
            if (info != 0) return false;
            const double f = std::sqrt(pj / pp), g = pj + pp;
            for (size_t q = 0; q < blk; ++q) T[q] = f * (zp[q] + g * T[q]);
        }
        const double n2 = norm2(T.data());
        if (!std::isfinite(n2) || n2 > 1e4 * total) return false; // M is not stable
        if ((int)((size_t)(j + 1) * rank) > max_cols) return false;
        Z.insert(Z.end(), T.begin(), T.end());
        total += n2;
        if (n2 <= 1e-17 * total) {
            converged = true;
            break;
        }
        if (cur == L - 1) { // end of a cycle: against the same term of the cycle before it must have shrunk by a clear factor
            if (j >= 2 * L - 1 && n2 > 0.2 * cycle_start) {
                if (trace_lr) fprintf(stderr, "sb03md ADI: n %d, spectrum ~[%.3g, %.3g], %d shifts: term %d is %.2e of the same term one cycle earlier -- giving up\n", n, a, b, L, j, n2 / cycle_start);
                return false;
            }
            cycle_start = n2;
        }
    }
    lap(t_terms);
    if (trace_lr) fprintf(stderr, "sb03md ADI: n %d rank %d, spectrum ~[%.3g, %.3g], %d shifts, %d terms, converged %d\n", n, rank, a, b, L, (int)(Z.size() / blk), (int)converged);
    if (!converged) return false;
    const int cols = (int)(Z.size() / n);
    std::vector<double> Y(nn);
    if (g_lp.dsyrk) { // the upper triangle at half the flops, mirrored
        const double one_d = 1.0, zero_d = 0.0;
        g_lp.dsyrk("U", "N", &n, &cols, &one_d, Z.data(), &n, &zero_d, Y.data(), &n);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < j; ++i) Y[j + (size_t)i * n] = Y[i + (size_t)j * n];
    } else {
        gemm('N', 'T', n, n, cols, Z.data(), n, Z.data(), n, Y.data(), n);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < j; ++i) {
                const double vv = 0.5 * (Y[i + (size_t)j * n] + Y[j + (size_t)i * n]);
                Y[i + (size_t)j * n] = vv;
                Y[j + (size_t)i * n] = vv;
            }
    }
    for (size_t q = 0; q < nn; ++q) Y[q] *= -sign; // M Y + Y M' = C
    // verification with the ORIGINAL right-hand side (also covers the truncation of the pivoted Cholesky factor)
    gemm('N', 'N', n, n, n, M.data(), n, Y.data(), n, W.data(), n);
    double r2 = 0.0, m2 = 0.0, y2 = 0.0, c2 = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const double r = W[i + (size_t)j * n] + W[j + (size_t)i * n] - C[i + (size_t)j * n];
            r2 += r * r;
            m2 += M[i + (size_t)j * n] * M[i + (size_t)j * n];
            y2 += Y[i + (size_t)j * n] * Y[i + (size_t)j * n];
            c2 += C[i + (size_t)j * n] * C[i + (size_t)j * n];
        }
    lap(t_form);
    if (trace_lr)
        fprintf(stderr, "sb03md ADI: residual %.2e of %.2e allowed; ms: factor C %.2f, bounds %.2f, %d LU %.2f, terms %.2f, X = ZZ' + check %.2f, total %.2f\n", std::sqrt(r2),
                2e-15 * (2.0 * std::sqrt(m2 * y2) + std::sqrt(c2)), t_factor, t_bounds, L, t_lu, t_terms, t_form, now() - t_begin);
    if (!(std::sqrt(r2) <= 2e-15 * (2.0 * std::sqrt(m2 * y2) + std::sqrt(c2)))) return false;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) X[i + (size_t)j * ldx] = Y[i + (size_t)j * n];
    // what the next call may build on
    cache.n = n, cache.L = L, cache.tr = tr, cache.a = a, cache.b = b;
    cache.M.swap(M);
    cache.shift.swap(shift);
    cache.LUs.swap(LUs);
    cache.ips.swap(ips);
    cache.Inv.swap(Inv);
    return true;
}

static bool adi_lyapunov_lowrank(bool tr, int n, const double *A, int lda, double *X, int ldx, bool *not_applicable)
{
    bool inherited = false;
    if (adi_lyapunov_lowrank_once(tr, n, A, lda, X, ldx, not_applicable, false, &inherited)) return true;
    if (!inherited || *not_applicable) return false;
    // inherited shifts or factors did not do: once more from scratch (X still holds the right-hand side)
    return adi_lyapunov_lowrank_once(tr, n, A, lda, X, ldx, not_applicable, true, &inherited);
}

// Continuous-time Lyapunov equation, SB03MD('C','X','N',trans):
//   trans = 'T':  A X + X A^T = scale * C        trans = 'N':  A^T X + X A = scale * C
// C symmetric, X overwrites C.  A is overwritten by its real Schur form (as SLICOT does with FACT='N').
extern "C" void rails_sb03md(char dico, char job, char fact, char trans, int n, double *A, int lda, double *X, int ldx, double *scale,
                             int *info)
{
    *info = 0;
    if (n < 1) { // the reference prints "n < 1 is not supported" and returns (src/SlicotWrapper.cpp:12-16)
        fprintf(stderr, "rails_sb03md: n < 1 is not supported\n");
        return;
    }
    if ((dico != 'C' && dico != 'c') || (job != 'X' && job != 'x') || (fact != 'N' && fact != 'n')) {
        fprintf(stderr, "rails_sb03md: only DICO='C', JOB='X', FACT='N' is supported\n");
        *info = -1;
        return;
    }
    if (rails_host_lapack_init(nullptr) != RAILS_OK) {
        *info = -100;
        return;
    }
    if (!all_finite(n, n, A, lda) || !all_finite(n, n, X, ldx)) {
        fprintf(stderr, "rails_sb03md: A or C holds NaN or Inf entries\n");
        *info = n + 2;
        return;
    }
    const bool tr = (trans == 'T' || trans == 't' || trans == 'C' || trans == 'c');
    // Symmetric A (projections of symmetric operators: Laplacians, stencils): its Schur form is its eigen-decomposition
    // A = Q L Q', and the triangular Sylvester solve collapses to Y_ij = F_ij / (l_i + l_j).  Same equation, same result up to
    // rounding, about a third of the cost of dgees + dtrsyl.  Only for A symmetric to rounding; RAILS_SB03MD_SYMMETRIC=0 disables.
    static const bool use_symmetric = [] {
        const char *e = getenv("RAILS_SB03MD_SYMMETRIC");
        return e ? atoi(e) != 0 : true;
    }();
    if (use_symmetric && n >= 8) {
        double amax = 0.0, asym = 0.0;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < j; ++i) {
                const double a = A[i + (size_t)j * lda], b = A[j + (size_t)i * lda];
                amax = std::max(amax, std::max(std::fabs(a), std::fabs(b)));
                asym = std::max(asym, std::fabs(a - b));
            }
        for (int i = 0; i < n; ++i) amax = std::max(amax, std::fabs(A[i + (size_t)i * lda]));
        if (amax > 0.0 && asym <= 1e-13 * amax) {
            std::vector<double> Q((size_t)n * n), lam(n), F((size_t)n * n), W((size_t)n * n);
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) Q[i + (size_t)j * n] = 0.5 * (A[i + (size_t)j * lda] + A[j + (size_t)i * lda]);
            int sinfo = 0;
            if (g_lp.dsyevd) {
                int lwork = -1, liwork = -1, iq = 0;
                double wq = 0.0;
                g_lp.dsyevd("V", "U", &n, Q.data(), &n, lam.data(), &wq, &lwork, &iq, &liwork, &sinfo);
                lwork = (int)wq + 1;
                liwork = iq + 1;
                std::vector<double> work((size_t)lwork);
                std::vector<int> iwork((size_t)liwork);
                g_lp.dsyevd("V", "U", &n, Q.data(), &n, lam.data(), work.data(), &lwork, iwork.data(), &liwork, &sinfo);
            } else {
                int lwork = 3 * n + 64;
                std::vector<double> work((size_t)lwork);
                g_lp.dsyev("V", "U", &n, Q.data(), &n, lam.data(), work.data(), &lwork, &sinfo);
            }
            if (sinfo == 0) {
                gemm('T', 'N', n, n, n, Q.data(), n, X, ldx, W.data(), n);
                gemm('N', 'N', n, n, n, W.data(), n, Q.data(), n, F.data(), n);
                bool singular = false;
                for (int j = 0; j < n && !singular; ++j)
                    for (int i = 0; i < n; ++i) {
                        const double den = lam[i] + lam[j];
                        if (!(std::fabs(den) > 1e-300)) {
                            singular = true;
                            break;
                        }
                        F[i + (size_t)j * n] /= den;
                    }
                if (!singular) {
                    gemm('N', 'N', n, n, n, Q.data(), n, F.data(), n, W.data(), n);
                    gemm('N', 'T', n, n, n, W.data(), n, Q.data(), n, F.data(), n);
                    for (int j = 0; j < n; ++j)
                        for (int i = 0; i < n; ++i) {
                            X[i + (size_t)j * ldx] = F[i + (size_t)j * n];
                            A[i + (size_t)j * lda] = (i == j) ? lam[i] : 0.0; // the (diagonal) Schur form, as the general path leaves it
                        }
                    *scale = 1.0;
                    return;
                }
            }
            // eigen-solver trouble or lambda_i + lambda_j = 0: let the general path deal with it
        }
    }
    // Nonsymmetric A: the squared Smith iteration first (level-3 BLAS, verified; RAILS_SB03MD_SMITH=0 disables).  Where it does not
    // apply it says so after a few products; a solver whose projected matrices are of that kind stops asking for a while.  A keeps
    // its values on this path (SLICOT would leave the Schur form; no caller of this library reads A afterwards).
    static const bool use_smith = [] {
        const char *e = getenv("RAILS_SB03MD_SMITH");
        return e ? atoi(e) != 0 : true;
    }();
    int &smith_pause = sb03md_smith_pause();
    int &lowrank_pause = sb03md_factored_pause();
    bool adi_na = false; // the factored route did not apply (right-hand side not of low rank relative to n): no reason to rest it
    if (use_smith && n >= 32) {
        if (smith_pause > 0)
            --smith_pause;
        else if ((lowrank_pause > 0 ? (--lowrank_pause, false) : (adi_lyapunov_lowrank(tr, n, A, lda, X, ldx, &adi_na) || (lowrank_pause = adi_na ? 0 : 30, false))) ||
                 smith_lyapunov(tr, n, A, lda, X, ldx)) {
            g_sb03md_smith++;
            *scale = 1.0;
            return;
        } else
            smith_pause = 30;
    }
    g_sb03md_schur++;
    std::vector<double> U((size_t)n * n), wr(n), wi(n), F((size_t)n * n), W((size_t)n * n);
    int sdim = 0, lwork = -1, linfo = 0;
    double wq = 0.0;
    g_lp.dgees("V", "N", nullptr, &n, A, &lda, &sdim, wr.data(), wi.data(), U.data(), &n, &wq, &lwork, nullptr, &linfo);
    lwork = std::max((int)wq, 3 * n);
    std::vector<double> work((size_t)lwork);
    g_lp.dgees("V", "N", nullptr, &n, A, &lda, &sdim, wr.data(), wi.data(), U.data(), &n, work.data(), &lwork, nullptr, &linfo);
    if (linfo != 0) { // QR iteration failed: SLICOT reports 0 < info <= n
        *info = linfo > n ? n : linfo;
        return;
    }
    // F = U^T C U
    gemm('T', 'N', n, n, n, U.data(), n, X, ldx, W.data(), n);
    gemm('N', 'N', n, n, n, W.data(), n, U.data(), n, F.data(), n);
    // S Y + Y S^T = scale F (trans='T')   or   S^T Y + Y S = scale F (trans='N')
    int isgn = 1, tinfo = 0;
    static const bool use_blocked = [] {
        const char *e = getenv("RAILS_SB03MD_BLOCKED");
        return e ? atoi(e) != 0 : true;
    }();
    bool done = false;
    if (g_lp.dtrsyl3 && use_blocked && n >= 64) { // level-3 form: 2-4x faster than dtrsyl at n = 128..256, same equation
        int liwork = -1, ldswork = -1, iq = 0, qinfo = 0;
        double sq[2] = {0.0, 0.0};
        g_lp.dtrsyl3(tr ? "N" : "T", tr ? "T" : "N", &isgn, &n, &n, A, &lda, A, &lda, F.data(), &n, scale, &iq, &liwork, sq, &ldswork, &qinfo);
        if (qinfo == 0 && iq > 0 && sq[0] > 0 && sq[1] > 0) {
            liwork = iq;
            ldswork = (int)sq[0];
            const int swcols = (int)sq[1];
            std::vector<int> iwork((size_t)liwork);
            std::vector<double> swork((size_t)ldswork * swcols);
            g_lp.dtrsyl3(tr ? "N" : "T", tr ? "T" : "N", &isgn, &n, &n, A, &lda, A, &lda, F.data(), &n, scale, iwork.data(), &liwork, swork.data(),
                         &ldswork, &tinfo);
            done = tinfo >= 0;
        }
    }
    if (!done) g_lp.dtrsyl(tr ? "N" : "T", tr ? "T" : "N", &isgn, &n, &n, A, &lda, A, &lda, F.data(), &n, scale, &tinfo);
    // X = U Y U^T
    gemm('N', 'N', n, n, n, U.data(), n, F.data(), n, W.data(), n);
    gemm('N', 'T', n, n, n, W.data(), n, U.data(), n, F.data(), n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) X[i + (size_t)j * ldx] = F[i + (size_t)j * n];
    if (tinfo == 1) *info = n + 1; // perturbed to avoid overflow: A and -A^T have (nearly) common eigenvalues
}
