// spmm.hip -- CSR x tall-skinny panel product Y = op(A) X for gfx950.
//
// Replaces `A_ * W` of the reference (src/LyapunovSolver.hpp:146), i.e.
// Epetra_CrsMatrix::Apply (src/Epetra_OperatorWrapper.cpp:87) / the dense DGEMM of the Stl
// path (src/StlWrapper.cpp:181).  HBM-bound integer/fp64 streaming work: no MFMA here.
//
// Data layout: panels are row-major (ld padded to 128 B), so one nonzero a_ij gathers ONE
// contiguous row segment X[j, c0:c0+nc] -- a single 1 KiB wave-wide dwordx4 load at nc = 128.
//
// Kernel 1 (row-gather): LPR lanes own one row; (col,val) of the row are group-uniform
// (scalar loads when LPR == 64), U = 8 X-row loads are kept in flight per lane.
// Kernel 2 (LDS-staged footprint, see below): for matrices whose row blocks share columns
// (stencils, banded), the union of X rows a row block touches is staged once in LDS per
// column chunk and re-used by all rows of the block.
#include "rails_internal.h"

#include <algorithm>

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int VEC>
struct Acc;
template <>
struct Acc<1> {
    double v;
    __device__ __forceinline__ void zero() { v = 0.0; }
    __device__ __forceinline__ void fma(double a, const double *x) { v = __builtin_fma(a, *x, v); }
    __device__ __forceinline__ void store(double *y) { *y = v; }
};
template <>
struct Acc<2> {
    double2_t v;
    __device__ __forceinline__ void zero() { v = (double2_t){0.0, 0.0}; }
    __device__ __forceinline__ void fma(double a, const double *x)
    {
        double2_t t = *reinterpret_cast<const double2_t *>(x);
        v.x = __builtin_fma(a, t.x, v.x);
        v.y = __builtin_fma(a, t.y, v.y);
    }
    __device__ __forceinline__ void store(double *y) { *reinterpret_cast<double2_t *>(y) = v; }
};

// One group of LPR lanes per row, RPG consecutive rows per group.
// Xg/ldg: ghost rows (column index >= m_local) for row-partitioned runs, else unused.
template <int LPR, int VEC, int RPG>
__global__ __launch_bounds__(256) void k_spmm_rowgather(int64_t m, const int64_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ X, int ldx, const double *__restrict__ Xg,
                                                        int ldg, double *__restrict__ Y, int ldy, int nc)
{
    constexpr int GROUPS = 256 / LPR;
    constexpr int U = 8;
    const int g = threadIdx.x / LPR;
    const int l = threadIdx.x % LPR;
    int64_t row_base = ((int64_t)blockIdx.x * GROUPS + g) * RPG;
    if (LPR == 64) row_base = ((int64_t)blockIdx.x * GROUPS + __builtin_amdgcn_readfirstlane(g)) * RPG;

    for (int rr = 0; rr < RPG; ++rr) {
        const int64_t row = row_base + rr;
        if (row >= m) break;
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        for (int cb = l * VEC; cb < nc; cb += LPR * VEC) {
            const bool full = (cb + VEC <= nc);
            Acc<VEC> acc;
            acc.zero();
            double tail = 0.0; // VEC == 2 and only one valid column
            for (int64_t p = p0; p < p1; p += U) {
                int32_t c[U];
                double a[U];
                const int cnt = (int)((p1 - p) < U ? (p1 - p) : U);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = u < cnt;
                    c[u] = col[ok ? p + u : p0];
                    a[u] = ok ? val[p + u] : 0.0;
                }
                const double *src[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    src[u] = (c[u] < m) ? (X + (int64_t)c[u] * ldx + cb) : (Xg + ((int64_t)c[u] - m) * ldg + cb);
                if (full) {
#pragma unroll
                    for (int u = 0; u < U; ++u) acc.fma(a[u], src[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) tail = __builtin_fma(a[u], *src[u], tail);
                }
            }
            double *dst = Y + row * ldy + cb;
            if (full)
                acc.store(dst);
            else
                *dst = tail;
        }
    }
}

__global__ void k_pack_rows(const int64_t *__restrict__ rows, int64_t n, const double *__restrict__ X, int ldx, int nc,
                            double *__restrict__ out)
{
    int64_t total = n * nc;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = idx / nc;
        int cidx = (int)(idx - i * nc);
        out[idx] = X[rows[i] * ldx + cidx];
    }
}

template <int LPR, int VEC>
int launch_rg(rails_ctx *c, const rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc)
{
    constexpr int GROUPS = 256 / LPR;
    constexpr int RPG = (LPR >= 32) ? 4 : 2;
    int64_t rows_per_block = (int64_t)GROUPS * RPG;
    int64_t grid = (A->m + rows_per_block - 1) / rows_per_block;
    RAILS_REQUIRE(grid <= 0x7fffffffLL, "rails_spmm: grid too large");
    hipLaunchKernelGGL((k_spmm_rowgather<LPR, VEC, RPG>), dim3((unsigned)grid), dim3(256), 0, c->stream, A->m, A->rowptr, A->col,
                       A->val, X, ldx, Xg, ldg, Y, ldy, nc);
    return RAILS_OK;
}

template <int VEC>
int dispatch_rg(rails_ctx *c, const rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc)
{
    int need = (nc + VEC - 1) / VEC;
    if (need >= 64) return launch_rg<64, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc);
    if (need > 16) return launch_rg<32, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc);
    if (need > 8) return launch_rg<16, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc);
    if (need > 4) return launch_rg<8, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc);
    if (need > 2) return launch_rg<4, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc);
    if (need > 1) return launch_rg<2, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc);
    return launch_rg<1, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc);
}

int build_transpose(rails_csr *A)
{
    if (A->AT) return RAILS_OK;
    RAILS_REQUIRE(A->n_ghost == 0 && A->ncols_ext == A->m, "rails_spmm: transposed apply is single-GPU only");
    const int64_t m = A->m, nnz = A->nnz;
    std::vector<int64_t> rp(m + 1, 0);
    for (int64_t p = 0; p < nnz; ++p) rp[A->h_col[p] + 1]++;
    for (int64_t i = 0; i < m; ++i) rp[i + 1] += rp[i];
    std::vector<int32_t> ci(nnz);
    std::vector<double> va(nnz);
    std::vector<int64_t> next(rp.begin(), rp.end() - 1);
    for (int64_t i = 0; i < m; ++i)
        for (int64_t p = A->h_rowptr[i]; p < A->h_rowptr[i + 1]; ++p) {
            int64_t q = next[A->h_col[p]]++;
            ci[q] = (int32_t)i;
            va[q] = A->h_val[p];
        }
    return rails_csr_create(A->ctx, m, m, rp.data(), ci.data(), va.data(), &A->AT);
}

} // namespace

extern "C" int rails_csr_create(rails_ctx *c, int64_t m_local, int64_t n_cols_ext, const int64_t *rowptr, const int32_t *col,
                                const double *val, rails_csr **out)
{
    RAILS_REQUIRE(c && out && rowptr, "rails_csr_create: null argument");
    RAILS_REQUIRE(m_local >= 0 && n_cols_ext >= 0 && n_cols_ext <= 0x7fffffffLL, "rails_csr_create: bad shape %lld x %lld",
                  (long long)m_local, (long long)n_cols_ext);
    RAILS_REQUIRE(rowptr[0] == 0, "rails_csr_create: rowptr[0] != 0");
    int64_t nnz = rowptr[m_local];
    RAILS_REQUIRE(nnz >= 0 && (nnz == 0 || (col && val)), "rails_csr_create: bad nnz / null arrays");
    int maxrow = 0;
    for (int64_t i = 0; i < m_local; ++i) {
        int64_t d = rowptr[i + 1] - rowptr[i];
        RAILS_REQUIRE(d >= 0 && d <= 0x7fffffffLL, "rails_csr_create: rowptr not monotone at row %lld", (long long)i);
        if (d > maxrow) maxrow = (int)d;
    }
    // host-side index validation: an out-of-range column would fault the kernel
    for (int64_t p = 0; p < nnz; ++p)
        RAILS_REQUIRE(col[p] >= 0 && col[p] < n_cols_ext, "rails_csr_create: column %d out of range at nz %lld", col[p], (long long)p);
    rails_csr *A = new rails_csr();
    A->ctx = c;
    A->m = m_local;
    A->ncols_ext = n_cols_ext;
    A->nnz = nnz;
    A->max_row_nnz = maxrow;
    A->h_rowptr.assign(rowptr, rowptr + m_local + 1);
    if (nnz) {
        A->h_col.assign(col, col + nnz);
        A->h_val.assign(val, val + nnz);
    }
    hipError_t e1 = hipMalloc((void **)&A->rowptr, (size_t)(m_local + 1) * sizeof(int64_t));
    hipError_t e2 = hipMalloc((void **)&A->col, (size_t)(nnz ? nnz : 1) * sizeof(int32_t));
    hipError_t e3 = hipMalloc((void **)&A->val, (size_t)(nnz ? nnz : 1) * sizeof(double));
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        rails_set_error("rails_csr_create: device allocation failed");
        rails_csr_destroy(A);
        return RAILS_ENOMEM;
    }
    RAILS_HIP_CHECK(hipMemcpyAsync(A->rowptr, rowptr, (size_t)(m_local + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (nnz) {
        RAILS_HIP_CHECK(hipMemcpyAsync(A->col, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        RAILS_HIP_CHECK(hipMemcpyAsync(A->val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    RAILS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *out = A;
    return RAILS_OK;
}

extern "C" int rails_csr_destroy(rails_csr *A)
{
    if (!A) return RAILS_OK;
    hipStreamSynchronize(A->ctx->stream);
    if (A->AT) rails_csr_destroy(A->AT);
    if (A->rowptr) hipFree(A->rowptr);
    if (A->col) hipFree(A->col);
    if (A->val) hipFree(A->val);
    if (A->send_rows) hipFree(A->send_rows);
    if (A->send_buf) hipFree(A->send_buf);
    if (A->ext) hipFree(A->ext);
    if (A->t_fp_ptr) hipFree(A->t_fp_ptr);
    if (A->t_fp) hipFree(A->t_fp);
    if (A->t_lcol) hipFree(A->t_lcol);
    delete A;
    return RAILS_OK;
}

extern "C" int64_t rails_csr_rows(const rails_csr *A) { return A ? A->m : -1; }
extern "C" int64_t rails_csr_nnz(const rails_csr *A) { return A ? A->nnz : -1; }
extern "C" const char *rails_csr_last_kernel(const rails_csr *A) { return A ? A->last_kernel : ""; }

extern "C" int rails_csr_set_variant(rails_csr *A, int variant)
{
    RAILS_REQUIRE(A && variant >= 0 && variant <= 2, "rails_csr_set_variant: bad argument");
    A->variant = variant;
    return RAILS_OK;
}

extern "C" int rails_csr_set_halo(rails_csr *A, int64_t n_send, const int64_t *send_rows, int64_t n_ghost, rails_halo_fn fn,
                                  void *user)
{
    RAILS_REQUIRE(A, "null operator");
    RAILS_REQUIRE(n_send >= 0 && n_ghost >= 0 && A->m + n_ghost == A->ncols_ext,
                  "rails_csr_set_halo: m_local %lld + ghosts %lld != extended columns %lld", (long long)A->m, (long long)n_ghost,
                  (long long)A->ncols_ext);
    RAILS_REQUIRE((n_send == 0 && n_ghost == 0) || fn, "rails_csr_set_halo: halo hook missing");
    for (int64_t i = 0; i < n_send; ++i)
        RAILS_REQUIRE(send_rows[i] >= 0 && send_rows[i] < A->m, "rails_csr_set_halo: send row %lld out of range", (long long)send_rows[i]);
    rails_ctx *c = A->ctx;
    if (A->send_rows) {
        RAILS_HIP_CHECK(hipStreamSynchronize(c->stream));
        RAILS_HIP_CHECK(hipFree(A->send_rows));
        A->send_rows = nullptr;
    }
    A->n_send = n_send;
    A->n_ghost = n_ghost;
    A->halo = fn;
    A->halo_user = user;
    if (n_send) {
        RAILS_HIP_CHECK(hipMalloc((void **)&A->send_rows, (size_t)n_send * sizeof(int64_t)));
        RAILS_HIP_CHECK(hipMemcpy(A->send_rows, send_rows, (size_t)n_send * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    return RAILS_OK;
}

int rails_spmm_tiled(rails_ctx *c, rails_csr *A, const double *X, int ldx, double *Y, int ldy, int nc, bool *done);

extern "C" int rails_spmm(rails_ctx *c, rails_csr *A, int trans, const rails_panel *X, int xc0, int nc, rails_panel *Y, int yc0)
{
    RAILS_REQUIRE(c && A && X && Y, "rails_spmm: null argument");
    RAILS_REQUIRE(xc0 >= 0 && nc >= 0 && xc0 + nc <= X->cap, "rails_spmm: X columns [%d,%d) outside capacity %d", xc0, xc0 + nc, X->cap);
    RAILS_REQUIRE(yc0 >= 0 && yc0 + nc <= Y->cap, "rails_spmm: Y columns [%d,%d) outside capacity %d", yc0, yc0 + nc, Y->cap);
    RAILS_REQUIRE(X->m == A->m && Y->m == A->m, "rails_spmm: operator has %lld rows, X %lld, Y %lld", (long long)A->m, (long long)X->m,
                  (long long)Y->m);
    if (X->d == Y->d) RAILS_REQUIRE(xc0 + nc <= yc0 || yc0 + nc <= xc0, "rails_spmm: X and Y windows alias");
    if (nc == 0 || A->m == 0) return RAILS_OK;
    if (trans) {
        RAILS_TRY(build_transpose(A));
        A->AT->variant = A->variant;
        int rc = rails_spmm(c, A->AT, 0, X, xc0, nc, Y, yc0);
        A->last_kernel = A->AT->last_kernel;
        return rc;
    }
    const double *Xp = X->d + xc0;
    double *Yp = Y->d + yc0;
    const double *Xg = Xp;
    int ldg = X->ld;
    if (A->n_ghost > 0 || A->n_send > 0) {
        // pack the rows the neighbours need, exchange, gather from [local | ghost]
        size_t sbytes = (size_t)(A->n_send ? A->n_send : 1) * nc * sizeof(double);
        size_t gbytes = (size_t)(A->n_ghost ? A->n_ghost : 1) * nc * sizeof(double);
        if (sbytes > A->send_cap) {
            RAILS_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (A->send_buf) RAILS_HIP_CHECK(hipFree(A->send_buf));
            A->send_buf = nullptr;
            RAILS_HIP_CHECK(hipMalloc((void **)&A->send_buf, sbytes));
            A->send_cap = sbytes;
        }
        if (gbytes > A->ext_cap) {
            RAILS_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (A->ext) RAILS_HIP_CHECK(hipFree(A->ext));
            A->ext = nullptr;
            RAILS_HIP_CHECK(hipMalloc((void **)&A->ext, gbytes));
            A->ext_cap = gbytes;
        }
        if (A->n_send) {
            int64_t total = A->n_send * nc;
            int grid = (int)std::min<int64_t>((total + 255) / 256, (int64_t)c->num_cu * 8);
            hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, c->stream, A->send_rows, A->n_send, Xp, X->ld, nc, A->send_buf);
        }
        int rc = A->halo(A->halo_user, A->send_buf, A->ext, nc, (void *)c->stream);
        if (rc != 0) {
            rails_set_error("rails_spmm: halo hook failed with code %d", rc);
            return RAILS_ECOMM;
        }
        Xg = A->ext;
        ldg = nc;
    }
    bool done = false;
    if (A->variant != 1 && A->n_ghost == 0) RAILS_TRY(rails_spmm_tiled(c, A, Xp, X->ld, Yp, Y->ld, nc, &done));
    if (!done) {
        RAILS_REQUIRE(A->variant != 2, "rails_spmm: LDS-staged kernel requested but not applicable to this operator/shape");
        bool vec2 = ((xc0 | yc0) & 1) == 0 && (ldg % 2 == 0);
        if (vec2)
            RAILS_TRY((dispatch_rg<2>(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc)));
        else
            RAILS_TRY((dispatch_rg<1>(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc)));
        A->last_kernel = "k_spmm_rowgather";
    }
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

// -----------------------------------------------------------------------------------------
// Kernel 2: LDS-staged footprint kernel -- placeholder until the tiling analysis is built
// (rails_spmm_tiled reports done = false and the row-gather kernel runs).
// -----------------------------------------------------------------------------------------
int rails_spmm_tiled(rails_ctx *, rails_csr *, const double *, int, double *, int, int, bool *done)
{
    *done = false;
    return RAILS_OK;
}
