// sptrsv.hip -- sparse triangular solves on device panels: the A11 systems of the Schur-complement operator (src/SchurOperator.cpp:171-214:
// the reference solves them with Amesos KLU on the host; here the L and U factors a host factorisation produced are applied on the
// device, so that a product S * X = A22 X - A21 A11^-1 A12 X keeps every block where the SpMM kernels left it).
//
// Level scheduling: row i of a triangular matrix can be finished once the rows its off-diagonal entries name are; level(i) = 1 + the
// largest level among those rows, and all rows of a level are independent.  Rows are sorted by level on the host (rails_sptrsv_create);
// a solve walks the levels in order.  Wide levels get a launch of their own (one thread per row and column of the panel); runs of
// consecutive narrow levels (every level of the run at most 1024 row-columns) are walked by ONE workgroup with a barrier between levels
// -- factors of banded or nearly banded matrices have thousands of levels of a few rows each, and a launch per level would cost more
// than the arithmetic.  Memory-bound gather work: no matrix cores here.
#include "rails_internal.h"

#include <algorithm>
#include <vector>

struct rails_sptrsv {
    rails_ctx *ctx = nullptr;
    int64_t n = 0, nnz = 0;
    int unit_diag = 0;
    int64_t *rowptr = nullptr; // device CSR of the triangle, diagonal included unless unit_diag
    int32_t *col = nullptr;
    double *val = nullptr;
    int32_t *order = nullptr;     // rows sorted by level
    int64_t *level_ptr_dev = nullptr; // the level pointers once more, for the kernel that walks runs of narrow levels
    std::vector<int64_t> level_ptr; // host: level l is order[level_ptr[l] .. level_ptr[l + 1])
};

namespace {

// rows order[r0 .. r1) of one level, nc columns of the panel: x(row, :) = (x(row, :) - sum_j T(row, j) x(j, :)) / T(row, row)
// (XP: `double *` where everything read was written before the launch; `volatile double *` in the one-workgroup chain, whose levels read
// what the level before wrote -- those loads go to the L2)
template <typename XP>
__device__ __forceinline__ void sptrsv_row(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val, int unit_diag,
                                           int32_t row, int c, XP X, int ld)
{
    double acc = X[(int64_t)row * ld + c];
    double d = 1.0;
    for (int64_t q = rowptr[row]; q < rowptr[row + 1]; ++q) {
        const int32_t j = col[q];
        if (j == row)
            d = val[q];
        else
            acc -= val[q] * X[(int64_t)j * ld + c];
    }
    X[(int64_t)row * ld + c] = unit_diag ? acc : acc / d;
}

__global__ void k_sptrsv_level(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val, int unit_diag,
                               const int32_t *__restrict__ order, int64_t r0, int64_t r1, double *X, int ld, int nc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = r0 + t / nc;
    if (r >= r1) return;
    sptrsv_row<double *>(rowptr, col, val, unit_diag, order[r], (int)(t % nc), X, ld);
}

// levels l0 .. l1 - 1, each narrow, by one workgroup.  What a level writes the next one reads: stores are complete (and this CU's L1
// holds no older copy of the lines: it is written through) before anyone passes the barrier.
__global__ __launch_bounds__(1024) void k_sptrsv_chain(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val, int unit_diag,
                                                       const int32_t *__restrict__ order, const int64_t *__restrict__ level_ptr, int l0, int l1, double *X, int ld, int nc)
{
    for (int l = l0; l < l1; ++l) {
        const int64_t r0 = level_ptr[l], r1 = level_ptr[l + 1];
        for (int64_t t = threadIdx.x; t < (r1 - r0) * nc; t += blockDim.x)
            sptrsv_row<volatile double *>(rowptr, col, val, unit_diag, order[r0 + t / nc], (int)(t % nc), X, ld);
        __threadfence();
        __syncthreads();
    }
}

__global__ void k_gather_rows(const double *__restrict__ src, int lds, const int32_t *__restrict__ perm, int scatter, double *__restrict__ dst, int ldd, int64_t n, int nc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / nc;
    if (i >= n) return;
    const int c = (int)(t % nc);
    if (scatter)
        dst[(int64_t)perm[i] * ldd + c] = src[i * lds + c];
    else
        dst[i * ldd + c] = src[(int64_t)perm[i] * lds + c];
}

} // namespace

// A triangular matrix in CSR (columns of a row in any order; lower != 0: entries on or below the diagonal, else on or above; unit_diag != 0:
// the diagonal is one and not stored, entries on it are refused).  The level analysis runs here, on the host.
extern "C" int rails_sptrsv_create(rails_ctx *c, int64_t n, const int64_t *rowptr, const int32_t *col, const double *val, int lower, int unit_diag, rails_sptrsv **out)
{
    if (c) hipSetDevice(c->device);
    RAILS_REQUIRE(c && out && n >= 1 && n < ((int64_t)1 << 31) && rowptr && rowptr[0] == 0, "rails_sptrsv_create: bad argument");
    const int64_t nnz = rowptr[n];
    RAILS_REQUIRE(nnz >= 0 && (nnz == 0 || (col && val)), "rails_sptrsv_create: bad arrays");
    std::vector<int32_t> level(n, 0);
    int32_t nlev = 0;
    for (int64_t s = 0; s < n; ++s) {
        const int64_t i = lower ? s : n - 1 - s;
        int32_t lv = 0;
        bool diag = false;
        RAILS_REQUIRE(rowptr[i + 1] >= rowptr[i], "rails_sptrsv_create: row pointers decrease at row %lld", (long long)i);
        for (int64_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
            const int32_t j = col[q];
            RAILS_REQUIRE(j >= 0 && j < n && (lower ? j <= i : j >= i), "rails_sptrsv_create: entry (%lld, %d) is outside the %s triangle", (long long)i, j, lower ? "lower" : "upper");
            if (j == i) {
                RAILS_REQUIRE(!unit_diag, "rails_sptrsv_create: a unit triangle with a stored diagonal entry in row %lld", (long long)i);
                RAILS_REQUIRE(val[q] != 0.0, "rails_sptrsv_create: zero on the diagonal in row %lld", (long long)i);
                diag = true;
            } else
                lv = std::max(lv, level[j] + 1);
        }
        RAILS_REQUIRE(unit_diag || diag, "rails_sptrsv_create: no diagonal entry in row %lld", (long long)i);
        level[i] = lv;
        nlev = std::max(nlev, lv + 1);
    }
    rails_sptrsv *T = new rails_sptrsv;
    T->ctx = c;
    T->n = n;
    T->nnz = nnz;
    T->unit_diag = unit_diag ? 1 : 0;
    T->level_ptr.assign((size_t)nlev + 1, 0);
    for (int64_t i = 0; i < n; ++i) T->level_ptr[level[i] + 1]++;
    for (int32_t l = 0; l < nlev; ++l) T->level_ptr[l + 1] += T->level_ptr[l];
    std::vector<int32_t> order(n);
    {
        std::vector<int64_t> fill(T->level_ptr.begin(), T->level_ptr.end() - 1);
        for (int64_t i = 0; i < n; ++i) order[fill[level[i]]++] = (int32_t)i;
    }
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        *dst = nullptr;
        if (bytes == 0) return RAILS_OK;
        RAILS_HIP_CHECK(hipMalloc(dst, bytes));
        RAILS_HIP_CHECK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return RAILS_OK;
    };
    int rc = up((void **)&T->rowptr, rowptr, (size_t)(n + 1) * sizeof(int64_t));
    if (rc == RAILS_OK) rc = up((void **)&T->col, col, (size_t)nnz * sizeof(int32_t));
    if (rc == RAILS_OK) rc = up((void **)&T->val, val, (size_t)nnz * sizeof(double));
    if (rc == RAILS_OK) rc = up((void **)&T->order, order.data(), (size_t)n * sizeof(int32_t));
    if (rc == RAILS_OK) rc = up((void **)&T->level_ptr_dev, T->level_ptr.data(), T->level_ptr.size() * sizeof(int64_t));
    if (rc != RAILS_OK) {
        rails_sptrsv_destroy(T);
        return rc;
    }
    *out = T;
    return RAILS_OK;
}

extern "C" void rails_sptrsv_destroy(rails_sptrsv *T)
{
    if (!T) return;
    if (T->ctx) hipSetDevice(T->ctx->device);
    hipFree(T->rowptr);
    hipFree(T->col);
    hipFree(T->val);
    hipFree(T->order);
    hipFree(T->level_ptr_dev);
    delete T;
}

extern "C" int64_t rails_sptrsv_levels(const rails_sptrsv *T) { return T ? (int64_t)T->level_ptr.size() - 1 : 0; }

// X[:, c0:c0+nc] <- T^-1 X[:, c0:c0+nc], in place, on the context's stream
extern "C" int rails_sptrsv_solve(rails_ctx *c, const rails_sptrsv *T, rails_panel *X, int c0, int nc)
{
    if (c) hipSetDevice(c->device);
    RAILS_REQUIRE(c && T && X, "rails_sptrsv_solve: null argument");
    RAILS_REQUIRE(X->m == T->n && c0 >= 0 && nc >= 0 && c0 + nc <= X->cap, "rails_sptrsv_solve: a panel of %lld rows, window [%d, %d) for a triangle of order %lld",
                  (long long)X->m, c0, c0 + nc, (long long)T->n);
    if (nc == 0) return RAILS_OK;
    const int nlev = (int)T->level_ptr.size() - 1;
    const int64_t *lp = T->level_ptr_dev;
    double *Xd = X->d + c0;
    int l = 0;
    while (l < nlev) {
        const int64_t work = (T->level_ptr[l + 1] - T->level_ptr[l]) * nc;
        if (work > 1024) {
            RAILS_LAUNCH(k_sptrsv_level, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, c->stream, T->rowptr, T->col, T->val, T->unit_diag, T->order, T->level_ptr[l],
                         T->level_ptr[l + 1], Xd, X->ld, nc);
            ++l;
        } else {
            int l1 = l + 1;
            while (l1 < nlev && (T->level_ptr[l1 + 1] - T->level_ptr[l1]) * nc <= 1024) ++l1;
            RAILS_LAUNCH(k_sptrsv_chain, dim3(1), dim3(1024), 0, c->stream, T->rowptr, T->col, T->val, T->unit_diag, T->order, lp, l, l1, Xd, X->ld, nc);
            l = l1;
        }
    }
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

// Y[:, yc0:yc0+nc] row i <- X[:, xc0:xc0+nc] row perm[i] (scatter == 0) or Y row perm[i] <- X row i (scatter != 0); perm: n indices in DEVICE
// memory (rails_index_upload), a permutation of 0 .. n - 1.  X and Y must not overlap.
extern "C" int rails_panel_permute_rows(rails_ctx *c, const rails_panel *X, int xc0, int nc, const int32_t *perm_dev, int scatter, rails_panel *Y, int yc0)
{
    if (c) hipSetDevice(c->device);
    RAILS_REQUIRE(c && X && Y && perm_dev, "rails_panel_permute_rows: null argument");
    RAILS_REQUIRE(X->m == Y->m && nc >= 0 && xc0 >= 0 && yc0 >= 0 && xc0 + nc <= X->cap && yc0 + nc <= Y->cap, "rails_panel_permute_rows: bad windows");
    RAILS_REQUIRE(X->d != Y->d, "rails_panel_permute_rows: in place is not supported");
    if (nc == 0 || X->m == 0) return RAILS_OK;
    const int64_t work = X->m * nc;
    RAILS_LAUNCH(k_gather_rows, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, c->stream, X->d + xc0, X->ld, perm_dev, scatter ? 1 : 0, Y->d + yc0, Y->ld, X->m, nc);
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

extern "C" int rails_index_upload(rails_ctx *c, const int32_t *host, int64_t n, int32_t **out_dev)
{
    if (c) hipSetDevice(c->device);
    RAILS_REQUIRE(c && host && out_dev && n >= 1, "rails_index_upload: bad argument");
    RAILS_HIP_CHECK(hipMalloc((void **)out_dev, (size_t)n * sizeof(int32_t)));
    RAILS_HIP_CHECK(hipMemcpy(*out_dev, host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    return RAILS_OK;
}

extern "C" void rails_index_free(rails_ctx *c, int32_t *dev)
{
    if (c) hipSetDevice(c->device);
    if (dev) hipFree(dev);
}
