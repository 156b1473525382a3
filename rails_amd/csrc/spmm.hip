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
typedef double2_t double2_a8_t __attribute__((aligned(8))); // 16-byte loads of two doubles that are only 8-byte aligned (odd column offsets)

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
        // (16-byte accesses that may be only 8-byte aligned -- windows on odd columns, odd panel strides: global memory asks for dword
        // alignment, and the instructions are the same ones)
        double2_t t = *reinterpret_cast<const double2_a8_t *>(x);
        v.x = __builtin_fma(a, t.x, v.x);
        v.y = __builtin_fma(a, t.y, v.y);
    }
    __device__ __forceinline__ void store(double *y) { *reinterpret_cast<double2_a8_t *>(y) = v; }
};

// One group of LPR lanes per row, RPG consecutive rows per group.
// Xg/ldg: ghost rows (column index >= m_local) for row-partitioned runs, else unused.
template <int LPR, int VEC, int RPG>
__global__ __launch_bounds__(256) void k_spmm_rowgather(int64_t m, const int64_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ X, int ldx, const double *__restrict__ Xg,
                                                        int ldg, double *__restrict__ Y, int ldy, int nc, int64_t blocks_per_xcd, int64_t mc)
{
    // m rows (of this launch: the operator's, or a range of them -- rowptr and Y then start at the range); mc local columns: column
    // indices from mc on are ghost rows, taken from Xg
    constexpr int GROUPS = 256 / LPR;
    constexpr int U = 8;
    const int g = threadIdx.x / LPR;
    const int l = threadIdx.x % LPR;
    // XCD-aware block -> row-range map: blocks are dealt round-robin over the 8 XCDs (b and b+8 share one), so giving
    // XCD x the contiguous logical blocks [x*bpx, (x+1)*bpx) keeps the sliding window of gathered X rows of each XCD
    // inside its own 4 MiB L2 (speed only: any placement computes the same rows).
    int64_t lb = blockIdx.x;
    if (blocks_per_xcd > 0) lb = (int64_t)(blockIdx.x & 7) * blocks_per_xcd + (blockIdx.x >> 3);
    int64_t row_base = (lb * GROUPS + g) * RPG;
    if (LPR == 64) row_base = (lb * GROUPS + __builtin_amdgcn_readfirstlane(g)) * RPG;

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
                    src[u] = (c[u] < mc) ? (X + (int64_t)c[u] * ldx + cb) : (Xg + ((int64_t)c[u] - mc) * ldg + cb);
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

int spmm_env(const char *name, int def)
{
    const char *e = getenv(name);
    return e ? atoi(e) : def;
}

// rows [r0, r0 + nrows) of the operator on stream st (defaults: all rows, the context's stream).  A span is how the row-partitioned
// product overlaps its halo exchange: interior rows on a second stream while the ghost rows travel, boundary rows afterwards.
struct RowSpan {
    int64_t r0 = 0, nrows = -1;
    hipStream_t st = nullptr;
};
// (the busy meter brackets launches on the context's stream only)
#define RAILS_LAUNCH_ON(st__, kern__, grid__, block__, lds__, ...)                           \
    do {                                                                                     \
        if ((st__) == c->stream)                                                             \
            RAILS_LAUNCH(kern__, grid__, block__, lds__, c->stream, __VA_ARGS__);            \
        else                                                                                 \
            hipLaunchKernelGGL(kern__, grid__, block__, lds__, (st__), __VA_ARGS__);        \
    } while (0)

template <int LPR, int VEC>
int launch_rg(rails_ctx *c, const rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, const RowSpan &sp = RowSpan())
{
    const int64_t m_rows = sp.nrows < 0 ? A->m : sp.nrows;
    hipStream_t st = sp.st ? sp.st : c->stream;
    constexpr int GROUPS = 256 / LPR;
    constexpr int RPG = (LPR >= 32) ? 4 : 2;
    int64_t rows_per_block = (int64_t)GROUPS * RPG;
    int64_t grid = (m_rows + rows_per_block - 1) / rows_per_block;
    static const int xcd_aware = spmm_env("RAILS_SPMM_XCD", 1);
    int64_t bpx = 0;
    if (xcd_aware && grid >= 64) {
        bpx = (grid + 7) / 8;
        grid = bpx * 8;
    }
    RAILS_REQUIRE(grid <= 0x7fffffffLL, "rails_spmm: grid too large");
    RAILS_LAUNCH_ON(st, (k_spmm_rowgather<LPR, VEC, RPG>), dim3((unsigned)grid), dim3(256), 0, m_rows, A->rowptr + sp.r0, A->col,
                    A->val, X, ldx, Xg, ldg, Y + sp.r0 * ldy, ldy, nc, bpx, A->m);
    return RAILS_OK;
}


// Kernel 1b (row-gather, column chunks inside one launch): for wide X (nc = 128) the sliding window of X rows that the
// row blocks of one XCD gather from (window_rows x nc x 8 B: 8 MiB for |j-i| <= 4096) does not fit the XCD's 4 MiB L2 and
// two thirds of the gathers miss it.  Here the panel is cut into chunks of CC = 2*LPR columns and every XCD walks its
// row range once per chunk (blocks are dealt to the XCDs round-robin and, per XCD, in launch order: chunk-major), so the
// window is window_rows x CC x 8 B and the gathers hit L2.  LPR lanes own a row (4 rows per wave at CC = 32); the block's
// (col, val) run is staged in LDS once so that the per-nonzero loads are LDS broadcasts instead of vector-memory loads.
template <int LPR, int RPG>
__global__ __launch_bounds__(256) void k_spmm_rowgather_cc(int64_t m, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           const double *__restrict__ val, const double *__restrict__ X, int ldx,
                                                           const double *__restrict__ Xg, int ldg, double *__restrict__ Y, int ldy, int nc,
                                                           int64_t blocks_per_xcd, int lds_cap, int y_vec, int64_t mc)
{
    constexpr int GROUPS = 256 / LPR;
    constexpr int ROWS = GROUPS * RPG;
    constexpr int CC = 2 * LPR;
    constexpr int U = 8;
    extern __shared__ double smem[];
    double *s_val = smem;
    int32_t *s_col = reinterpret_cast<int32_t *>(smem + lds_cap);
    const int g = threadIdx.x / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t s = blockIdx.x >> 3;
    const int chunk = (int)(s / blocks_per_xcd);
    const int64_t lb = (int64_t)(blockIdx.x & 7) * blocks_per_xcd + (s - (int64_t)chunk * blocks_per_xcd);
    const int64_t r0 = lb * ROWS;
    if (r0 >= m) return;
    const int64_t r1 = (r0 + ROWS < m) ? r0 + ROWS : m;
    const int64_t nz0 = rowptr[r0], nz1 = rowptr[r1];
    const bool staged = (nz1 - nz0) <= lds_cap; // block-uniform
    if (staged) {
        for (int64_t q = nz0 + threadIdx.x; q < nz1; q += 256) {
            s_col[q - nz0] = col[q];
            s_val[q - nz0] = val[q];
        }
        __syncthreads();
    }
    const int cb = chunk * CC + l * 2;
    if (cb >= nc) return;
    const bool full = (cb + 2 <= nc);
    for (int rr = 0; rr < RPG; ++rr) {
        const int64_t row = r0 + (int64_t)rr * GROUPS + g; // the 256/LPR rows in flight are consecutive
        if (row >= m) break;
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        double2_t acc = (double2_t){0.0, 0.0};
        for (int64_t p = p0; p < p1; p += U) {
            int32_t c[U];
            double a[U];
            const int cnt = (int)((p1 - p) < U ? (p1 - p) : U);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = u < cnt;
                const int64_t q = ok ? p + u : p0;
                c[u] = staged ? s_col[q - nz0] : col[q];
                const double av = staged ? s_val[q - nz0] : val[q];
                a[u] = ok ? av : 0.0;
            }
            if (full) {
                double2_t x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double *src = (c[u] < mc) ? (X + (int64_t)c[u] * ldx + cb) : (Xg + ((int64_t)c[u] - mc) * ldg + cb);
                    x[u] = *reinterpret_cast<const double2_t *>(src);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc.x = __builtin_fma(a[u], x[u].x, acc.x);
                    acc.y = __builtin_fma(a[u], x[u].y, acc.y);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double *src = (c[u] < mc) ? (X + (int64_t)c[u] * ldx + cb) : (Xg + ((int64_t)c[u] - mc) * ldg + cb);
                    acc.x = __builtin_fma(a[u], *src, acc.x);
                }
            }
        }
        double *dst = Y + row * ldy + cb;
        if (full && y_vec)
            *reinterpret_cast<double2_t *>(dst) = acc;
        else if (full) { // Y window starts on an odd column: two 8-byte stores
            dst[0] = acc.x;
            dst[1] = acc.y;
        } else
            *dst = acc.x;
    }
}

// Kernel 1c: the in-loop product A * W at Expand size <= 16 on operators whose X rows are all addressable as X + c * ldx with 32-bit
// byte offsets (no ghost rows, panel below 4 GiB, fewer than 2^24 rows).  Same scheme as kernel 1b with one chunk -- 8 lanes own a row,
// the block's (col, val) run staged in LDS -- but the per-nonzero work is cut to what the product needs: two LDS broadcasts, one
// 24-bit multiply-add for the byte offset, one 16-byte load with a scalar base, two multiply-adds.  Kernel 1b spends ~37 vector
// instructions per nonzero on 64-bit addresses, the ghost-row select and the masks of its tail and ran at the instruction rate
// (rocprofv3 --pmc: 2000 VALU instructions per wave of 16 rows, TA and L2 far from busy): 0.40 ms at 16 columns, 18 % of the HBM rate.
template <int RPG, bool GHOST, int LPR = 8>
__global__ __launch_bounds__(256) void k_spmm_narrow(int64_t m, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     const double *__restrict__ val, const double *__restrict__ X, uint32_t ldx8,
                                                     const double *__restrict__ Xg, uint32_t ldg8, double *__restrict__ Y, int ldy, int nc,
                                                     int64_t blocks_per_xcd, int y_vec, int64_t mc)
{
    constexpr int GROUPS = 256 / LPR, ROWS = GROUPS * RPG, CAP = 2048; // LPR lanes own a row: 16 columns with 8, 32 with 16
    __shared__ double s_val[CAP];
    __shared__ int32_t s_col[CAP];
    const int g = threadIdx.x / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t lb = (int64_t)(blockIdx.x & 7) * blocks_per_xcd + (blockIdx.x >> 3);
    const int64_t r0 = lb * ROWS;
    if (r0 >= m) return;
    const int64_t r1 = (r0 + ROWS < m) ? r0 + ROWS : m;
    const int64_t nz0 = rowptr[r0], nz1 = rowptr[r1];
    const bool staged = (nz1 - nz0) <= CAP; // block-uniform
    if (staged) {
        for (int q = threadIdx.x; q < (int)(nz1 - nz0); q += 256) {
            s_col[q] = col[nz0 + q];
            s_val[q] = val[nz0 + q];
        }
        __syncthreads();
    }
    const int cb = l * 2;
    if (cb >= nc) return;
    const bool full = (cb + 2 <= nc);
    const uint32_t cb8 = (uint32_t)cb * 8u, m32 = (uint32_t)mc;
    const char *Xb = reinterpret_cast<const char *>(X), *Gb = reinterpret_cast<const char *>(Xg);
    // the address of this lane's two columns of X row c: local rows from the panel, ghost rows (c >= m, row-partitioned runs) from the
    // buffer the halo exchange filled -- a select between two bases and two strides, still 32-bit offsets
    auto xrow = [&](uint32_t c) -> const char * {
        if (!GHOST) return Xb + (__umul24(c, ldx8) + cb8);
        const bool local = c < m32;
        return (local ? Xb : Gb) + ((local ? __umul24(c, ldx8) : __umul24(c - m32, ldg8)) + cb8);
    };
    for (int rr = 0; rr < RPG; ++rr) {
        const int64_t row = r0 + (int64_t)rr * GROUPS + g; // the 32 rows in flight are consecutive
        if (row >= m) break;
        double2_t acc = (double2_t){0.0, 0.0};
        if (staged && full) {
            int i = (int)(rowptr[row] - nz0);
            const int i1 = (int)(rowptr[row + 1] - nz0);
            for (; i + 8 <= i1; i += 8) {
                double2_t x[8];
                double a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x[u] = *reinterpret_cast<const double2_a8_t *>(xrow((uint32_t)s_col[i + u]));
                    a[u] = s_val[i + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc.x = __builtin_fma(a[u], x[u].x, acc.x);
                    acc.y = __builtin_fma(a[u], x[u].y, acc.y);
                }
            }
            for (; i < i1; i += 4) { // the last one to seven: groups of four, the slots past the end repeat the last entry with a zero
                double2_t x[4];
                double a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = i + u < i1 ? i + u : i1 - 1;
                    x[u] = *reinterpret_cast<const double2_a8_t *>(xrow((uint32_t)s_col[q]));
                    a[u] = i + u < i1 ? s_val[q] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc.x = __builtin_fma(a[u], x[u].x, acc.x);
                    acc.y = __builtin_fma(a[u], x[u].y, acc.y);
                }
            }
        } else {
            // (a block with more nonzeros than the LDS buffer holds, or the lane of an odd last column: one entry at a time)
            for (int64_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
                const double a = val[p];
                const double *src = reinterpret_cast<const double *>(xrow((uint32_t)col[p]));
                acc.x = __builtin_fma(a, src[0], acc.x);
                if (full) acc.y = __builtin_fma(a, src[1], acc.y);
            }
        }
        double *dst = Y + row * ldy + cb;
        if (full && y_vec)
            *reinterpret_cast<double2_t *>(dst) = acc;
        else if (full) { // Y window starts on an odd column: two 8-byte stores
            dst[0] = acc.x;
            dst[1] = acc.y;
        } else
            *dst = acc.x;
    }
}

template <int LPR>
int launch_rg_cc(rails_ctx *c, const rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, bool y_vec = true,
                 const char **kernel = nullptr, bool narrow_only = false, const RowSpan &sp = RowSpan(), int want_ghost = -1)
{
    // want_ghost: -1 = the operator's own form; 0 / 1 = a span whose rows have no / may have ghost columns
    const int64_t m_rows = sp.nrows < 0 ? A->m : sp.nrows;
    hipStream_t st = sp.st ? sp.st : c->stream;
    const int64_t *rowptr = A->rowptr + sp.r0;
    Y += sp.r0 * ldy;
    if (kernel) *kernel = "k_spmm_rowgather_cc";
    constexpr int GROUPS = 256 / LPR;
    constexpr int RPG = (LPR >= 32) ? 8 : (LPR >= 16 ? 4 : 2);
    constexpr int ROWS = GROUPS * RPG; // 64 rows per block
    constexpr int CC = 2 * LPR;
    const int nchunks = (nc + CC - 1) / CC;
    const int64_t blocks = (m_rows + ROWS - 1) / ROWS;
    const int64_t bpx = (blocks + 7) / 8;
    const int64_t grid = bpx * 8 * nchunks;
    RAILS_REQUIRE(grid <= 0x7fffffffLL, "rails_spmm: grid too large");
    const int lds_cap = 2048; // nonzeros of one block staged in LDS (24 KiB); longer runs read (col, val) from global memory
    if ((LPR == 8 || LPR == 16) && nchunks == 1) {
        // every X row at X + c * ldx (no ghost rows; a rectangular operator's extra rows follow X in the same panel) within 32-bit byte offsets
        static const int narrow_fast = spmm_env("RAILS_SPMM_NARROW_FAST", 1);
        const bool flat = want_ghost == 0 || (want_ghost < 0 && ((A->n_ghost == 0 && !A->rect) || (Xg == X + (int64_t)A->m * ldx && ldg == ldx)));
        const bool small = A->ncols_ext < (1 << 24) && (int64_t)ldx * 8 < (1 << 24) && (int64_t)ldg * 8 < (1 << 24);
        if (narrow_fast && small && flat && (uint64_t)A->ncols_ext * (uint64_t)ldx * 8u < 0xffffff00ull) {
            RAILS_LAUNCH_ON(st, (k_spmm_narrow<RPG, false, (LPR == 16 ? 16 : 8)>), dim3((unsigned)grid), dim3(256), 0, m_rows, rowptr, A->col, A->val, X, (uint32_t)ldx * 8u, X, 0u, Y,
                            ldy, nc, bpx, y_vec ? 1 : 0, A->m);
            if (kernel) *kernel = "k_spmm_narrow";
            return RAILS_OK;
        }
        // row-partitioned runs: columns >= m are ghost rows in the halo buffer (16-byte aligned rows there too)
        if (narrow_fast && small && !flat && !A->rect && (uint64_t)A->m * (uint64_t)ldx * 8u < 0xffffff00ull &&
            (uint64_t)(A->ncols_ext - A->m) * (uint64_t)ldg * 8u < 0xffffff00ull ) {
            RAILS_LAUNCH_ON(st, (k_spmm_narrow<RPG, true, (LPR == 16 ? 16 : 8)>), dim3((unsigned)grid), dim3(256), 0, m_rows, rowptr, A->col, A->val, X, (uint32_t)ldx * 8u, Xg,
                            (uint32_t)ldg * 8u, Y, ldy, nc, bpx, y_vec ? 1 : 0, A->m);
            if (kernel) *kernel = "k_spmm_narrow";
            return RAILS_OK;
        }
    }
    if (narrow_only) { // nothing launched: the caller goes on to the plain row-gather kernel
        if (kernel) *kernel = nullptr;
        return RAILS_OK;
    }
    RAILS_LAUNCH_ON(st, (k_spmm_rowgather_cc<LPR, RPG>), dim3((unsigned)grid), dim3(256), (size_t)lds_cap * 12, m_rows, rowptr,
                    A->col, A->val, X, ldx, Xg, ldg, Y, ldy, nc, bpx, lds_cap, y_vec ? 1 : 0, A->m);
    return RAILS_OK;
}

// Columns per chunk for the row-gather kernel; 0 = whole width.  Measured on MI355X at nc = 128 (profiles/r01_spmm_chunked.md):
// chunking removes the L2 misses as intended (banded |j-i| <= 4096: 18.7 GB -> 3.1 GB of L2 fills per product) but the
// product does not get faster (2.34 -> 2.6 ms): the 27.6 GB of gathered row segments are bounded by L2 -> L1 throughput
// (~12 TB/s), not by the fabric.  So auto = whole width; the chunked kernel stays selectable (variants 4/5, RAILS_SPMM_CHUNK).
int rowgather_chunk(const rails_csr *A, int nc)
{
    static const int chunk_env = spmm_env("RAILS_SPMM_CHUNK", 0);
    if (A->variant == 4 || A->variant == 5) return nc > 32 * (A->variant - 3) ? 32 * (A->variant - 3) : 0;
    if (A->variant == 3) return 0;
    if (chunk_env == 32 || chunk_env == 64) return nc > chunk_env ? chunk_env : 0;
    return 0;
}

template <int VEC>
int dispatch_rg(rails_ctx *c, const rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, const RowSpan &sp = RowSpan())
{
    int need = (nc + VEC - 1) / VEC;
    if (need >= 64) return launch_rg<64, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
    if (need > 16) return launch_rg<32, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
    if (need > 8) return launch_rg<16, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
    if (need > 4) return launch_rg<8, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
    if (need > 2) return launch_rg<4, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
    if (need > 1) return launch_rg<2, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
    return launch_rg<1, VEC>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
}

// The rows of a span with the row kernels (what the serial path below picks for the same width): bitwise the same product row by row,
// whichever launch a row belongs to.  ghost: the span's rows may have ghost columns (taken from Xg).
int spmm_span(rails_ctx *c, const rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, bool x_vec2, bool y_vec2,
              const RowSpan &sp, bool ghost, const char **kname = nullptr)
{
    if (sp.nrows <= 0) return RAILS_OK;
    if (kname) *kname = "k_spmm_rowgather";
    static const int narrow_env = spmm_env("RAILS_SPMM_NARROW_CC", 1);
    if (x_vec2 && narrow_env && A->variant != 3 && nc > 8 && nc <= 32 && A->max_row_nnz <= 64) {
        const char *k = nullptr;
        if (nc <= 16)
            RAILS_TRY((launch_rg_cc<8>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, y_vec2, &k, false, sp, ghost ? 1 : 0)));
        else
            RAILS_TRY((launch_rg_cc<16>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, y_vec2, &k, false, sp, ghost ? 1 : 0)));
        if (kname) *kname = k;
        return RAILS_OK;
    }
    if (nc >= 2) return dispatch_rg<2>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
    return dispatch_rg<1>(c, A, X, ldx, Xg, ldg, Y, ldy, nc, sp);
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
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
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
    {
        const int64_t stride = std::max<int64_t>(1, m_local / 4096);
        int64_t sum = 0, cnt = 0;
        for (int64_t i = 0; i < m_local; i += stride) {
            if (rowptr[i + 1] == rowptr[i]) continue;
            int32_t lo = col[rowptr[i]], hi = lo;
            for (int64_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
                lo = std::min(lo, col[q]);
                hi = std::max(hi, col[q]);
            }
            sum += (int64_t)hi - lo + 1;
            cnt++;
        }
        A->window_rows = cnt ? sum / cnt : 0;
    }
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
    hipError_t ce = hipMemcpyAsync(A->rowptr, rowptr, (size_t)(m_local + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    if (ce == hipSuccess && nnz) ce = hipMemcpyAsync(A->col, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    if (ce == hipSuccess && nnz) ce = hipMemcpyAsync(A->val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (ce == hipSuccess) ce = rails_stream_sync(c);
    if (ce != hipSuccess) { // the half-made operator is released, not leaked
        rails_set_error("rails_csr_create: upload failed: %s", hipGetErrorString(ce));
        rails_csr_destroy(A);
        return RAILS_EHIP;
    }
    *out = A;
    return RAILS_OK;
}

extern "C" int rails_csr_create_callback(rails_ctx *c, int64_t m_local, rails_apply_fn fn, void *user, rails_csr **out)
{
    RAILS_REQUIRE(c && out && fn && m_local >= 0, "rails_csr_create_callback: bad argument");
    rails_csr *A = new rails_csr();
    A->ctx = c;
    A->m = m_local;
    A->ncols_ext = m_local;
    A->apply_cb = fn;
    A->apply_user = user;
    A->last_kernel = "callback";
    *out = A;
    return RAILS_OK;
}

extern "C" int rails_csr_destroy(rails_csr *A)
{
    if (!A) return RAILS_OK;
    hipStreamSynchronize(A->ctx->stream);
    if (A->AT) rails_csr_destroy(A->AT);
    rails_sweep_release(A);
    rails_planes_release(A);
    if (A->rowptr) hipFree(A->rowptr);
    if (A->col) hipFree(A->col);
    if (A->val) hipFree(A->val);
    if (A->send_rows) hipFree(A->send_rows);
    if (A->send_buf) hipFree(A->send_buf);
    if (A->ext) hipFree(A->ext);
    if (A->t_fp_ptr) hipFree(A->t_fp_ptr);
    if (A->t_fp) hipFree(A->t_fp);
    if (A->t_lcol) hipFree(A->t_lcol);
    if (A->t_rowptr) hipFree(A->t_rowptr);
    if (A->t_rows) hipFree(A->t_rows);
    if (A->t_nzptr) hipFree(A->t_nzptr);
    if (A->t_rp) hipFree(A->t_rp);
    if (A->t_val) hipFree(A->t_val);
    if (A->t_fpos) hipFree(A->t_fpos);
    delete A;
    return RAILS_OK;
}

extern "C" int64_t rails_csr_rows(const rails_csr *A) { return A ? A->m : -1; }
extern "C" int64_t rails_csr_nnz(const rails_csr *A) { return A ? A->nnz : -1; }
extern "C" const char *rails_csr_last_kernel(const rails_csr *A) { return A ? A->last_kernel : ""; }

// n_rows x n_cols with all columns local: X of a product has n_cols rows, Y n_rows.  The blocks A12, A21 of a Schur complement
// (src/SchurOperator.cpp:181-214) are of this kind.  Plain row-gather kernels only (the tile and sweep plans assume a square operator).
extern "C" int rails_csr_create_rect(rails_ctx *c, int64_t n_rows, int64_t n_cols, const int64_t *rowptr, const int32_t *col, const double *val,
                                     rails_csr **out)
{
    RAILS_REQUIRE(n_cols >= 1, "rails_csr_create_rect: no columns");
    RAILS_TRY(rails_csr_create(c, n_rows, n_cols, rowptr, col, val, out));
    (*out)->rect = true;
    (*out)->variant = 1;
    return RAILS_OK;
}

extern "C" int rails_csr_set_variant(rails_csr *A, int variant)
{
    RAILS_REQUIRE(A && variant >= 0 && variant <= 9, "rails_csr_set_variant: bad argument");
    if (A->rect) return RAILS_OK; // rectangular operators stay on the plain row-gather kernel
    A->variant = variant;
    return RAILS_OK;
}

extern "C" int rails_csr_set_halo_counts(rails_csr *A, int nranks, const int64_t *send_counts, const int64_t *recv_counts)
{
    RAILS_REQUIRE(A && nranks >= 1 && send_counts && recv_counts, "rails_csr_set_halo_counts: bad argument");
    int64_t ns = 0, nr = 0;
    for (int r = 0; r < nranks; ++r) {
        RAILS_REQUIRE(send_counts[r] >= 0 && recv_counts[r] >= 0, "rails_csr_set_halo_counts: negative count");
        ns += send_counts[r];
        nr += recv_counts[r];
    }
    RAILS_REQUIRE(ns == A->n_send && nr == A->n_ghost, "rails_csr_set_halo_counts: counts add up to %lld sent / %lld received rows, the plan has %lld / %lld",
                  (long long)ns, (long long)nr, (long long)A->n_send, (long long)A->n_ghost);
    A->send_counts.assign(send_counts, send_counts + nranks);
    A->recv_counts.assign(recv_counts, recv_counts + nranks);
    return RAILS_OK;
}

extern "C" int rails_csr_set_halo(rails_csr *A, int64_t n_send, const int64_t *send_rows, int64_t n_ghost, rails_halo_fn fn,
                                  void *user)
{
    RAILS_REQUIRE(A, "null operator");
    RAILS_REQUIRE(n_send >= 0 && n_ghost >= 0 && A->m + n_ghost == A->ncols_ext,
                  "rails_csr_set_halo: m_local %lld + ghosts %lld != extended columns %lld", (long long)A->m, (long long)n_ghost,
                  (long long)A->ncols_ext);
    RAILS_REQUIRE((n_send == 0 && n_ghost == 0) || fn || A->ctx->rccl,
                  "rails_csr_set_halo: neither a halo hook nor an RCCL communicator on the context (rails_ctx_init_rccl)");
    for (int64_t i = 0; i < n_send; ++i)
        RAILS_REQUIRE(send_rows[i] >= 0 && send_rows[i] < A->m, "rails_csr_set_halo: send row %lld out of range", (long long)send_rows[i]);
    rails_ctx *c = A->ctx;
    if (A->send_rows) {
        RAILS_HIP_CHECK(rails_stream_sync(c));
        RAILS_HIP_CHECK(hipFree(A->send_rows));
        A->send_rows = nullptr;
    }
    A->n_send = n_send;
    A->n_ghost = n_ghost;
    A->halo = fn;
    A->halo_user = user;
    // the interior rows: the contiguous range in the middle of the block whose rows reference no ghost column (row blocks of banded and
    // grid operators keep their boundary rows at the two ends).  Their product does not wait for the exchange (rails_spmm).
    A->int_lo = 0;
    A->int_hi = A->m;
    if (n_ghost > 0) {
        const int64_t m = A->m, mid = m / 2;
        for (int64_t r = 0; r < m; ++r) {
            bool boundary = false;
            for (int64_t p2 = A->h_rowptr[r]; p2 < A->h_rowptr[r + 1] && !boundary; ++p2) boundary = A->h_col[p2] >= m;
            if (!boundary) continue;
            if (r < mid)
                A->int_lo = r + 1;
            else {
                A->int_hi = r;
                break;
            }
        }
        // the sliding window of the interior rows alone (the boundary rows' ghost columns sit behind the local ones: they would inflate it)
        const int64_t nint = A->int_hi - A->int_lo, stride = std::max<int64_t>(1, nint / 4096);
        int64_t sum = 0, cnt = 0;
        for (int64_t r = A->int_lo; r < A->int_hi; r += stride) {
            if (A->h_rowptr[r + 1] == A->h_rowptr[r]) continue;
            int32_t lo = A->h_col[A->h_rowptr[r]], hi = lo;
            for (int64_t q = A->h_rowptr[r]; q < A->h_rowptr[r + 1]; ++q) {
                lo = std::min(lo, A->h_col[q]);
                hi = std::max(hi, A->h_col[q]);
            }
            sum += (int64_t)hi - lo + 1;
            cnt++;
        }
        A->window_rows_int = cnt ? sum / cnt : 0;
    }
    if (n_send) {
        RAILS_HIP_CHECK(hipMalloc((void **)&A->send_rows, (size_t)n_send * sizeof(int64_t)));
        RAILS_HIP_CHECK(hipMemcpy(A->send_rows, send_rows, (size_t)n_send * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    return RAILS_OK;
}

int rails_spmm_tiled(rails_ctx *c, rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, bool vec2,
                     int x_room, bool *done);

static bool rails_csr_is_grid(rails_csr *A); // structured-grid stencil (the LDS-staged box kernel's territory), looked at once

// The sweep kernel as the automatic choice, from what is known without building its schedule: 64 to 256 columns in chunks of 16 that
// divide the 32 workgroups of an XCD; the window of columns of a row fits (phases - 1) blocks of 2816 rows (the planner decides
// exactly); every XCD's part holds a few blocks per phase; an X row is staged by at most 8 workgroups per row of the part (phases x
// (1 + window / rows of a part): the launcher's own bound); and the pattern is not a structured-grid stencil -- few nonzeros per row
// leave the sweep at its floor of one LDS-DMA latency per step and the LDS-staged box kernel is faster (7-point Laplacian 50 x 50 x
// 400 at 128 columns: 0.65 ms against 0.82).
static bool sweep_worthwhile(rails_csr *A, int nc)
{
    if (!(nc >= 64 && nc <= 256 && nc % 16 == 0 && 32 % (nc / 16) == 0 && A->n_ghost == 0 && A->window_rows > 0)) return false;
    const int64_t phases = 32 / (nc / 16), part_rows = A->m / 8;
    if (!(A->window_rows + 256 <= (phases - 1) * 2816 && A->m >= 8 * phases * 2816 &&
          (double)phases * (double)(part_rows + A->window_rows + 1024) <= 8.0 * (double)part_rows))
        return false;
    return !rails_csr_is_grid(A);
}

extern "C" int rails_csr_prepare(rails_ctx *c, rails_csr *A, int trans, int nc, int *kernel_ready)
{
    RAILS_REQUIRE(c && A && nc >= 1, "rails_csr_prepare: bad argument");
    if (kernel_ready) *kernel_ready = 0;
    if (A->apply_cb) return RAILS_OK;
    if (trans) {
        RAILS_TRY(build_transpose(A));
        A->AT->variant = A->variant;
        return rails_csr_prepare(c, A->AT, 0, nc, kernel_ready);
    }
    if (A->variant == 7 || (A->variant == 0 && sweep_worthwhile(A, nc))) {
        bool fits = false;
        RAILS_TRY(rails_sweep_prepare(c, A, nc, &fits));
        if (kernel_ready) *kernel_ready = fits ? 1 : 0;
    } else if (A->variant == 0 && A->n_ghost > 0 && !rails_csr_is_grid(A)) { // row-partitioned: the schedule of the interior rows
        bool fits = false;
        RAILS_TRY(rails_sweep_prepare_interior(c, A, nc, &fits));
        if (kernel_ready) *kernel_ready = fits ? 1 : 0;
    }
    return RAILS_OK;
}

extern "C" int rails_spmm(rails_ctx *c, rails_csr *A, int trans, const rails_panel *X, int xc0, int nc, rails_panel *Y, int yc0)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    rails_slow_guard slow__(c, "rails_spmm", nc, A ? A->m : 0);
    RAILS_REQUIRE(c && A && X && Y, "rails_spmm: null argument");
    RAILS_REQUIRE(xc0 >= 0 && nc >= 0 && xc0 + nc <= X->cap, "rails_spmm: X columns [%d,%d) outside capacity %d", xc0, xc0 + nc, X->cap);
    RAILS_REQUIRE(yc0 >= 0 && yc0 + nc <= Y->cap, "rails_spmm: Y columns [%d,%d) outside capacity %d", yc0, yc0 + nc, Y->cap);
    RAILS_REQUIRE(X->m == (A->rect ? A->ncols_ext : A->m) && Y->m == A->m, "rails_spmm: operator is %lld x %lld, X has %lld rows, Y %lld", (long long)A->m,
                  (long long)(A->rect ? A->ncols_ext : A->m), (long long)X->m, (long long)Y->m);
    RAILS_REQUIRE(!(A->rect && trans), "rails_spmm: a rectangular operator has no transposed apply (create the transposed matrix)");
    if (X->d == Y->d) RAILS_REQUIRE(xc0 + nc <= yc0 || yc0 + nc <= xc0, "rails_spmm: X and Y windows alias");
    if (nc == 0 || A->m == 0) return RAILS_OK;
    if (A->apply_cb) {
        int rc = A->apply_cb(A->apply_user, trans ? 1 : 0, X, xc0, nc, Y, yc0);
        if (rc != 0) {
            rails_set_error("rails_spmm: the operator callback failed with code %d", rc);
            return RAILS_ECOMM;
        }
        c->n_spmm_callback++;
        return RAILS_OK;
    }
    if (trans) {
        RAILS_TRY(build_transpose(A));
        A->AT->variant = A->variant;
        int rc = rails_spmm(c, A->AT, 0, X, xc0, nc, Y, yc0);
        A->last_kernel = A->AT->last_kernel;
        return rc;
    }
    const double *Xp = X->d + xc0;
    double *Yp = Y->d + yc0;
    // columns beyond the operator's rows come from the ghost buffer; for a rectangular operator (more columns than rows, all of them
    // local) that buffer is the rest of X itself
    const double *Xg = A->rect ? Xp + (size_t)std::min(A->m, A->ncols_ext) * X->ld : Xp;
    int ldg = X->ld;
    if (A->n_ghost > 0 || A->n_send > 0) {
        // pack the rows the neighbours need, exchange, gather from [local | ghost]
        size_t sbytes = (size_t)(A->n_send ? A->n_send : 1) * nc * sizeof(double);
        size_t gbytes = (size_t)(A->n_ghost ? A->n_ghost : 1) * nc * sizeof(double);
        if (sbytes > A->send_cap) {
            RAILS_HIP_CHECK(rails_stream_sync(c));
            if (A->send_buf) RAILS_HIP_CHECK(hipFree(A->send_buf));
            A->send_buf = nullptr;
            RAILS_HIP_CHECK(hipMalloc((void **)&A->send_buf, sbytes));
            A->send_cap = sbytes;
        }
        if (gbytes > A->ext_cap) {
            RAILS_HIP_CHECK(rails_stream_sync(c));
            if (A->ext) RAILS_HIP_CHECK(hipFree(A->ext));
            A->ext = nullptr;
            RAILS_HIP_CHECK(hipMalloc((void **)&A->ext, gbytes));
            A->ext_cap = gbytes;
        }
        // Overlap: the interior rows' product runs on the context's second stream while the ghost rows are packed, exchanged and
        // waited for on the first; the boundary rows follow the exchange, and the first stream then waits for the interior.  Row by
        // row the same kernels as the serial order below (`RAILS_SPMM_HALO_OVERLAP=0`): the result is bitwise the same.
        const int overlap_env = spmm_env("RAILS_SPMM_HALO_OVERLAP", 1); // (read per product: the tests switch it inside one process)
        const bool x_vec2o = (xc0 & 1) == 0 && (X->ld % 2 == 0) && (nc % 2 == 0), y_vec2o = (yc0 & 1) == 0 && (Y->ld % 2 == 0);
        bool interior_sweep = false;
        const bool overlap = overlap_env && A->n_ghost > 0 && !A->rect && (A->variant == 0 || A->variant == 1 || A->variant == 3) &&
                             A->int_hi - A->int_lo >= A->m / 2 && A->int_hi - A->int_lo >= 1024;
        if (overlap) {
            RAILS_TRY(rails_ctx_second_stream(c));
            RAILS_HIP_CHECK(hipEventRecord(c->ev_fork, c->stream));
            RAILS_HIP_CHECK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            RowSpan in;
            in.r0 = A->int_lo;
            in.nrows = A->int_hi - A->int_lo;
            in.st = c->stream2;
            bool planes_done = false, sweep_done = false;
            RAILS_TRY(rails_spmm_planes_interior(c, A, Xp, X->ld, Yp, Y->ld, nc, x_vec2o, c->stream2, &planes_done));
            if (!planes_done) RAILS_TRY(rails_spmm_sweep_interior(c, A, Xp, X->ld, Yp, Y->ld, nc, x_vec2o && y_vec2o, c->stream2, &sweep_done));
            interior_sweep = sweep_done;
            if (!planes_done && !sweep_done) RAILS_TRY(spmm_span(c, A, Xp, X->ld, Xp, X->ld, Yp, Y->ld, nc, x_vec2o, y_vec2o, in, false));
            RAILS_HIP_CHECK(hipEventRecord(c->ev_join, c->stream2));
            c->n_spmm_overlapped++;
        }
        if (A->n_send) {
            int64_t total = A->n_send * nc;
            int grid = (int)std::min<int64_t>((total + 255) / 256, (int64_t)c->num_cu * 8);
            RAILS_LAUNCH(k_pack_rows, dim3(grid), dim3(256), 0, c->stream, A->send_rows, A->n_send, Xp, X->ld, nc, A->send_buf);
        }
        if (A->halo) {
            int rc = A->halo(A->halo_user, A->send_buf, A->ext, nc, (void *)c->stream);
            if (rc != 0) {
                rails_set_error("rails_spmm: halo hook failed with code %d", rc);
                return RAILS_ECOMM;
            }
        } else {
            RAILS_REQUIRE(c->rccl, "rails_spmm: ghost rows but neither a halo hook nor an RCCL communicator");
            RAILS_TRY(rails_rccl_halo(c, A, A->send_buf, A->ext, nc));
        }
        if (overlap) {
            const bool gvec = x_vec2o; // (ghost rows are packed with ld = nc: even, 16-byte aligned rows when nc is even)
            RowSpan lo, hi;
            lo.r0 = 0;
            lo.nrows = A->int_lo;
            hi.r0 = A->int_hi;
            hi.nrows = A->m - A->int_hi;
            const char *kb = "k_spmm_rowgather";
            RAILS_TRY(spmm_span(c, A, Xp, X->ld, A->ext, nc, Yp, Y->ld, nc, gvec, y_vec2o, lo, true, &kb));
            RAILS_TRY(spmm_span(c, A, Xp, X->ld, A->ext, nc, Yp, Y->ld, nc, gvec, y_vec2o, hi, true, &kb));
            RAILS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            // (what ran: the kernel of the boundary rows, and of the interior rows where that is another one)
            const bool narrow = !strcmp(kb, "k_spmm_narrow");
            if (interior_sweep)
                A->last_kernel = "k_spmm_rowgather + k_spmm_sweep (halo overlapped)";
            else if (planes_last_interior(A))
                A->last_kernel = narrow ? "k_spmm_narrow + k_spmm_planes (halo overlapped)" : "k_spmm_rowgather + k_spmm_planes (halo overlapped)";
            else
                A->last_kernel = narrow ? "k_spmm_narrow (halo overlapped)" : (!strcmp(kb, "k_spmm_rowgather_cc") ? "k_spmm_rowgather_cc (halo overlapped)" : "k_spmm_rowgather (halo overlapped)");
            c->n_spmm_rowgather++;
            RAILS_HIP_CHECK(hipGetLastError());
            return RAILS_OK;
        }
        Xg = A->ext;
        ldg = nc;
    }
    bool done = false;
    const bool vec2 = ((xc0 | yc0) & 1) == 0 && (ldg % 2 == 0);
    // banded patterns at panel width: the sweep kernel (spmm_sweep.hip).  Auto (all of it known without building the schedule, which takes
    // a second per million rows): 64 to 256 columns in chunks of 16 that divide the 32 workgroups of an XCD; the window of columns of a
    // row fits (phases - 1) blocks of 2816 rows (the planner decides exactly); every XCD's part holds a few blocks per phase; and an X
    // row is staged by at most 8 workgroups per row of the part (phases x (1 + window / rows of a part): the launcher's own bound).
    const bool sweep_auto = A->variant == 0 && sweep_worthwhile(A, nc);
    if (A->variant == 7 || sweep_auto) {
        const bool al = ((xc0 | yc0) & 1) == 0 && X->ld % 2 == 0 && Y->ld % 2 == 0 && ldg % 2 == 0;
        RAILS_TRY(rails_spmm_sweep(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc, al, A->variant == 7, &done));
        if (done) return RAILS_OK;
    }
    // structured-grid stencils (complete 7- / 27-point patterns): the plane-sweep kernel (spmm_planes.hip) at every even width
    // -- its plan is one pass over the matrix on the device (a product's worth of time), made by the first product that asks
    if (A->variant == 9 || (A->variant == 0 && nc >= 2 && A->n_ghost == 0 && !A->rect && spmm_env("RAILS_SPMM_PLANES", 1) && rails_csr_is_grid(A))) {
        const bool al = (xc0 & 1) == 0 && X->ld % 2 == 0; // (16-byte aligned X rows for the LDS-DMA; Y's window may sit on an odd column)
        RAILS_TRY(rails_spmm_planes(c, A, Xp, X->ld, Yp, Y->ld, nc, al, true, &done));
        if (done) {
            c->n_spmm_planes++;
            return RAILS_OK;
        }
        RAILS_REQUIRE(A->variant != 9, "rails_spmm: plane-sweep kernel requested but not applicable to this operator/shape");
    }
    // at Expand size <= 16 the lean row kernel (1c) beats the LDS-staged box kernel on the stencils too (27-point: 0.206 against 0.256 ms,
    // 7-point: 0.097 against 0.130): in automatic mode the box kernel is for wider panels
    static const int narrow_first = spmm_env("RAILS_SPMM_NARROW_CC", 1) && spmm_env("RAILS_SPMM_NARROW_FAST", 1);
    const bool leave_to_narrow = A->variant == 0 && narrow_first && nc > 8 && nc <= 16 && A->max_row_nnz <= 64 && A->ncols_ext < (1 << 24);
    if ((A->variant == 0 && !leave_to_narrow) || A->variant == 2 || A->variant == 6)
        RAILS_TRY(rails_spmm_tiled(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc, vec2, X->ld - xc0, &done));
    if (done) c->n_spmm_tiled++;
    if (!done) {
        RAILS_REQUIRE(A->variant != 2 && A->variant != 6, "rails_spmm: LDS-staged kernel requested but not applicable to this operator/shape");
        int cc = vec2 ? rowgather_chunk(A, nc) : 0;
        // narrow panels (the in-loop A*W at Expand size <= 32): one chunk, (col, val) of a 64-row block staged in LDS instead of
        // per-lane vector-memory loads -- kernel 1c where every X row is within 32-bit byte offsets (0.24 vs 0.45 ms at 16 columns,
        // 0.44 vs 0.71 ms at 32, banded pattern), kernel 1b otherwise (RAILS_SPMM_NARROW_CC=0 disables both)
        static const int narrow_env = spmm_env("RAILS_SPMM_NARROW_CC", 1);
        // (the gathers need 16-byte aligned X rows; a Y window on an odd column -- A*W written behind an odd number of basis columns --
        // only changes the form of the stores)
        const bool x_vec2 = (xc0 & 1) == 0 && (X->ld % 2 == 0) && (ldg % 2 == 0);
        const bool y_vec2 = (yc0 & 1) == 0 && (Y->ld % 2 == 0);
        if (cc == 0 && x_vec2 && narrow_env && A->variant != 3 && nc > 8 && nc <= 32 && A->max_row_nnz <= 64) cc = nc <= 16 ? 16 : -32;
        // X rows that are only 8-byte aligned (a window on an odd column: A*W of the direct back end, W a view of V): kernel 1c loads its
        // two doubles with one 16-byte instruction all the same (global memory asks for dword alignment); kernel 1b does not take them
        bool narrow_only = false;
        if (cc == 0 && !x_vec2 && narrow_env && A->variant != 3 && nc > 8 && nc <= 16 && A->max_row_nnz <= 64) cc = 16, narrow_only = true;
        const char *cc_kernel = "k_spmm_rowgather_cc";
        bool launched = false;
        if (cc == 16) {
            RAILS_TRY((launch_rg_cc<8>(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc, y_vec2, &cc_kernel, narrow_only)));
            launched = cc_kernel != nullptr;
            if (!launched) cc = 0; // (declined: kernel 1c does not apply and kernel 1b needs 16-byte aligned rows)
        }
        if (launched)
            ;
        else if (cc == -32)
            RAILS_TRY((launch_rg_cc<16>(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc, y_vec2, &cc_kernel)));
        else if (cc == 32)
            RAILS_TRY((launch_rg_cc<16>(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc)));
        else if (cc == 64)
            RAILS_TRY((launch_rg_cc<32>(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc)));
        else if (vec2 || nc >= 2) // (two columns per lane whatever the alignment of the windows: see Acc<2>)
            RAILS_TRY((dispatch_rg<2>(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc)));
        else
            RAILS_TRY((dispatch_rg<1>(c, A, Xp, X->ld, Xg, ldg, Yp, Y->ld, nc)));
        if (cc) {
            A->last_kernel = cc_kernel;
            c->n_spmm_rowgather++;
            RAILS_HIP_CHECK(hipGetLastError());
            return RAILS_OK;
        }
        A->last_kernel = "k_spmm_rowgather";
        c->n_spmm_rowgather++;
    }
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

// -----------------------------------------------------------------------------------------
// Kernel 2: LDS-staged footprint kernel (k_spmm_tiled).
//
// Rows are grouped into tiles.  The set of X rows a tile touches (its column footprint, sorted) is computed once per
// operator on the host; every nonzero gets a 16-bit index into its tile's footprint and the tile's (val, index)
// pairs are stored tile-major.  A workgroup owns one tile: it copies the tile's CSR block into LDS once, then per
// chunk of KC columns stages footprint x KC doubles of X in LDS (coalesced 64/128-B row segments) and every row of
// the tile accumulates from LDS.  Each X row segment crosses the L2->CU fabric once per tile instead of once per
// nonzero, and the inner loop has no global loads.
//
// Tiles: if the matrix is a structured-grid stencil in natural ordering (detected from the column offsets of a
// sample of rows: 5/7/9/27-point patterns), tiles are bx x by x bz boxes of grid points, whose footprint is
// (bx+2)(by+2)(bz+2) -- 9.6 uses per staged row for an 8x4x4 box of a 27-point stencil.  Otherwise tiles are runs of
// consecutive rows (banded matrices, ~2.8 uses per staged row for a 27-point pattern) and the kernel is used only
// when a staged row is used about twice or more.
// -----------------------------------------------------------------------------------------
namespace {

template <int KC>
__global__ __launch_bounds__(256) void k_spmm_tiled(int64_t m, int64_t ntiles, const int32_t *__restrict__ t_rowptr /* tile row starts into t_rows */,
                                                    const int32_t *__restrict__ t_rows, const int64_t *__restrict__ t_nzptr,
                                                    const int32_t *__restrict__ t_rp /* per tile-row nz offsets, local to the tile */,
                                                    const double *__restrict__ t_val, const uint16_t *__restrict__ t_lcol,
                                                    const int32_t *__restrict__ fp_ptr, const int32_t *__restrict__ fp, const uint16_t *__restrict__ fpos,
                                                    const double *__restrict__ X, int ldx, const double *__restrict__ Xg, int ldg,
                                                    double *__restrict__ Y, int ldy, int nc, int64_t tiles_per_xcd, int nz_cap, int xs_doubles)
{
    extern __shared__ double smem[];
    constexpr int LPR = KC / 2;    // lanes per row, 2 doubles (16 B) each
    constexpr int RPP = 256 / LPR; // rows per pass of the workgroup
    // LDS carve-up (all 8-byte aligned): vals[nz_cap] | Xs[fp x KC] | rp[rows_cap+1] (int32) | lcols[nz_cap] (uint16)
    int64_t t = blockIdx.x;
    if (tiles_per_xcd > 0) t = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
    const int tr0 = t_rowptr[t];
    const int nrows = t_rowptr[t + 1] - tr0;
    const int64_t z0 = t_nzptr[t];
    const int nz = (int)(t_nzptr[t + 1] - z0);
    const int f0 = fp_ptr[t];
    const int nf = fp_ptr[t + 1] - f0;
    // LDS carve-up: vals[nz_cap] | Xs[xs_doubles] | rp[264] (int32) | lcols[nz_cap] (uint16)
    double *vals = smem;
    double *Xs = vals + nz_cap;
    int32_t *rp = reinterpret_cast<int32_t *>(Xs + xs_doubles);
    uint16_t *lcols = reinterpret_cast<uint16_t *>(rp + 264);
    const int tid = threadIdx.x;
    const int part = tid % LPR;

    for (int i = tid; i < nz; i += 256) {
        vals[i] = t_val[z0 + i];
        lcols[i] = t_lcol[z0 + i];
    }
    for (int i = tid; i <= nrows; i += 256) rp[i] = t_rp[tr0 + t + i]; // nrows+1 offsets per tile
    __syncthreads();

    for (int c0 = 0; c0 < nc; c0 += KC) {
        const int cidx = c0 + 2 * part;
        for (int idx = tid; idx < nf * LPR; idx += 256) {
            const int f = idx / LPR;
            const int32_t c = fp[f0 + f];
            const double *src = (c < m) ? (X + (int64_t)c * ldx) : (Xg + ((int64_t)c - m) * ldg);
            double2_t v = (double2_t){0.0, 0.0};
            if (cidx + 1 < nc)
                v = *reinterpret_cast<const double2_t *>(src + cidx);
            else if (cidx < nc)
                v.x = src[cidx];
            *reinterpret_cast<double2_t *>(&Xs[(int)fpos[f0 + f] * KC + 2 * part]) = v;
        }
        __syncthreads();
        for (int i = tid / LPR; i < nrows; i += RPP) {
            const int p0 = rp[i], p1 = rp[i + 1];
            double2_t acc = (double2_t){0.0, 0.0};
            int p = p0;
            for (; p + 4 <= p1; p += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double a = vals[p + u];
                    const double2_t x = *reinterpret_cast<const double2_t *>(&Xs[(int)lcols[p + u] * KC + 2 * part]);
                    acc.x = __builtin_fma(a, x.x, acc.x);
                    acc.y = __builtin_fma(a, x.y, acc.y);
                }
            }
            for (; p < p1; ++p) {
                const double a = vals[p];
                const double2_t x = *reinterpret_cast<const double2_t *>(&Xs[(int)lcols[p] * KC + 2 * part]);
                acc.x = __builtin_fma(a, x.x, acc.x);
                acc.y = __builtin_fma(a, x.y, acc.y);
            }
            double *dst = Y + (int64_t)t_rows[tr0 + i] * ldy + cidx;
            if (cidx + 1 < nc)
                *reinterpret_cast<double2_t *>(dst) = acc;
            else if (cidx < nc)
                *dst = acc.x;
        }
        __syncthreads();
    }
}

// Pipelined form of k_spmm_tiled: every thread keeps the source pointers of its NL staging slots in registers (the
// footprint is the same for every column chunk), the loads of chunk c+1 are in flight while chunk c is consumed from
// the other LDS buffer, one barrier per chunk.
template <int KC, int NL>
__global__ __launch_bounds__(256) void k_spmm_tiled_pipe(int64_t m, int64_t ntiles, const int32_t *__restrict__ t_rowptr,
                                                         const int32_t *__restrict__ t_rows, const int64_t *__restrict__ t_nzptr,
                                                         const int32_t *__restrict__ t_rp, const double *__restrict__ t_val,
                                                         const uint16_t *__restrict__ t_lcol, const int32_t *__restrict__ fp_ptr,
                                                         const int32_t *__restrict__ fp, const uint16_t *__restrict__ fpos, const double *__restrict__ X, int ldx,
                                                         const double *__restrict__ Xg, int ldg, double *__restrict__ Y, int ldy, int nc,
                                                         int64_t tiles_per_xcd, int nz_cap, int xs_doubles)
{
    extern __shared__ double smem[];
    constexpr int LPR = KC / 2;
    constexpr int RPP = 256 / LPR;
    int64_t t = blockIdx.x;
    if (tiles_per_xcd > 0) t = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
    const int tr0 = t_rowptr[t];
    const int nrows = t_rowptr[t + 1] - tr0;
    const int64_t z0 = t_nzptr[t];
    const int nz = (int)(t_nzptr[t + 1] - z0);
    const int f0 = fp_ptr[t];
    const int nf = fp_ptr[t + 1] - f0;
    // LDS: vals[nz_cap] | Xs0[xs_doubles] | Xs1[xs_doubles] | rp[264] (int32) | lcols[nz_cap] (uint16)
    double *vals = smem;
    double *Xs0 = vals + nz_cap;
    int32_t *rp = reinterpret_cast<int32_t *>(Xs0 + 2 * (size_t)xs_doubles);
    uint16_t *lcols = reinterpret_cast<uint16_t *>(rp + 264);
    const int tid = threadIdx.x;
    const int part = tid % LPR;

    const double *srcp[NL];
    int dsto[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + 256 * i;
        srcp[i] = nullptr;
        dsto[i] = 0;
        if (idx < nf * LPR) {
            const int f = idx / LPR;
            const int32_t c = fp[f0 + f];
            srcp[i] = ((c < m) ? (X + (int64_t)c * ldx) : (Xg + ((int64_t)c - m) * ldg)) + 2 * part;
            dsto[i] = (int)fpos[f0 + f] * KC + 2 * part;
        }
    }
    // D column chunks are in flight in registers; chunk ci is written to LDS buffer (ci & 1) just before it is
    // consumed, and its register slot is refilled with chunk ci + D.  One barrier per chunk.
    constexpr int D = 4;
    double2_t stage[D][NL];
    const int nchunks = (nc + KC - 1) / KC;
#define RAILS_LOAD_CHUNK(SLOT, CI)                                                              \
    do {                                                                                        \
        const int c0__ = (CI)*KC;                                                               \
        const int cidx__ = c0__ + 2 * part;                                                     \
        _Pragma("unroll") for (int i = 0; i < NL; ++i)                                          \
        {                                                                                       \
            stage[SLOT][i] = (double2_t){0.0, 0.0};                                             \
            if (srcp[i]) {                                                                      \
                if (cidx__ + 1 < nc)                                                            \
                    stage[SLOT][i] = *reinterpret_cast<const double2_t *>(srcp[i] + c0__);      \
                else if (cidx__ < nc)                                                           \
                    stage[SLOT][i].x = srcp[i][c0__];                                           \
            }                                                                                   \
        }                                                                                       \
    } while (0)

#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nchunks) RAILS_LOAD_CHUNK(d, d);
    for (int i = tid; i < nz; i += 256) {
        vals[i] = t_val[z0 + i];
        lcols[i] = t_lcol[z0 + i];
    }
    for (int i = tid; i <= nrows; i += 256) rp[i] = t_rp[tr0 + t + i];

    for (int cbase = 0; cbase < nchunks; cbase += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int ci = cbase + d;
            if (ci < nchunks) {
                double *Xs = Xs0 + (size_t)(ci & 1) * xs_doubles;
#pragma unroll
                for (int i = 0; i < NL; ++i)
                    if (srcp[i]) *reinterpret_cast<double2_t *>(&Xs[dsto[i]]) = stage[d][i];
                if (ci + D < nchunks) RAILS_LOAD_CHUNK(d, ci + D);
                __syncthreads();
                const int cidx = ci * KC + 2 * part;
                for (int i = tid / LPR; i < nrows; i += RPP) {
                    const int p0 = rp[i], p1 = rp[i + 1];
                    double2_t acc = (double2_t){0.0, 0.0};
                    int p = p0;
                    for (; p + 4 <= p1; p += 4) {
                        double a[4];
                        int lc[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            a[u] = vals[p + u];
                            lc[u] = lcols[p + u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double2_t x = *reinterpret_cast<const double2_t *>(&Xs[lc[u] * KC + 2 * part]);
                            acc.x = __builtin_fma(a[u], x.x, acc.x);
                            acc.y = __builtin_fma(a[u], x.y, acc.y);
                        }
                    }
                    for (; p < p1; ++p) {
                        const double a = vals[p];
                        const double2_t x = *reinterpret_cast<const double2_t *>(&Xs[(int)lcols[p] * KC + 2 * part]);
                        acc.x = __builtin_fma(a, x.x, acc.x);
                        acc.y = __builtin_fma(a, x.y, acc.y);
                    }
                    double *dst = Y + (int64_t)t_rows[tr0 + i] * ldy + cidx;
                    if (cidx + 1 < nc)
                        *reinterpret_cast<double2_t *>(dst) = acc;
                    else if (cidx < nc)
                        *dst = acc.x;
                }
            }
        }
    }
#undef RAILS_LOAD_CHUNK
}

// Register-resident form: one tile row per slot of KC/2 lanes; the row's (val, footprint index) pairs are loaded
// into registers once per tile and reused for every column chunk, so the inner loop is ONE ds_read_b128 + 2 FMA per
// nonzero (the LDS-resident forms above spend three LDS reads per nonzero and are LDS-issue bound).  X chunks are
// double-buffered in LDS, the next chunk's global loads are in flight during the current chunk's arithmetic.
//
// V2 = double2 vectors per lane: V2 = 1 gives KC/2 lanes per row (KC = 8: 4 lanes x 16 B); V2 = 2 with KC = 16 keeps 4 lanes
// per row (64-row tiles) with 32 B per lane, i.e. whole 128-B lines per staged X row, twice the bytes in flight per
// workgroup at the same register cost for the row's CSR, and half as many barrier phases.  The two 64-B halves of an LDS
// row are swapped when bit 1 of the row position is set, so the four x-consecutive row slots a 16-lane group of a
// ds_read_b128 serves still fall into four different bank quarters.
// NS = column chunks in flight in registers per thread (2, or 1 where the registers do not allow two).
template <int KC, int NNZ, int NL, int V2, int NS>
__global__ __launch_bounds__(256) void k_spmm_tiled_reg(int64_t m, int64_t ntiles, const int32_t *__restrict__ t_rowptr,
                                                        const int32_t *__restrict__ t_rows, const int64_t *__restrict__ t_nzptr,
                                                        const int32_t *__restrict__ t_rp, const double *__restrict__ t_val,
                                                        const uint16_t *__restrict__ t_lcol, const int32_t *__restrict__ fp_ptr,
                                                        const int32_t *__restrict__ fp, const uint16_t *__restrict__ fpos, const double *__restrict__ X, int ldx,
                                                        const double *__restrict__ Xg, int ldg, double *__restrict__ Y, int ldy, int nc,
                                                        int64_t tiles_per_xcd, int xs_doubles)
{
    // Every global load in this kernel is UNCONDITIONAL (clamped indices, duplicate staging slots, full-width chunks
    // guaranteed by the host): a load under a lane-dependent branch makes hipcc wait vmcnt(0) at the join, which
    // serialises the staging loads into dependent round trips (measured: 4 us per 14-KB chunk).
    extern __shared__ double smem[];
    constexpr int LPR = KC / (2 * V2); // compute lanes per row
    constexpr int SPR = KC / 2;        // 16-byte staging pieces per row
    static_assert(V2 == 1 || (V2 == 2 && KC == 16), "supported: one double2 per lane, or two with 16-column chunks");
    int64_t t = blockIdx.x;
    if (tiles_per_xcd > 0) t = (int64_t)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (t >= ntiles) return;
    const int tr0 = t_rowptr[t];
    const int nrows = t_rowptr[t + 1] - tr0;
    const int64_t z0 = t_nzptr[t];
    const int f0 = fp_ptr[t];
    const int nf = fp_ptr[t + 1] - f0;
    const int tid = threadIdx.x;
    const int part = tid % LPR;
    const int slot = tid / LPR;
    const bool has_row = slot < nrows;
    const int rslot = has_row ? slot : 0; // idle slots shadow row 0 of the tile (never stored)

    // this slot's row: values and LDS offsets of its X rows, padded to NNZ entries with zero coefficients that alias
    // the row's own first entry (a non-finite value in an unrelated X row can never leak in)
    double a[NNZ];
    unsigned xo2[(NNZ + 1) / 2]; // two 16-bit LDS offsets (in doubles) per register: keeps the kernel at <= 128 VGPRs
    const int p0 = t_rp[tr0 + t + rslot];
    const int cnt = t_rp[tr0 + t + rslot + 1] - p0;
    {
        const int64_t base = z0 + p0;
        const int last = cnt > 0 ? cnt - 1 : 0;
#pragma unroll
        for (int u = 0; u < NNZ; ++u) {
            const int uu = u < last ? u : last;
            const double av = t_val[base + uu];
            const unsigned lp = (unsigned)t_lcol[base + uu];
            const unsigned off = lp * KC + 2 * part + (V2 == 2 ? ((lp >> 1) & 1u) * 8u : 0u); // vector 0; vector 1 is off ^ 8
            a[u] = (u < cnt) ? av : 0.0;
            if (u & 1)
                xo2[u / 2] |= off << 16;
            else
                xo2[u / 2] = off;
        }
    }
    const int64_t yrow = (int64_t)t_rows[tr0 + rslot];

    // staging slots: slot indices past the footprint duplicate footprint row 0 (same bytes to the same LDS address).
    // Sources are kept as 32-bit element offsets (top bit: ghost buffer) and LDS targets as packed 16-bit offsets to
    // stay within the register budget of 3 waves per SIMD.
    unsigned soff[NL];
    unsigned dst2[(NL + 1) / 2];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        int idx = tid + 256 * i;
        int f = idx / SPR;
        const int q = idx % SPR;
        f = f < nf ? f : 0;
        const int32_t c = fp[f0 + f];
        const unsigned ghost = (c < m) ? 0u : 0x80000000u;
        const unsigned eo = ghost ? (unsigned)((int64_t)(c - m) * ldg) : (unsigned)((int64_t)c * ldx);
        soff[i] = (eo + 2 * q) | ghost;
        const unsigned lp = (unsigned)fpos[f0 + f];
        const unsigned d = lp * KC + (V2 == 2 ? (((unsigned)(q / 4) ^ ((lp >> 1) & 1u)) * 8u + 2u * (q % 4)) : 2u * q);
        if (i & 1)
            dst2[i / 2] |= d << 16;
        else
            dst2[i / 2] = d;
    }
#define RAILS_SRC(i) (((soff[i] & 0x80000000u) ? Xg : X) + (soff[i] & 0x7fffffffu))
#define RAILS_DST(i) (((i)&1) ? (dst2[(i) / 2] >> 16) : (dst2[(i) / 2] & 0xffffu))
    // Two column chunks are in flight in registers per thread (Little's law: with one chunk in flight the kernel is bound
    // by bytes-in-flight x latency, ~41 KB per CU); chunk ci goes to LDS buffer (ci & 1) right before use and its
    // register set is refilled with chunk ci + 2.  Loads past the last chunk are clamped to it (unused).
    double2_t stage0[NL], stage1[NL], stage2[NL];
    const int nchunks = (nc + KC - 1) / KC;
    const int lastc = (nchunks - 1) * KC;
#pragma unroll
    for (int i = 0; i < NL; ++i) stage0[i] = *reinterpret_cast<const double2_t *>(RAILS_SRC(i));
    if (NS >= 2) {
        const int c1 = KC < lastc ? KC : lastc;
#pragma unroll
        for (int i = 0; i < NL; ++i) stage1[i] = *reinterpret_cast<const double2_t *>(RAILS_SRC(i) + c1);
    }
    if (NS >= 3) {
        const int c2 = 2 * KC < lastc ? 2 * KC : lastc;
#pragma unroll
        for (int i = 0; i < NL; ++i) stage2[i] = *reinterpret_cast<const double2_t *>(RAILS_SRC(i) + c2);
    }
#define RAILS_TILE_STEP(STAGE, CI)                                                                      \
    do {                                                                                                \
        const int ci__ = (CI);                                                                          \
        double *Xs = smem + (size_t)(ci__ & 1) * xs_doubles;                                            \
        _Pragma("unroll") for (int i = 0; i < NL; ++i) *reinterpret_cast<double2_t *>(&Xs[RAILS_DST(i)]) = STAGE[i]; \
        const int cn__ = (ci__ + NS) * KC < lastc ? (ci__ + NS) * KC : lastc;                           \
        _Pragma("unroll") for (int i = 0; i < NL; ++i) STAGE[i] = *reinterpret_cast<const double2_t *>(RAILS_SRC(i) + cn__); \
        __syncthreads();                                                                                \
        double2_t acc[V2];                                                                              \
        _Pragma("unroll") for (int v = 0; v < V2; ++v) acc[v] = (double2_t){0.0, 0.0};                  \
        _Pragma("unroll") for (int u = 0; u < NNZ; ++u)                                                 \
        {                                                                                               \
            const unsigned off = (u & 1) ? (xo2[u / 2] >> 16) : (xo2[u / 2] & 0xffffu);                 \
            _Pragma("unroll") for (int v = 0; v < V2; ++v)                                              \
            {                                                                                           \
                const double2_t x = *reinterpret_cast<const double2_t *>(&Xs[v ? (off ^ 8u) : off]);    \
                acc[v].x = __builtin_fma(a[u], x.x, acc[v].x);                                          \
                acc[v].y = __builtin_fma(a[u], x.y, acc[v].y);                                          \
            }                                                                                           \
        }                                                                                               \
        _Pragma("unroll") for (int v = 0; v < V2; ++v)                                                  \
        {                                                                                               \
            const int cidx = ci__ * KC + v * 2 * LPR + 2 * part;                                        \
            if (cnt == 0) acc[v] = (double2_t){0.0, 0.0};                                               \
            if (has_row) {                                                                              \
                double *dst = Y + yrow * ldy + cidx;                                                    \
                if (cidx + 1 < nc)                                                                      \
                    *reinterpret_cast<double2_t *>(dst) = acc[v];                                       \
                else if (cidx < nc)                                                                     \
                    *dst = acc[v].x;                                                                    \
            }                                                                                           \
        }                                                                                               \
    } while (0)
    if (NS == 3) {
        for (int ci = 0; ci < nchunks; ci += 3) {
            RAILS_TILE_STEP(stage0, ci);
            if (ci + 1 < nchunks) RAILS_TILE_STEP(stage1, ci + 1);
            if (ci + 2 < nchunks) RAILS_TILE_STEP(stage2, ci + 2);
        }
    } else if (NS == 2) {
        for (int ci = 0; ci < nchunks; ci += 2) {
            RAILS_TILE_STEP(stage0, ci);
            if (ci + 1 < nchunks) RAILS_TILE_STEP(stage1, ci + 1);
        }
    } else {
        for (int ci = 0; ci < nchunks; ++ci) RAILS_TILE_STEP(stage0, ci);
    }
#undef RAILS_TILE_STEP
#undef RAILS_SRC
#undef RAILS_DST
}

// Structured-grid detection from the column offsets of local columns: returns true and (nx, ny, nz) when every
// sampled offset decomposes as dx + nx*dy + nx*ny*dz with |dx|,|dy|,|dz| <= 1.
bool detect_grid(const rails_csr *A, int64_t *nx, int64_t *ny, int64_t *nz)
{
    const int64_t m = A->m;
    if (m < 64) return false;
    std::vector<int64_t> offs;
    int64_t step = std::max<int64_t>(1, m / 4096);
    for (int64_t r = 0; r < m; r += step)
        for (int64_t p = A->h_rowptr[r]; p < A->h_rowptr[r + 1]; ++p) {
            int64_t c = A->h_col[p];
            if (c < m && c > r) offs.push_back(c - r);
        }
    std::sort(offs.begin(), offs.end());
    offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
    if (offs.empty() || offs.size() > 13 || offs[0] != 1) return false;
    auto has = [&](int64_t v) { return std::binary_search(offs.begin(), offs.end(), v); };
    if (offs.size() < 2) return false;
    int64_t a = offs[1]; // smallest offset > 1: nx (5/7-point) or nx-1 (9/27-point)
    int64_t gx = (has(a + 1) && has(a + 2)) ? a + 1 : a;
    if (gx < 3 || m % gx != 0) return false;
    // offsets beyond the in-plane cluster {1, nx-1, nx, nx+1} form a symmetric cluster around nx*ny
    int64_t gxy = 0, lo = 0, hi = 0;
    for (int64_t v : offs)
        if (v > gx + 1) {
            if (!lo) lo = v;
            hi = v;
        }
    if (lo) gxy = (lo + hi) / 2;
    if (gxy && (gxy % gx != 0 || !has(gxy))) return false;
    if (gxy == 0) gxy = m; // 2D grid
    if (m % gxy != 0) return false;
    // validate on the sample
    for (int64_t r = 0; r < m; r += step)
        for (int64_t p = A->h_rowptr[r]; p < A->h_rowptr[r + 1]; ++p) {
            int64_t c = A->h_col[p];
            if (c >= m) continue;
            int64_t x = r % gx, y = (r % gxy) / gx, z = r / gxy;
            int64_t cx = c % gx, cy = (c % gxy) / gx, cz = c / gxy;
            if (std::llabs(cx - x) > 1 || std::llabs(cy - y) > 1 || std::llabs(cz - z) > 1) return false;
        }
    *nx = gx;
    *ny = gxy / gx;
    *nz = m / gxy;
    return true;
}

struct GridInfo {
    bool valid = false;
    int64_t gx = 0, gy = 0, gz = 0;
    int bx = 0, by = 0, bz = 0;
};

struct TilePlan {
    std::vector<int32_t> t_rowptr, t_rows, t_rp, fp_ptr, fp;
    std::vector<uint16_t> fp_pos; // LDS row of every footprint entry
    int max_pos = 0;
    std::vector<int64_t> t_nzptr;
    std::vector<double> t_val;
    std::vector<uint16_t> t_lcol;
    int max_fp = 0, max_nz = 0, max_rows = 0;
    double reuse = 0.0;
};

// tile_of_row -> plan; returns false when a tile exceeds the caps
bool make_plan(const rails_csr *A, const std::vector<int32_t> &tile_of_row, int64_t ntiles, int fp_cap, int nz_cap, const GridInfo &G, TilePlan &P)
{
    const int64_t m = A->m;
    P.t_rowptr.assign(ntiles + 1, 0);
    for (int64_t r = 0; r < m; ++r) P.t_rowptr[tile_of_row[r] + 1]++;
    for (int64_t t = 0; t < ntiles; ++t) P.t_rowptr[t + 1] += P.t_rowptr[t];
    P.t_rows.resize(m);
    {
        std::vector<int32_t> next(P.t_rowptr.begin(), P.t_rowptr.end() - 1);
        for (int64_t r = 0; r < m; ++r) P.t_rows[next[tile_of_row[r]]++] = (int32_t)r;
    }
    P.t_nzptr.assign(ntiles + 1, 0);
    P.fp_ptr.assign(ntiles + 1, 0);
    P.t_rp.resize((size_t)m + ntiles);
    P.t_val.resize((size_t)A->nnz);
    P.t_lcol.resize((size_t)A->nnz);
    P.fp.reserve((size_t)A->nnz / 4 + 16);
    std::vector<int32_t> tmp;
    int64_t z = 0;
    for (int64_t t = 0; t < ntiles; ++t) {
        int r0 = P.t_rowptr[t], r1 = P.t_rowptr[t + 1];
        if (r1 - r0 > 256) return false;
        tmp.clear();
        for (int i = r0; i < r1; ++i) {
            int64_t r = P.t_rows[i];
            tmp.insert(tmp.end(), A->h_col.begin() + A->h_rowptr[r], A->h_col.begin() + A->h_rowptr[r + 1]);
        }
        int nzt = (int)tmp.size();
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        if ((int)tmp.size() > fp_cap || nzt > nz_cap) return false;
        // LDS row of every footprint entry.  Box tiles: position inside the halo box with the x extent padded to a
        // multiple of 4 rows, so the four row slots a ds_read_b128 lane group serves (x-consecutive rows) hit four
        // different bank quarters; ghost columns and non-grid tiles: consecutive positions.
        std::vector<uint16_t> pos(tmp.size());
        int npos = (int)tmp.size();
        if (G.valid && r1 > r0) {
            const int64_t rr = P.t_rows[r0];
            const int64_t ox = (rr % G.gx) / G.bx * G.bx, oy = ((rr / G.gx) % G.gy) / G.by * G.by, oz = (rr / (G.gx * G.gy)) / G.bz * G.bz;
            const int W = (G.bx + 2 + 3) / 4 * 4, H = G.by + 2;
            const int box = W * H * (G.bz + 2);
            int extra = 0;
            bool ok = true;
            for (size_t f = 0; f < tmp.size(); ++f) {
                int64_t c = tmp[f];
                if (c < m) {
                    int64_t fx = c % G.gx - ox + 1, fy = (c / G.gx) % G.gy - oy + 1, fz = c / (G.gx * G.gy) - oz + 1;
                    if (fx < 0 || fx >= W || fy < 0 || fy >= H || fz < 0 || fz >= G.bz + 2) {
                        ok = false;
                        break;
                    }
                    pos[f] = (uint16_t)(fx + W * (fy + H * fz));
                } else
                    pos[f] = (uint16_t)(box + extra++);
            }
            if (ok)
                npos = box + extra;
            else
                for (size_t f = 0; f < tmp.size(); ++f) pos[f] = (uint16_t)f;
        } else
            for (size_t f = 0; f < tmp.size(); ++f) pos[f] = (uint16_t)f;
        if (npos > 65535) return false;
        int loc = 0;
        for (int i = r0; i < r1; ++i) {
            int64_t r = P.t_rows[i];
            P.t_rp[(size_t)r0 + t + (i - r0)] = loc;
            for (int64_t p = A->h_rowptr[r]; p < A->h_rowptr[r + 1]; ++p) {
                P.t_val[z + loc] = A->h_val[p];
                P.t_lcol[z + loc] = pos[std::lower_bound(tmp.begin(), tmp.end(), A->h_col[p]) - tmp.begin()];
                loc++;
            }
        }
        P.t_rp[(size_t)r0 + t + (r1 - r0)] = loc;
        z += nzt;
        P.t_nzptr[t + 1] = z;
        P.fp.insert(P.fp.end(), tmp.begin(), tmp.end());
        P.fp_pos.insert(P.fp_pos.end(), pos.begin(), pos.end());
        P.max_pos = std::max(P.max_pos, npos);
        P.fp_ptr[t + 1] = (int32_t)P.fp.size();
        P.max_fp = std::max(P.max_fp, (int)tmp.size());
        P.max_nz = std::max(P.max_nz, nzt);
        P.max_rows = std::max(P.max_rows, r1 - r0);
    }
    P.reuse = P.fp.empty() ? 0.0 : (double)A->nnz / (double)P.fp.size();
    return true;
}

template <class T>
int upload(T **dst, const std::vector<T> &src)
{
    size_t n = src.empty() ? 1 : src.size();
    RAILS_HIP_CHECK(hipMalloc((void **)dst, n * sizeof(T)));
    if (!src.empty()) RAILS_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RAILS_OK;
}

} // namespace

bool rails_detect_grid(const rails_csr *A, int64_t *nx, int64_t *ny, int64_t *nz) { return detect_grid(A, nx, ny, nz); }

static bool rails_csr_is_grid(rails_csr *A)
{
    if (A->is_grid < 0) {
        int64_t gx = 0, gy = 0, gz = 0;
        A->is_grid = (spmm_env("RAILS_SPMM_TILE_BOX", 1) && detect_grid(A, &gx, &gy, &gz)) ? 1 : 0;
    }
    return A->is_grid == 1;
}

int rails_spmm_tiled(rails_ctx *c, rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, bool vec2,
                     int x_room, bool *done)
{
    *done = false;
    if (!vec2 || nc < 8 || A->nnz == 0 || A->m >= 0x7fffffffLL) return RAILS_OK;
    // the tile plan costs a host analysis of the whole matrix (~0.45 s per million rows): built on first use by a WIDE product
    // (warm start, the A*V benchmark) or when the kernel is asked for; the narrow in-loop products keep the row-gather kernel
    if (!A->tiled_ready && nc < 64 && A->variant == 0) return RAILS_OK;
    static const int env_rows = spmm_env("RAILS_SPMM_TILE_ROWS", 64);
    static const int env_kc = spmm_env("RAILS_SPMM_TILE_KC", 8);
    static const int env_box = spmm_env("RAILS_SPMM_TILE_BOX", 1);
    const int KC = (env_kc == 16) ? 16 : 8;
    const int lds_budget = 150 * 1024;
    if (!A->tiled_ready) {
        A->tiled_ready = true;
        A->tiled_ok = false;
        const int64_t m = A->m;
        std::vector<int32_t> tile_of_row(m);
        int64_t ntiles = 0;
        int64_t gx = 0, gy = 0, gz = 0;
        bool grid = env_box && detect_grid(A, &gx, &gy, &gz);
        GridInfo G;
        if (grid) {
            // box of about env_rows grid points: x longest (contiguous in memory), then y, then z
            int bx = 8, by = 4, bz = 4;
            if (env_rows <= 64) { bx = 4; by = 4; bz = 4; }
            if (env_rows >= 256) { bx = 8; by = 8; bz = 4; }
            if (gz == 1) { bz = 1; by = std::max(1, env_rows / bx); }
            int64_t tx = (gx + bx - 1) / bx, ty = (gy + by - 1) / by, tz = (gz + bz - 1) / bz;
            ntiles = tx * ty * tz;
            G.valid = true;
            G.gx = gx;
            G.gy = gy;
            G.gz = gz;
            G.bx = bx;
            G.by = by;
            G.bz = bz;
            // tiles are numbered along a Morton (Z-order) curve over their (x, y, z) box coordinates: tiles that share
            // halo rows are processed close together in time (and, with the XCD-aware block map, on the same XCD), so
            // the halo re-reads are served by L2 / Infinity Cache instead of HBM
            static const int env_morton = spmm_env("RAILS_SPMM_TILE_MORTON", 1);
            std::vector<int32_t> rank(ntiles);
            {
                std::vector<std::pair<uint64_t, int32_t>> keys(ntiles);
                auto spread = [](uint64_t v) { // 21 bits -> every third bit
                    v &= 0x1fffff;
                    v = (v | v << 32) & 0x1f00000000ffffull;
                    v = (v | v << 16) & 0x1f0000ff0000ffull;
                    v = (v | v << 8) & 0x100f00f00f00f00full;
                    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
                    v = (v | v << 2) & 0x1249249249249249ull;
                    return v;
                };
                for (int64_t z = 0; z < tz; ++z)
                    for (int64_t y = 0; y < ty; ++y)
                        for (int64_t x = 0; x < tx; ++x) {
                            int64_t id = z * ty * tx + y * tx + x;
                            uint64_t key = env_morton ? (spread(x) | spread(y) << 1 | spread(z) << 2) : (uint64_t)id;
                            keys[id] = std::make_pair(key, (int32_t)id);
                        }
                std::sort(keys.begin(), keys.end());
                for (int64_t i = 0; i < ntiles; ++i) rank[keys[i].second] = (int32_t)i;
            }
            for (int64_t r = 0; r < m; ++r) {
                int64_t x = r % gx, y = (r / gx) % gy, z = r / (gx * gy);
                tile_of_row[r] = rank[(z / bz) * ty * tx + (y / by) * tx + (x / bx)];
            }
        } else {
            int rows = std::min(env_rows, 256);
            ntiles = (m + rows - 1) / rows;
            // cheap pre-check on a sample of tiles before the full analysis: is a staged row used ~twice or more?
            {
                std::vector<int32_t> tmp;
                double snz = 0, sfp = 0;
                int64_t step = std::max<int64_t>(1, ntiles / 64);
                for (int64_t tt = 0; tt < ntiles; tt += step) {
                    int64_t r0 = tt * rows, r1 = std::min<int64_t>(m, r0 + rows);
                    tmp.assign(A->h_col.begin() + A->h_rowptr[r0], A->h_col.begin() + A->h_rowptr[r1]);
                    snz += (double)tmp.size();
                    std::sort(tmp.begin(), tmp.end());
                    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
                    sfp += (double)tmp.size();
                }
                if (sfp <= 0 || snz / sfp < 1.8) return RAILS_OK;
            }
            for (int64_t r = 0; r < m; ++r) tile_of_row[r] = (int32_t)(r / rows);
        }
        // caps from the LDS budget: vals 8 B + lcol 2 B per nonzero, KC*8 B per footprint row, 1 KiB of row offsets
        TilePlan P;
        int nz_cap = 256 * std::max(1, A->max_row_nnz);
        int fp_cap = 65535;
        if (make_plan(A, tile_of_row, ntiles, fp_cap, nz_cap, G, P)) {
            size_t need = (size_t)((P.max_nz + 3) / 4 * 4) * 8 + (size_t)P.max_pos * KC * 8 + 264 * 4 + (size_t)((P.max_nz + 3) / 4 * 4) * 2 + 64;
            if (P.reuse >= 1.8 && need <= (size_t)lds_budget) {
                RAILS_TRY(upload(&A->t_rowptr, P.t_rowptr));
                RAILS_TRY(upload(&A->t_rows, P.t_rows));
                RAILS_TRY(upload(&A->t_nzptr, P.t_nzptr));
                RAILS_TRY(upload(&A->t_rp, P.t_rp));
                RAILS_TRY(upload(&A->t_val, P.t_val));
                RAILS_TRY(upload(&A->t_lcol, P.t_lcol));
                RAILS_TRY(upload(&A->t_fp_ptr, P.fp_ptr));
                RAILS_TRY(upload(&A->t_fp, P.fp));
                RAILS_TRY(upload(&A->t_fpos, P.fp_pos));
                A->max_pos = P.max_pos;
                A->n_tiles = ntiles;
                A->max_fp = P.max_fp;
                A->max_nz = (P.max_nz + 3) / 4 * 4;
                A->tile_rows = P.max_rows;
                A->tile_reuse = P.reuse;
                A->tile_grid = grid;
                A->tiled_ok = true;
            }
        }
    }
    if (!A->tiled_ok) return RAILS_OK;
    const int xs_doubles = A->max_pos * KC;
    size_t lds = (size_t)A->max_nz * 8 + (size_t)xs_doubles * 8 + 264 * 4 + (size_t)A->max_nz * 2 + 64;
    int64_t grid = A->n_tiles, tpx = 0;
    static const int xcd_aware = spmm_env("RAILS_SPMM_XCD", 1);
    if (xcd_aware && grid >= 64) {
        tpx = (grid + 7) / 8;
        grid = tpx * 8;
    }
    static const int env_reg = spmm_env("RAILS_SPMM_TILE_REG", 1);
    // the register-resident kernel reads whole KC-column chunks unconditionally: the padded row must have room for the
    // rounded-up last chunk, and ghost rows (stored with ld = nc) must be a whole number of chunks
    const bool full_width_ok = ((nc + KC - 1) / KC * KC <= x_room) && (A->n_ghost == 0 || nc % KC == 0);
    {
        // wide chunks (16 columns, 4 lanes x 32 B per row) for wide panels; narrow panels keep two 8-column chunks in flight
        // (measured on MI355X: not faster than 8-column chunks -- 0.82 vs 0.79 ms on the 27-point stencil at nc = 128; both forms
        // move ~10 B/clk/CU through the load path, which is what bounds this kernel: profiles/r01_spmm_tiled_wide.md -- so it
        // is off unless asked for: operator variant 6 or RAILS_SPMM_TILE_WIDE=1)
        static const int env_wide = spmm_env("RAILS_SPMM_TILE_WIDE", 0);
        const bool wide = (env_wide || A->variant == 6) && nc >= 32 && ((nc + 15) / 16 * 16 <= x_room) && (A->n_ghost == 0 || nc % 16 == 0);
        const int KCr = wide ? 16 : KC, V2r = wide ? 2 : 1;
        const int lpr_r = KCr / (2 * V2r);
        const int need_nl_r = (A->max_fp * (KCr / 2) + 255) / 256;
        const int xs_r = A->max_pos * KCr;
        const size_t lds_reg = 2 * (size_t)xs_r * 8;
        if (env_reg && (wide || full_width_ok) && xs_r < 65536 && (int64_t)A->m * ldx < 0x7fffffffLL && (int64_t)(A->n_ghost + 1) * ldg < 0x7fffffffLL && A->tile_rows <= 256 / lpr_r && A->max_row_nnz <= 32 && need_nl_r <= 8 && lds_reg <= (size_t)lds_budget) {
            const int nnz4 = (A->max_row_nnz + 3) / 4;
            // chunks in flight in registers for long rows: 2 = deeper pipeline at 2 waves/SIMD, 1 = 16 fewer VGPRs, 3 waves/SIMD
            static const int env_ns = spmm_env("RAILS_SPMM_TILE_NS", 2);
#define RAILS_REG_ARGS A->m, A->n_tiles, A->t_rowptr, A->t_rows, A->t_nzptr, A->t_rp, A->t_val, A->t_lcol, A->t_fp_ptr, A->t_fp, A->t_fpos, X, ldx, Xg, ldg, Y, ldy, nc, tpx, xs_r
// two chunks in flight unless that needs more than 256 VGPRs (wide chunks with long rows and 8 staging slots)
#define RAILS_LAUNCH_REG_NS(KCV, NNZV, NLV, V2V, NSV)                                                                                 \
    do {                                                                                                                               \
        RAILS_HIP_CHECK(hipFuncSetAttribute((const void *)k_spmm_tiled_reg<KCV, NNZV, NLV, V2V, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg)); \
        RAILS_LAUNCH((k_spmm_tiled_reg<KCV, NNZV, NLV, V2V, NSV>), dim3((unsigned)grid), dim3(256), lds_reg, c->stream, RAILS_REG_ARGS); \
    } while (0)
#define RAILS_LAUNCH_REG(KCV, NNZV, NLV, V2V)                                                                                         \
    do {                                                                                                                               \
        constexpr bool tight = (V2V == 2 && NNZV * 2 + NLV * 8 > 100);                                                                 \
        if (tight || (env_ns == 1 && NNZV > 16))                                                                                       \
            RAILS_LAUNCH_REG_NS(KCV, NNZV, NLV, V2V, 1);                                                                               \
        else if (env_ns == 3 && V2V == 1 && NLV == 4)                                                                                  \
            RAILS_LAUNCH_REG_NS(KCV, NNZV, NLV, V2V, 3);                                                                               \
        else                                                                                                                           \
            RAILS_LAUNCH_REG_NS(KCV, NNZV, NLV, V2V, 2);                                                                               \
    } while (0)
#define RAILS_REG_NL(KCV, NNZV, V2V)                             \
    do {                                                         \
        if (need_nl_r <= 4) RAILS_LAUNCH_REG(KCV, NNZV, 4, V2V); \
        else RAILS_LAUNCH_REG(KCV, NNZV, 8, V2V);                \
    } while (0)
#define RAILS_REG_NNZ(KCV, V2V)                                  \
    do {                                                         \
        if (nnz4 <= 2) RAILS_REG_NL(KCV, 8, V2V);                \
        else if (nnz4 <= 4) RAILS_REG_NL(KCV, 16, V2V);          \
        else if (nnz4 <= 7) RAILS_REG_NL(KCV, 28, V2V);          \
        else RAILS_REG_NL(KCV, 32, V2V);                         \
    } while (0)
            if (wide)
                RAILS_REG_NNZ(16, 2);
            else if (KC == 8)
                RAILS_REG_NNZ(8, 1);
            else
                RAILS_REG_NNZ(16, 1);
#undef RAILS_REG_NNZ
#undef RAILS_REG_NL
#undef RAILS_LAUNCH_REG
#undef RAILS_LAUNCH_REG_NS
#undef RAILS_REG_ARGS
            A->last_kernel = "k_spmm_tiled_reg";
            *done = true;
            return RAILS_OK;
        }
    }
    static const int env_pipe = spmm_env("RAILS_SPMM_TILE_PIPE", 1);
    const int lpr = KC / 2;
    const int need_nl = (A->max_fp * lpr + 255) / 256;
    size_t lds_pipe = lds + (size_t)xs_doubles * 8;
    bool pipe = env_pipe && need_nl <= 8 && lds_pipe <= (size_t)lds_budget;
#define RAILS_TILED_ARGS A->m, A->n_tiles, A->t_rowptr, A->t_rows, A->t_nzptr, A->t_rp, A->t_val, A->t_lcol, A->t_fp_ptr, A->t_fp, A->t_fpos, X, ldx, Xg, ldg, Y, ldy, nc, tpx, A->max_nz, xs_doubles
#define RAILS_LAUNCH_PIPE(KCV, NLV)                                                                                                    \
    do {                                                                                                                               \
        RAILS_HIP_CHECK(hipFuncSetAttribute((const void *)k_spmm_tiled_pipe<KCV, NLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pipe)); \
        RAILS_LAUNCH((k_spmm_tiled_pipe<KCV, NLV>), dim3((unsigned)grid), dim3(256), lds_pipe, c->stream, RAILS_TILED_ARGS);  \
    } while (0)
    if (pipe) {
        if (KC == 8) {
            if (need_nl <= 4) RAILS_LAUNCH_PIPE(8, 4);
            else RAILS_LAUNCH_PIPE(8, 8);
        } else {
            if (need_nl <= 4) RAILS_LAUNCH_PIPE(16, 4);
            else RAILS_LAUNCH_PIPE(16, 8);
        }
    } else if (KC == 8) {
        RAILS_HIP_CHECK(hipFuncSetAttribute((const void *)k_spmm_tiled<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RAILS_LAUNCH((k_spmm_tiled<8>), dim3((unsigned)grid), dim3(256), lds, c->stream, RAILS_TILED_ARGS);
    } else {
        RAILS_HIP_CHECK(hipFuncSetAttribute((const void *)k_spmm_tiled<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RAILS_LAUNCH((k_spmm_tiled<16>), dim3((unsigned)grid), dim3(256), lds, c->stream, RAILS_TILED_ARGS);
    }
#undef RAILS_LAUNCH_PIPE
#undef RAILS_TILED_ARGS
    A->last_kernel = pipe ? "k_spmm_tiled_pipe" : "k_spmm_tiled";
    *done = true;
    return RAILS_OK;
}
