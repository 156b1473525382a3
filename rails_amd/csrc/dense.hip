// dense.hip -- tall-skinny dense kernels on v_mfma_f64_16x16x4_f64 (gfx950):
//   rails_gram        C = X^T Y          (MultiVector::dot,  src/StlWrapper.cpp:394-412)
//   rails_panel_gemm  Y = beta Y + alpha X C   (MultiVector * DenseMatrix, src/StlWrapper.cpp:168-187)
//
// fp64 MFMA on gfx950 runs at the fp64 vector rate; it is used here because it needs ONE operand
// register per lane per 2048 flop (no LDS broadcast of the small operand, no register tiling),
// which leaves the load path free: both kernels are HBM-bound whenever one small dimension is
// <= 32 (every call of the solver) and reach the fp64 ridge only for k x 128 restart products.
//
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md section 3): lane l supplies A[i=l&15][k=l>>4]
// and B[k=l>>4][j=l&15]; it receives D[row=(l>>4)+4*v][col=l&15] in result register v (0..3).
//
// Reductions over rows are two-stage and deterministic: per-block partial tiles are written to a
// workspace and summed in a fixed order by a second kernel (no float atomics), then all-reduced
// over the ranks through the context hook.
#include "rails_internal.h"

#include <dlfcn.h>
#include <atomic>
#include <mutex>

#include <algorithm>

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v4f64 mfma_f64(double a, double b, v4f64 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------- Gram ---
// grid.x = row slabs, grid.y = tile groups (gi over X column tiles, gj over Y column tiles).
// Each of the 4 waves accumulates TI x TJ output tiles over its share of the slab's rows.
template <int TI, int TJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_gram(const double *__restrict__ X, int ldx, int a, const double *__restrict__ Y, int ldy,
                                              int b, int64_t m, int64_t rows_per_slab, int ngj, double *__restrict__ partial)
{
    __shared__ double red[TI * TJ * 256];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int gi = blockIdx.y / ngj, gj = blockIdx.y % ngj;
    const int xcol0 = gi * TI * 16, ycol0 = gj * TJ * 16;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_slab;
    int64_t r_end = r_begin + rows_per_slab;
    if (r_end > m) r_end = m;

    v4f64 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

    // Loads without branches (a load inside a conditional gets its own exec-masked block, and the compiler then waits for ALL loads in
    // flight -- the next step's too -- before the first MFMA: see k_gram_cols): columns outside the operands are clamped to the tile's
    // first column (their results are never written), rows past the slab to its last row with the Y operand zeroed.
    int xoff[TI], yoff[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) xoff[i] = (xcol0 + 16 * i + li) < a ? xcol0 + 16 * i + li : 0;
#pragma unroll
    for (int j = 0; j < TJ; ++j) yoff[j] = (ycol0 + 16 * j + li) < b ? ycol0 + 16 * j + li : 0;

    // software pipeline: the operands of the next 4-row step are in flight while the MFMAs of this one run (one step's loads per
    // wave do not cover the HBM latency at 3 waves per SIMD: the Gram at 17 columns ran at 37 % of the HBM rate without it)
    auto fetch = [&](int64_t r, double *xa, double *yb) {
        const int64_t row = r + kk;
        const bool rok = row < r_end;
        const int64_t rc = rok ? row : (r_end > 0 ? r_end - 1 : 0);
        const double *xr = X + rc * ldx;
        const double *yr = Y + rc * ldy;
#pragma unroll
        for (int i = 0; i < TI; ++i) xa[i] = xr[xoff[i]];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const double t = yr[yoff[j]];
            yb[j] = rok ? t : 0.0;
        }
    };
    double xa[TI], yb[TJ], xn[TI], yn[TJ];
    int64_t r = r_begin + 4 * wave;
    if (r < r_end) fetch(r, xa, yb);
    for (; r < r_end; r += 16) {
        const bool more = r + 16 < r_end;
        if (more) fetch(r + 16, xn, yn);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = mfma_f64(xa[i], yb[j], acc[i][j]);
        if (more) {
#pragma unroll
            for (int i = 0; i < TI; ++i) xa[i] = xn[i];
#pragma unroll
            for (int j = 0; j < TJ; ++j) yb[j] = yn[j];
        }
    }

    // cross-wave reduction in a fixed order (wave 0 += wave 1, 2, 3)
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v) red[((i * TJ + j) * 4 + v) * 64 + lane] = acc[i][j][v];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[i][j][v] += red[((i * TJ + j) * 4 + v) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        double *P = partial + (int64_t)blockIdx.x * a * b;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    int ci = xcol0 + 16 * i + kk + 4 * v; // D row  -> X column
                    int cj = ycol0 + 16 * j + li;         // D col  -> Y column
                    if (ci < a && cj < b) P[ci + (int64_t)cj * a] = acc[i][j][v];
                }
    }
}

// Wide-X / narrow-Y form (the projections [P | X]' X of the coordinate-space back end: a = dim + w >> b = w <= 32): the four waves
// of a block take four ADJACENT 64-column strips of X for the SAME rows, so a block reads whole 2 KiB row segments (the row-split
// form above reads 512 B per row and block; measured 3.0 -> see profiles/r01_kernels.md) and Y's few columns once; every wave owns
// its output tiles, so there is no cross-wave reduction.  Two 4-row steps are in flight per wave.
// NE columns of Y beyond 16 * TJ (the prefetched random vector that rides at the end of an A*W block: b = 17) are done with plain
// multiply-adds on the values the lanes hold anyway: a second MFMA column tile for ONE column doubles the matrix work, and the pass
// took a third longer for it (0.81 against 0.61 ms at 369 x 17 / 369 x 16, 1M rows).  A lane sums its own rows (r + kk, r + 4 + kk, ...) of
// its own column; the four row classes are added at the end (kk = 0 + 1, + 2, + 3: a fixed order).
template <int TI, int TJ, int NE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_gram_cols(const double *__restrict__ X, int ldx, int a,
                                                                                             const double *__restrict__ Y, int ldy, int b, int64_t m,
                                                                                             int64_t rows_per_slab, double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int xcol0 = ((int)blockIdx.y * 4 + wave) * TI * 16;
    if (xcol0 >= a) return; // wave-uniform; no barrier below
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_slab;
    int64_t r_end = r_begin + rows_per_slab;
    if (r_end > m) r_end = m;

    v4f64 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double acce[TI][NE > 0 ? NE : 1];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < (NE > 0 ? NE : 1); ++q) acce[i][q] = 0.0;
    // Loads without branches and without selects on X: with `cond ? load : 0` every load sat in an exec-masked block of its own and the
    // compiler waited for everything in flight (`s_waitcnt vmcnt(0)`: the next step's operands too) before the first MFMA of a step.
    // Columns outside the operands are clamped to column 0 (their results are never written), rows past the slab to its last row
    // with the Y operand zeroed: what comes back from there is multiplied by zero.
    int xoff[TI], yoff[TJ], eoff[NE > 0 ? NE : 1];
#pragma unroll
    for (int i = 0; i < TI; ++i) xoff[i] = (xcol0 + 16 * i + li) < a ? xcol0 + 16 * i + li : 0;
#pragma unroll
    for (int j = 0; j < TJ; ++j) yoff[j] = (16 * j + li) < b ? 16 * j + li : 0;
#pragma unroll
    for (int q = 0; q < NE; ++q) eoff[q] = 16 * TJ + q < b ? 16 * TJ + q : 0;

    auto fetch = [&](int64_t r, double *xa, double *yb, double *ye) {
        const int64_t row = r + kk;
        const bool rok = row < r_end;
        const int64_t rc = rok ? row : (r_end > 0 ? r_end - 1 : 0);
        const double *xr = X + rc * ldx;
        const double *yr = Y + rc * ldy;
#pragma unroll
        for (int i = 0; i < TI; ++i) xa[i] = xr[xoff[i]];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const double t = yr[yoff[j]];
            yb[j] = rok ? t : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const double t = yr[eoff[q]]; // (one address per row: a broadcast; a column past b is clamped and its result dropped)
            ye[q] = rok ? t : 0.0;
        }
    };
    auto work = [&](const double *xa, const double *yb, const double *ye) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = mfma_f64(xa[i], yb[j], acc[i][j]);
#pragma unroll
            for (int q = 0; q < NE; ++q) acce[i][q] += xa[i] * ye[q];
        }
    };
    double xa[TI], ya[TJ], xb[TI], yb[TJ], ea[NE > 0 ? NE : 1], eb[NE > 0 ? NE : 1];
    fetch(r_begin, xa, ya, ea);
    fetch(r_begin + 4, xb, yb, eb);
    for (int64_t r = r_begin; r < r_end; r += 8) {
        work(xa, ya, ea);
        fetch(r + 8, xa, ya, ea); // rows past the slab come back as zeros
        work(xb, yb, eb);
        fetch(r + 12, xb, yb, eb);
    }
    double *P = partial + (int64_t)blockIdx.x * a * b;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                int ci = xcol0 + 16 * i + kk + 4 * v; // D row  -> X column
                int cj = 16 * j + li;                 // D col  -> Y column
                if (ci < a && cj < b) P[ci + (int64_t)cj * a] = acc[i][j][v];
            }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const double t0 = acce[i][q];
            const double t1 = __shfl(t0, li + 16, 64), t2 = __shfl(t0, li + 32, 64), t3 = __shfl(t0, li + 48, 64);
            const int ci = xcol0 + 16 * i + li;
            if (kk == 0 && ci < a && 16 * TJ + q < b) P[ci + (int64_t)(16 * TJ + q) * a] = ((t0 + t1) + t2) + t3;
        }
}

// out[e] = sum_t partial[t][e] in a fixed order: 16 interleaved strands per element, then the strands 0..15
__global__ __launch_bounds__(1024) void k_reduce_partials(const double *__restrict__ partial, int64_t nslab, int64_t n, double *__restrict__ out)
{
    __shared__ double sh[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + tx;
    double s = 0.0;
    if (e < n)
        for (int64_t t = ty; t < nslab; t += 16) s += partial[t * n + e];
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < n) {
        double r = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) r += sh[g][tx];
        out[e] = r;
    }
}

template <int TI, int TJ>
void launch_gram(rails_ctx *c, const double *X, int ldx, int a, const double *Y, int ldy, int b, int64_t m, int64_t rps,
                 int64_t nslab, double *partial)
{
    int ngi = (a + 16 * TI - 1) / (16 * TI), ngj = (b + 16 * TJ - 1) / (16 * TJ);
    RAILS_LAUNCH((k_gram<TI, TJ>), dim3((unsigned)nslab, (unsigned)(ngi * ngj)), dim3(256), 0, c->stream, X, ldx, a, Y, ldy, b,
                       m, rps, ngj, partial);
}

// ------------------------------------------------------------------------- panel GEMM ---
// Each wave owns 16 rows and all r (<= 16*TR) output columns; the block streams C through LDS in
// chunks of KC rows.  X is read with the k index permuted inside every 16-column block so that a
// lane reads 4 consecutive doubles: lane (i,kk) holds X[row i][kb + 4*kk + s], s = 0..3, and MFMA
// number s sums over k in {kb + 4*kk + s}; C's rows are fetched from LDS with the same permutation.
template <int TR, int KC>
__global__ __launch_bounds__(256) void k_panel_gemm(double alpha, const double *X, int ldx, int k,
                                                    const double *__restrict__ C, int r, double beta, double *Yp,
                                                    int ldy, int64_t m, int vec_ok)
{
    constexpr int RL = 16 * TR + 4; // LDS row length (doubles): +4 keeps the 4 k-groups on disjoint banks
    __shared__ double Cs[KC * RL];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
    const int64_t myrow = r0 + li;
    const bool rowok = myrow < m;

    v4f64 acc[TR];
#pragma unroll
    for (int t = 0; t < TR; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};

    const double *xrow = X + myrow * ldx;
    for (int kc = 0; kc < k; kc += KC) {
        __syncthreads();
        // stage C[kc:kc+KC, 0:r) (col-major, ld = k) into LDS row-major, zero padded
        for (int idx = threadIdx.x; idx < KC * 16 * TR; idx += 256) {
            int kl = idx % KC, j = idx / KC;
            double v = 0.0;
            if (kc + kl < k && j < r) v = C[(kc + kl) + (int64_t)j * k];
            Cs[kl * RL + j] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < KC; kb += 16) {
            const int kcol = kc + kb + 4 * kk;
            double xs[4];
            if (rowok && vec_ok && kcol + 4 <= k) {
                v2f64 t0 = *reinterpret_cast<const v2f64 *>(xrow + kcol);
                v2f64 t1 = *reinterpret_cast<const v2f64 *>(xrow + kcol + 2);
                xs[0] = t0.x;
                xs[1] = t0.y;
                xs[2] = t1.x;
                xs[3] = t1.y;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) xs[s] = (rowok && kcol + s < k) ? xrow[kcol + s] : 0.0;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double *crow = &Cs[(kb + 4 * kk + s) * RL + li];
#pragma unroll
                for (int t = 0; t < TR; ++t) acc[t] = mfma_f64(xs[s], crow[16 * t], acc[t]);
            }
        }
    }
    // D[row = kk + 4v][col = li]
#pragma unroll
    for (int t = 0; t < TR; ++t) {
        const int j = 16 * t + li;
        if (j >= r) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t row = r0 + kk + 4 * v;
            if (row >= m) continue;
            double *dst = Yp + row * ldy + j;
            double val = alpha * acc[t][v];
            if (beta != 0.0) val += beta * (*dst);
            *dst = val;
        }
    }
}

template <int TR, int KC>
void launch_pg(rails_ctx *c, double alpha, const double *X, int ldx, int k, const double *C, int r, double beta, double *Y,
               int ldy, int64_t m, int vec_ok)
{
    int64_t grid = (m + 63) / 64;
    RAILS_LAUNCH((k_panel_gemm<TR, KC>), dim3((unsigned)grid), dim3(256), 0, c->stream, alpha, X, ldx, k, C, r, beta, Y,
                       ldy, m, vec_ok);
}

// The restart rotation P2 = P Q (k and r in the hundreds: the one compute-bound product of the solver, src/StlWrapper.cpp:168-187 behind
// src/LyapunovSolver.hpp:265,290) in ONE pass: a wave keeps 16 rows x ALL r <= 288 output columns in registers (TR tiles of 16 x 16, 8
// VGPRs each), so X is read once (k_panel_gemm in 128-column slices read it once per slice and padded r = 268 to 384 columns of MFMA
// work) and the matrix pipe does exactly ceil(r / 16) tiles per step.  C comes packed (k_pack_ct: transposed, rows of 16 TR + 4 doubles
// -- the two k-groups a ds_read_b64 serves together fall on disjoint bank halves -- zero padded to whole chunks of 32 rows) and streams
// through a double buffer in LDS by LDS-DMA: chunk i + 1 lands while chunk i is multiplied, one barrier per chunk, no staging loop in
// the waves (a first form staged C with plain loads and LDS stores, 36 round trips per thread and chunk: 28 TFLOP/s).  8 waves x 16
// rows per workgroup, one workgroup per CU (2 x 73 KiB of LDS).  X and Y must not alias.
__global__ void k_pack_ct(const double *__restrict__ C, int k, int r, int kpad, int RL, double *__restrict__ out)
{
    const int64_t n = (int64_t)kpad * RL;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int kl = (int)(q / RL), j = (int)(q % RL);
        out[q] = (kl < k && j < r) ? C[kl + (int64_t)j * k] : 0.0;
    }
}

template <int TR>
__global__ __launch_bounds__(512) void k_panel_gemm_wide(double alpha, const double *__restrict__ X, int ldx, int k, const double *__restrict__ Ct /* packed, kpad x RL */,
                                                         int r, double beta, double *__restrict__ Yp, int ldy, int64_t m, int vec_ok)
{
    constexpr int KC = 32, RL = 16 * TR + 4, CHUNK_B = KC * RL * 8, PIECES = CHUNK_B / 1024; // (KC * RL * 8 is a multiple of 1024 for every TR)
    static_assert(CHUNK_B % 1024 == 0, "whole LDS-DMA pieces per chunk");
    extern __shared__ double Cs[]; // 2 x KC x RL
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, kk = lane >> 4;
    const int64_t r0 = ((int64_t)blockIdx.x * 8 + wave) * 16;
    const int64_t myrow = r0 + li;
    const bool rowok = myrow < m;
    const int nchunks = (k + KC - 1) / KC;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)Cs;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto stage = [&](int chunk) { // this wave's pieces of a chunk: piece p goes to wave p % 8
        const char *src = reinterpret_cast<const char *>(Ct) + (size_t)chunk * CHUNK_B;
        const uint32_t dst = lds_base + (uint32_t)((chunk & 1) * CHUNK_B);
        for (int p = wave; p < PIECES; p += 8) {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(src + (size_t)p * 1024), "s"(dst + (uint32_t)(p * 1024)) : "memory");
        }
    };
    v4f64 acc[TR];
#pragma unroll
    for (int t = 0; t < TR; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const double *xrow = X + (rowok ? myrow : m - 1) * ldx;
    auto fetch = [&](int kcol, double *xs) {
        if (vec_ok && kcol + 4 <= k) {
            v2f64 t0 = *reinterpret_cast<const v2f64 *>(xrow + kcol);
            v2f64 t1 = *reinterpret_cast<const v2f64 *>(xrow + kcol + 2);
            xs[0] = t0.x;
            xs[1] = t0.y;
            xs[2] = t1.x;
            xs[3] = t1.y;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) xs[s] = (kcol + s < k) ? xrow[kcol + s] : 0.0;
        }
    };
    double xa[4], xb[4];
    stage(0);
    fetch(4 * kk, xa);
    fetch(16 + 4 * kk, xb);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    __builtin_amdgcn_s_barrier();
    for (int ci = 0; ci < nchunks; ++ci) {
        if (ci + 1 < nchunks) stage(ci + 1); // into the buffer every wave left at the barrier above
        double xc[4], xd[4];
        fetch((ci + 1) * KC + 4 * kk, xc);
        fetch((ci + 1) * KC + 16 + 4 * kk, xd);
        const double *cb = Cs + (size_t)(ci & 1) * (KC * RL);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double *crow = &cb[(4 * kk + s) * RL + li];
#pragma unroll
            for (int t = 0; t < TR; ++t) acc[t] = mfma_f64(xa[s], crow[16 * t], acc[t]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double *crow = &cb[(16 + 4 * kk + s) * RL + li];
#pragma unroll
            for (int t = 0; t < TR; ++t) acc[t] = mfma_f64(xb[s], crow[16 * t], acc[t]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            xa[s] = xc[s];
            xb[s] = xd[s];
        }
        // the next chunk has landed (this wave's pieces; then everybody's) and nobody reads this chunk's buffer any more
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int t = 0; t < TR; ++t) {
        const int j = 16 * t + li;
        if (j >= r) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t row = r0 + kk + 4 * v;
            if (row >= m) continue;
            double *dst = Yp + row * ldy + j;
            double val = alpha * acc[t][v];
            if (beta != 0.0) val += beta * (*dst);
            *dst = val;
        }
    }
}

// C (k x r, column-major, on the device) -> packed in `ct`; then the product
template <int TR>
int launch_pg_wide(rails_ctx *c, double alpha, const double *X, int ldx, int k, const double *C, int r, double beta, double *Y, int ldy, int64_t m, int vec_ok, double *ct)
{
    constexpr int RL = 16 * TR + 4;
    const int kpad = (k + 31) / 32 * 32;
    RAILS_LAUNCH(k_pack_ct, dim3((unsigned)std::min<int64_t>(512, ((int64_t)kpad * RL + 255) / 256)), dim3(256), 0, c->stream, C, k, r, kpad, RL, ct);
    const size_t lds = (size_t)2 * 32 * RL * sizeof(double);
    RAILS_HIP_CHECK(hipFuncSetAttribute((const void *)k_panel_gemm_wide<TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RAILS_LAUNCH((k_panel_gemm_wide<TR>), dim3((unsigned)((m + 127) / 128)), dim3(512), lds, c->stream, alpha, X, ldx, k, ct, r, beta, Y, ldy, m, vec_ok);
    return RAILS_OK;
}

} // namespace

int rails_gram_dev(rails_ctx *c, const double *X, int ldx, const double *Y, int ldy, int64_t m, int a, int b, double *C_dev)
{
    if (a <= 0 || b <= 0) return RAILS_OK;
    // slab count: enough blocks to fill the chip, partial-tile traffic bounded to ~4% of the input
    double bound = 0.02 * (double)m * (double)(a + b) / ((double)a * (double)b);
    int64_t nslab = (int64_t)std::min<double>(1024.0, std::max<double>(1.0, bound));
    int64_t maxslab = (m + 15) / 16;
    if (nslab > maxslab) nslab = std::max<int64_t>(1, maxslab);
    int64_t rps = (m + nslab - 1) / nslab;
    rps = (rps + 15) / 16 * 16;
    if (rps < 16) rps = 16;
    nslab = std::max<int64_t>(1, (m + rps - 1) / rps);
    size_t n = (size_t)a * b;
    RAILS_TRY(rails_ws_reserve(c, (size_t)nslab * n * sizeof(double)));
    if (b <= 16 && a <= 16)
        launch_gram<1, 1>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
    else if (b <= 32 && a >= 128) {
        static const int cols_form = getenv("RAILS_GRAM_COLS") ? atoi(getenv("RAILS_GRAM_COLS")) : 1;
        if (cols_form) {
            // tiles of 16 X-columns per wave: the choice that leaves the fewest idle tile slots in blocks of four waves
            const int ntiles = (a + 15) / 16;
            // b = 17: one MFMA column tile and the 17th column on the vector unit (see k_gram_cols)
            static const int extra_env = getenv("RAILS_GRAM_EXTRA_COLUMN") ? atoi(getenv("RAILS_GRAM_EXTRA_COLUMN")) : 1;
            const bool extra = extra_env && b == 17;
            const int cand2[3] = {3, 4, 5}, cand1[3] = {4, 6, 8};
            const int *cand = (b <= 16 || extra) ? cand1 : cand2;
            int best = cand[1], best_cost = 1 << 30;
            for (int q = 0; q < 3; ++q) {
                int ti = cand[q], strips = (ntiles + ti - 1) / ti, cost = (strips + 3) / 4 * 4 * ti;
                if (cost < best_cost) best = ti, best_cost = cost;
            }
            const dim3 grid((unsigned)nslab, (unsigned)(((ntiles + best - 1) / best + 3) / 4));
#define RAILS_GRAM_COLS_CASE(TI, TJ, NE)                                                                                                 \
    RAILS_LAUNCH((k_gram_cols<TI, TJ, NE>), grid, dim3(256), 0, c->stream, X, ldx, a, Y, ldy, b, m, rps, c->ws)
            if (extra) {
                if (best == 4)
                    RAILS_GRAM_COLS_CASE(4, 1, 1);
                else if (best == 6)
                    RAILS_GRAM_COLS_CASE(6, 1, 1);
                else
                    RAILS_GRAM_COLS_CASE(8, 1, 1);
            } else if (b <= 16) {
                if (best == 4)
                    RAILS_GRAM_COLS_CASE(4, 1, 0);
                else if (best == 6)
                    RAILS_GRAM_COLS_CASE(6, 1, 0);
                else
                    RAILS_GRAM_COLS_CASE(8, 1, 0);
            } else {
                if (best == 3)
                    RAILS_GRAM_COLS_CASE(3, 2, 0);
                else if (best == 4)
                    RAILS_GRAM_COLS_CASE(4, 2, 0);
                else
                    RAILS_GRAM_COLS_CASE(5, 2, 0);
            }
#undef RAILS_GRAM_COLS_CASE
        } else if (b <= 16)
            launch_gram<8, 1>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
        else
            launch_gram<4, 2>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
    } else if (b <= 16)
        launch_gram<8, 1>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
    else if (a <= 16)
        launch_gram<1, 8>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
    else if (a <= 32 && b <= 32) // the block's own 17 x 17 Gram matrices of CholQR2
        launch_gram<2, 2>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
    else if (b <= 32)
        launch_gram<4, 2>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
    else if (a <= 32)
        launch_gram<2, 4>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
    else
        launch_gram<2, 4>(c, X, ldx, a, Y, ldy, b, m, rps, nslab, c->ws);
    RAILS_LAUNCH(k_reduce_partials, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, c->stream, c->ws, nslab, (int64_t)n, C_dev);
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

extern "C" int rails_gram(rails_ctx *c, const rails_panel *X, int xc0, int a, const rails_panel *Y, int yc0, int b, double *C_host,
                          int ldc)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    rails_slow_guard slow__(c, "rails_gram", a, b);
    RAILS_REQUIRE(c && X && Y, "rails_gram: null argument");
    RAILS_REQUIRE(a >= 0 && b >= 0 && xc0 >= 0 && yc0 >= 0 && xc0 + a <= X->cap && yc0 + b <= Y->cap,
                  "rails_gram: windows [%d,%d) / [%d,%d) outside capacities %d / %d", xc0, xc0 + a, yc0, yc0 + b, X->cap, Y->cap);
    RAILS_REQUIRE(X->m == Y->m, "rails_gram: row mismatch %lld vs %lld", (long long)X->m, (long long)Y->m);
    RAILS_REQUIRE(a == 0 || b == 0 || (C_host && ldc >= a), "rails_gram: bad output buffer (ldc %d < %d)", ldc, a);
    if (a == 0 || b == 0) return RAILS_OK;
    size_t n = (size_t)a * b;
    RAILS_TRY(rails_small_reserve(c, n * sizeof(double)));
    RAILS_TRY(rails_pinned_reserve(c, n * sizeof(double)));
    if (X->m > 0)
        RAILS_TRY(rails_gram_dev(c, X->d + xc0, X->ld, Y->d + yc0, Y->ld, X->m, a, b, c->small));
    else
        RAILS_HIP_CHECK(hipMemsetAsync(c->small, 0, n * sizeof(double), c->stream));
    RAILS_TRY(rails_allreduce_dev(c, c->small, n));
    RAILS_HIP_CHECK(hipMemcpyAsync(c->pinned, c->small, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RAILS_HIP_CHECK(rails_stream_sync(c));
    for (int j = 0; j < b; ++j) memcpy(C_host + (size_t)j * ldc, c->pinned + (size_t)j * a, sizeof(double) * a);
    return RAILS_OK;
}

// ---- deferred small results --------------------------------------------------------------------------------------------------
// The block orthogonalisation of the coordinate-space back end is a chain Gram -> update -> Gram -> Cholesky -> update ... whose small
// matrices the host only needs for bookkeeping.  With these entry points the chain runs on the device without the host in between:
// a Gram result stays in a slot of a device arena (and is copied to its pinned mirror, to be read after a later synchronisation), the
// next update takes its coefficients from the slot, the Cholesky factor of a w x w slot is inverted into another slot by a small
// kernel.  Nothing here synchronises.

namespace {

// rows [0, k) of the r columns of a slot (leading dimension ld) packed into k x r
__global__ void k_compact_cols(const double *__restrict__ src, int ld, int k, int r, double *__restrict__ dst)
{
    const int64_t n = (int64_t)k * r;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) dst[q] = src[(q / k) * ld + (q % k)];
}

// G (w x w, symmetric positive definite, w <= 48) -> M = D^-1 R^-1 with D = sqrt(diag G), R'R = D^-1 G D^-1 (upper triangular):
// X M has orthonormal columns when X'X = G.  One wave, lane j = column j: a right-looking Cholesky (row i of R, then the rank-one update
// of the columns behind it: w steps of at most w dependent LDS round trips each) and one back substitution per lane for R^-1 -- about
// ten microseconds at w = 17, where one thread working through the ~w^3 / 2 dependent operations took 110 (two of these sit on the
// device's critical path of every trip).  A G that is not positive definite gives NaNs (the host repeats the factorisation on its copy
// of G and decides).
__global__ __launch_bounds__(64) void k_small_chol(const double *__restrict__ G, int w, double *__restrict__ M)
{
    constexpr int LD = 49; // up to 48 columns (Expand size 32 + the prefetched random vector: 33), rows of S a bank apart
    __shared__ double S[48 * LD], Ri[48 * LD], d[48];
    const int t = threadIdx.x;
    const bool mine = t < w;
    for (int q = t; q < w * w; q += 64) S[(q / w) * LD + (q % w)] = G[q]; // S[col][row]
    __syncthreads();
    if (mine) d[t] = sqrt(S[t * LD + t]);
    __syncthreads();
    for (int q = t; q < w * w; q += 64) S[(q / w) * LD + (q % w)] /= d[q % w] * d[q / w];
    __syncthreads();
    // R[i][j] (i <= j) ends up at S[j][i]
    for (int i = 0; i < w; ++i) {
        const double piv = sqrt(S[i * LD + i]);
        double rij = 0.0;
        if (mine && t >= i) rij = (t == i) ? piv : S[t * LD + i] / piv;
        __syncthreads();
        if (mine && t >= i) S[t * LD + i] = rij;
        __syncthreads();
        if (mine && t > i)
            for (int l = i + 1; l <= t; ++l) S[t * LD + l] -= S[l * LD + i] * rij;
        __syncthreads();
    }
    // column t of R^-1 by back substitution (every lane its own column: no exchange)
    if (mine) {
        for (int l = 0; l < w; ++l) Ri[t * LD + l] = 0.0;
        for (int i = t; i >= 0; --i) {
            double s = (i == t) ? 1.0 : 0.0;
            for (int l = i + 1; l <= t; ++l) s -= S[l * LD + i] * Ri[t * LD + l];
            Ri[t * LD + i] = s / S[i * LD + i];
        }
    }
    __syncthreads();
    for (int q = t; q < w * w; q += 64) {
        const int j = q / w, i = q % w;
        M[q] = i <= j ? Ri[j * LD + i] / d[i] : 0.0;
    }
}

} // namespace

extern "C" int rails_deferred_reserve(rails_ctx *c, int nslots, int64_t doubles_per_slot)
{
    RAILS_REQUIRE(c && nslots >= 1 && doubles_per_slot >= 1, "rails_deferred_reserve: bad argument");
    hipSetDevice(c->device);
    if (c->defer_dev && c->defer_nslots >= nslots && c->defer_slot >= (size_t)doubles_per_slot) return RAILS_OK;
    RAILS_HIP_CHECK(rails_stream_sync(c));
    if (c->defer_dev) hipFree(c->defer_dev);
    if (c->defer_pin) hipHostFree(c->defer_pin);
    c->defer_dev = c->defer_pin = nullptr;
    const size_t slot = ((size_t)doubles_per_slot * 3 / 2 + 63) / 64 * 64, bytes = slot * (size_t)nslots * sizeof(double);
    c->n_dev_alloc++;
    RAILS_HIP_CHECK(hipMalloc((void **)&c->defer_dev, bytes));
    RAILS_HIP_CHECK(hipHostMalloc((void **)&c->defer_pin, bytes, hipHostMallocDefault));
    c->defer_slot = slot;
    c->defer_nslots = nslots;
    return RAILS_OK;
}

#define RAILS_SLOT_CHECK(slot, n, what)                                                                                                    \
    RAILS_REQUIRE(c && c->defer_dev && (slot) >= 0 && (slot) < c->defer_nslots && (size_t)(n) <= c->defer_slot,                            \
                  what ": slot %d / %lld doubles outside the arena (rails_deferred_reserve)", (int)(slot), (long long)(n))

// slot <- X[:, xc0:xc0+a]' * Y[:, yc0:yc0+b] (a x b, column-major, leading dimension a), summed over the ranks; also on its way to the
// pinned mirror of the slot
extern "C" int rails_gram_deferred(rails_ctx *c, const rails_panel *X, int xc0, int a, const rails_panel *Y, int yc0, int b, int slot)
{
    if (c) hipSetDevice(c->device);
    RAILS_REQUIRE(c && X && Y, "rails_gram_deferred: null argument");
    RAILS_REQUIRE(a >= 1 && b >= 1 && xc0 >= 0 && yc0 >= 0 && xc0 + a <= X->cap && yc0 + b <= Y->cap && X->m == Y->m, "rails_gram_deferred: bad windows");
    const size_t n = (size_t)a * b;
    RAILS_SLOT_CHECK(slot, n, "rails_gram_deferred");
    double *out = c->defer_dev + (size_t)slot * c->defer_slot;
    if (X->m > 0)
        RAILS_TRY(rails_gram_dev(c, X->d + xc0, X->ld, Y->d + yc0, Y->ld, X->m, a, b, out));
    else
        RAILS_HIP_CHECK(hipMemsetAsync(out, 0, n * sizeof(double), c->stream));
    RAILS_TRY(rails_allreduce_dev(c, out, n));
    RAILS_HIP_CHECK(hipMemcpyAsync(c->defer_pin + (size_t)slot * c->defer_slot, out, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return RAILS_OK;
}

// Y[:, yc0:yc0+r] = beta Y + alpha X[:, xc0:xc0+k] * C, C = rows [0, k) of the r columns held in a slot with leading dimension ld
extern "C" int rails_panel_gemm_deferred(rails_ctx *c, double alpha, const rails_panel *X, int xc0, int k, int slot, int ld, int r, double beta,
                                         rails_panel *Y, int yc0)
{
    if (c) hipSetDevice(c->device);
    RAILS_REQUIRE(c && X && Y, "rails_panel_gemm_deferred: null argument");
    RAILS_REQUIRE(k >= 1 && r >= 1 && r <= 256 && ld >= k && xc0 >= 0 && yc0 >= 0 && xc0 + k <= X->cap && yc0 + r <= Y->cap && X->m == Y->m,
                  "rails_panel_gemm_deferred: bad shapes");
    RAILS_SLOT_CHECK(slot, (size_t)ld * r, "rails_panel_gemm_deferred");
    if (X->d == Y->d) RAILS_REQUIRE(xc0 == yc0 || xc0 + k <= yc0 || yc0 + r <= xc0, "rails_panel_gemm_deferred: partially overlapping windows of one panel");
    if (X->m == 0) return RAILS_OK;
    const double *C = c->defer_dev + (size_t)slot * c->defer_slot;
    if (ld != k) {
        RAILS_TRY(rails_small_reserve(c, (size_t)k * r * sizeof(double)));
        RAILS_LAUNCH(k_compact_cols, dim3((unsigned)std::min<int64_t>(256, ((int64_t)k * r + 255) / 256)), dim3(256), 0, c->stream, C, ld, k, r, c->small);
        C = c->small;
    }
    return rails_panel_gemm_dev(c, alpha, X->d + xc0, X->ld, k, C, r, beta, Y->d + yc0, Y->ld, X->m);
}

// ---- first update and second projection of a block in ONE pass over the basis ----------------------------------------------------
// Round 2 of the block orthogonalisation (Stl path: the second sweep of src/StlWrapper.cpp:314-344) needs Y' = Y - X C1 and then
// C2 = X' Y'.  As two kernels that is two passes over X (2.8 GB each at 350 basis columns x 1M rows: 0.6 + 0.42 ms); here a workgroup
// takes 16 rows at a time, its four waves split X's columns in interleaved blocks of 16 and keep their part of the tile in registers:
//   step 1  partial Y' tile = X_part C_part on the MFMA (layout A: lane (row, k mod 4) holds four consecutive columns of its row),
//           summed over the waves through LDS, Y' written back and kept in LDS;
//   step 2  C2_part += X_part' Y' : the same registers, turned around through 2 KiB of LDS per wave into the layout in which the
//           ROWS are the contraction index (lane (column, row mod 4)).
// Columns beyond 16 of the update (the prefetched random vector that rides at the end of an A*W block: r = 17) are done with plain
// multiply-adds on the values the lanes hold anyway -- a second MFMA column tile for one column doubles the matrix work and makes the
// pass MFMA-bound.  The next tile's rows are loaded while this one is worked on.  Per-workgroup partial C2 tiles go to the workspace
// and are summed in a fixed order by k_reduce_partials, like every Gram matrix here.
template <int NBW, int NE>
__global__ __launch_bounds__(256, (NBW <= 6 ? 2 : 1)) void k_update_gram(double alpha, const double *__restrict__ X, int ldx, int k, const double *__restrict__ C, int r,
                                                     double *Yp, int ldy, int r2, int64_t m, int64_t ntiles, double *__restrict__ partial)
{
    constexpr int KMAX = 64 * NBW, RL = 20; // LDS row length of C (doubles): 16 + 4 keeps the four k-groups of a read on disjoint banks
    __shared__ double Cs[KMAX * RL];
    __shared__ double red[4][4][64]; // step-1 partial tiles: [wave][result register][lane]
    __shared__ double rede[4][1][16];
    __shared__ double awL[16 * 16];  // the updated tile, [row][column], zero beyond r2 columns / m rows
    __shared__ __attribute__((aligned(16))) double T[4][16 * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    for (int idx = threadIdx.x; idx < KMAX * RL; idx += 256) {
        const int kl = idx / RL, j = idx % RL;
        Cs[idx] = (kl < k && j < r) ? C[kl + (int64_t)j * k] : 0.0;
    }
    v4f64 acc2[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b) acc2[b] = (v4f64){0.0, 0.0, 0.0, 0.0};

    // Loads without branches and without selects (either makes the compiler wait for a block's loads before it issues the next
    // block's: six memory latencies per tile): addresses are clamped into the panel and what comes back from outside the operands is
    // multiplied by zeros -- rows past m repeat row m - 1 and meet the zero rows of the Y' tile in step 2 (their step-1 results are
    // not stored); columns k .. k4 - 1 (k rounded up to 4) are the leading columns of Y itself (the host checks that Y sits right
    // behind X and is that wide), columns from k4 on repeat column 0, and both meet zero rows of C (rows of C2 that are not written).
    const int k4 = (k + 3) & ~3;
    auto fetch = [&](int64_t tile, double (*xs)[4]) {
        const int64_t myrow = (tile < ntiles ? tile : ntiles - 1) * 16 + li;
        const double *xrow = X + (myrow < m ? myrow : m - 1) * ldx;
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int kcol = (b * 4 + wave) * 16 + 4 * kk;
            const double *src = xrow + (kcol < k4 ? kcol : 0);
            const v2f64 t0 = *reinterpret_cast<const v2f64 *>(src);
            const v2f64 t1 = *reinterpret_cast<const v2f64 *>(src + 2);
            xs[b][0] = t0.x;
            xs[b][1] = t0.y;
            xs[b][2] = t1.x;
            xs[b][3] = t1.y;
        }
    };
    // this thread's entry of a tile of Y: result register `wave` of lane `lane` is row kk + 4 wave, column li; the threads that finish
    // the columns beyond 16: row erow, column 16 + ene
    const int erow = threadIdx.x & 15, ene = threadIdx.x >> 4;
    auto fetch_y = [&](int64_t tile, double &yv, double &ye) {
        const int64_t t16 = (tile < ntiles ? tile : ntiles - 1) * 16;
        const int64_t yrow = t16 + kk + 4 * wave, er = t16 + erow;
        yv = Yp[(yrow < m ? yrow : m - 1) * ldy + (li < r ? li : 0)]; // (used only where it is valid)
        ye = NE > 0 ? Yp[(er < m ? er : m - 1) * ldy + (16 + ene < r ? 16 + ene : 0)] : 0.0;
    };
    // one tile: `xs`, `yv`, `ye` were requested while the tile before was worked on; the next tile's go out first (two register sets
    // that swap roles: no copies, and a wait for this tile's values does not wait for the next one's)
    auto work = [&](int64_t tile, double (*xs)[4], double yv, double ye, double (*xn)[4], double &yvn, double &yen) {
        const int64_t row0 = tile * 16;
        fetch(tile + gridDim.x, xn); // (zeros past the last tile)
        fetch_y(tile + gridDim.x, yvn, yen);
        const int64_t yrow = row0 + kk + 4 * wave;
        const bool yok = yrow < m && li < r;
        const bool eok = ene < NE && 16 + ene < r && row0 + erow < m;
        // step 1
        v4f64 p = (v4f64){0.0, 0.0, 0.0, 0.0};
        double e[NE > 0 ? NE : 1];
#pragma unroll
        for (int q = 0; q < NE; ++q) e[q] = 0.0;
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double *crow = &Cs[((b * 4 + wave) * 16 + 4 * kk + s) * RL];
                p = mfma_f64(xs[b][s], crow[li], p);
#pragma unroll
                for (int q = 0; q < NE; ++q) e[q] += xs[b][s] * crow[16 + q];
            }
#pragma unroll
        for (int v = 0; v < 4; ++v) red[wave][v][lane] = p[v];
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            double t = e[q];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (kk == 0) rede[wave][q][li] = t;
        }
        __syncthreads();
        {
            const double sum = ((red[0][wave][lane] + red[1][wave][lane]) + red[2][wave][lane]) + red[3][wave][lane];
            const double ynew = yv + alpha * sum;
            if (yok) Yp[yrow * ldy + li] = ynew;
            awL[(kk + 4 * wave) * 16 + li] = (yok && li < r2) ? ynew : 0.0;
            if (NE > 0 && eok) {
                const double se = ((rede[0][ene][erow] + rede[1][ene][erow]) + rede[2][ene][erow]) + rede[3][ene][erow];
                Yp[(row0 + erow) * ldy + 16 + ene] = ye + alpha * se;
            }
        }
        __syncthreads();
        // step 2
        double bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = awL[(4 * q + kk) * 16 + li];
        double *Tw = T[wave];
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            *reinterpret_cast<v2f64 *>(&Tw[li * 16 + 4 * kk]) = (v2f64){xs[b][0], xs[b][1]};
            *reinterpret_cast<v2f64 *>(&Tw[li * 16 + 4 * kk + 2]) = (v2f64){xs[b][2], xs[b][3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double a4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a4[q] = Tw[(4 * q + kk) * 16 + li];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) acc2[b] = mfma_f64(a4[q], bq[q], acc2[b]);
        }
    };
    double xa[NBW][4], xb[NBW][4], yva, yea, yvb, yeb;
    fetch(blockIdx.x, xa);
    fetch_y(blockIdx.x, yva, yea);
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += 2 * (int64_t)gridDim.x) {
        work(tile, xa, yva, yea, xb, yvb, yeb);
        if (tile + gridDim.x < ntiles) work(tile + gridDim.x, xb, yvb, yeb, xa, yva, yea);
    }
    double *P = partial + (int64_t)blockIdx.x * k * r2;
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int ci = (b * 4 + wave) * 16 + kk + 4 * v; // D row -> X column
            if (ci < k && li < r2) P[ci + (int64_t)li * k] = acc2[b][v];
        }
}

template <int NBW>
static void launch_update_gram(rails_ctx *c, int ne, unsigned grid, double alpha, const double *X, int ldx, int k, const double *C, int r, double *Y, int ldy, int r2,
                               int64_t m, int64_t ntiles)
{
    if (ne == 0)
        RAILS_LAUNCH((k_update_gram<NBW, 0>), dim3(grid), dim3(256), 0, c->stream, alpha, X, ldx, k, C, r, Y, ldy, r2, m, ntiles, c->ws);
    else
        RAILS_LAUNCH((k_update_gram<NBW, 1>), dim3(grid), dim3(256), 0, c->stream, alpha, X, ldx, k, C, r, Y, ldy, r2, m, ntiles, c->ws);
}

// Y[:, yc0:yc0+r] += alpha X[:, xc0:xc0+k] C (C on the host, k x r, leading dimension ldc), then slot <- X[:, xc0:xc0+k]' Y[:, yc0:yc0+r2]
// (k x r2, leading dimension k, summed over the ranks, on its way to the slot's pinned mirror).  One pass over X where the shapes
// allow (r <= 17, r2 <= 16, 32 <= k <= 512), the two separate kernels otherwise.
extern "C" int rails_update_gram_deferred(rails_ctx *c, double alpha, const rails_panel *X, int xc0, int k, const double *C_host, int ldc, int r,
                                          rails_panel *Y, int yc0, int r2, int slot)
{
    if (c) hipSetDevice(c->device);
    rails_slow_guard slow__(c, "rails_update_gram_deferred", k, r);
    RAILS_REQUIRE(c && X && Y && C_host, "rails_update_gram_deferred: null argument");
    RAILS_REQUIRE(k >= 1 && r >= 1 && r <= 256 && r2 >= 1 && r2 <= r && ldc >= k && xc0 >= 0 && yc0 >= 0 && xc0 + k <= X->cap && yc0 + r <= Y->cap && X->m == Y->m,
                  "rails_update_gram_deferred: bad shapes");
    if (X->d == Y->d) RAILS_REQUIRE(xc0 + k <= yc0 || yc0 + r <= xc0, "rails_update_gram_deferred: overlapping windows of one panel");
    const size_t n = (size_t)k * r2;
    RAILS_SLOT_CHECK(slot, n, "rails_update_gram_deferred");
    double *out = c->defer_dev + (size_t)slot * c->defer_slot;
    if (X->m == 0) {
        RAILS_HIP_CHECK(hipMemsetAsync(out, 0, n * sizeof(double), c->stream));
    } else {
        const size_t nc = (size_t)k * r;
        RAILS_TRY(rails_small_reserve(c, nc * sizeof(double)));
        RAILS_TRY(rails_pinned_begin_write(c, nc * sizeof(double)));
        for (int j = 0; j < r; ++j) memcpy(c->pinned + (size_t)j * k, C_host + (size_t)j * ldc, sizeof(double) * k);
        RAILS_HIP_CHECK(hipMemcpyAsync(c->small, c->pinned, nc * sizeof(double), hipMemcpyHostToDevice, c->stream));
        RAILS_TRY(rails_pinned_end_write(c));
        static const int fused_env = getenv("RAILS_FUSED_UPDATE_GRAM") ? atoi(getenv("RAILS_FUSED_UPDATE_GRAM")) : 1;
        const double *Xp = X->d + xc0;
        double *Ypp = Y->d + yc0;
        // (16-byte loads of four columns at a time: aligned rows, and the columns between k and k rounded up to 4 are Y's own)
        const int k4 = (k + 3) & ~3;
        const bool rows_ok = ((((uintptr_t)Xp) & 15) == 0 && (X->ld % 2) == 0) && (k4 == k || (X->d == Y->d && yc0 == xc0 + k && r >= k4 - k));
        if (fused_env && rows_ok && r <= 17 && r2 <= 16 && k >= 32 && k <= 512 && X->m >= 4096) {
            const int64_t ntiles = (X->m + 15) / 16;
            const unsigned grid = (unsigned)std::min<int64_t>(ntiles, 2 * (int64_t)std::max(c->num_cu, 1));
            const int ne = r > 16 ? r - 16 : 0;
            RAILS_TRY(rails_ws_reserve(c, (size_t)grid * n * sizeof(double)));
            if (k <= 256)
                launch_update_gram<4>(c, ne, grid, alpha, Xp, X->ld, k, c->small, r, Ypp, Y->ld, r2, X->m, ntiles);
            else if (k <= 384)
                launch_update_gram<6>(c, ne, grid, alpha, Xp, X->ld, k, c->small, r, Ypp, Y->ld, r2, X->m, ntiles);
            else
                launch_update_gram<8>(c, ne, grid, alpha, Xp, X->ld, k, c->small, r, Ypp, Y->ld, r2, X->m, ntiles);
            RAILS_LAUNCH(k_reduce_partials, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, c->stream, c->ws, (int64_t)grid, (int64_t)n, out);
            RAILS_HIP_CHECK(hipGetLastError());
            c->n_update_gram_fused++;
        } else {
            RAILS_TRY(rails_panel_gemm_dev(c, alpha, Xp, X->ld, k, c->small, r, 1.0, Ypp, Y->ld, X->m));
            RAILS_TRY(rails_gram_dev(c, Xp, X->ld, Ypp, Y->ld, X->m, k, r2, out));
        }
    }
    RAILS_TRY(rails_allreduce_dev(c, out, n));
    RAILS_HIP_CHECK(hipMemcpyAsync(c->defer_pin + (size_t)slot * c->defer_slot, out, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return RAILS_OK;
}

// slot_out <- D^-1 R^-1 for the w x w Gram matrix in slot_in (see k_small_chol)
extern "C" int rails_chol_inverse_deferred(rails_ctx *c, int slot_in, int w, int slot_out)
{
    if (c) hipSetDevice(c->device);
    RAILS_REQUIRE(c && w >= 1 && w <= 48 && slot_in != slot_out, "rails_chol_inverse_deferred: bad argument (w = %d)", w);
    RAILS_SLOT_CHECK(slot_in, (size_t)w * w, "rails_chol_inverse_deferred");
    RAILS_SLOT_CHECK(slot_out, (size_t)w * w, "rails_chol_inverse_deferred");
    RAILS_LAUNCH(k_small_chol, dim3(1), dim3(64), 0, c->stream, c->defer_dev + (size_t)slot_in * c->defer_slot, w, c->defer_dev + (size_t)slot_out * c->defer_slot);
    RAILS_HIP_CHECK(hipGetLastError());
    RAILS_HIP_CHECK(hipMemcpyAsync(c->defer_pin + (size_t)slot_out * c->defer_slot, c->defer_dev + (size_t)slot_out * c->defer_slot, (size_t)w * w * sizeof(double),
                                   hipMemcpyDeviceToHost, c->stream)); // (the mirror: diagnostics and tests)
    return RAILS_OK;
}

// the first n doubles of a slot's pinned mirror (valid once the stream has been synchronised after the call that filled the slot)
extern "C" int rails_deferred_fetch(rails_ctx *c, int slot, int64_t n, double *host_out)
{
    RAILS_REQUIRE(host_out && n >= 0, "rails_deferred_fetch: bad argument");
    RAILS_SLOT_CHECK(slot, n, "rails_deferred_fetch");
    memcpy(host_out, c->defer_pin + (size_t)slot * c->defer_slot, (size_t)n * sizeof(double));
    return RAILS_OK;
}

int rails_panel_gemm_dev(rails_ctx *c, double alpha, const double *X, int ldx, int k, const double *C_dev, int r, double beta,
                         double *Y, int ldy, int64_t m)
{
    if (r <= 0 || m <= 0) return RAILS_OK;
    int vec_ok = ((((uintptr_t)X) & 15) == 0 && (ldx % 2) == 0) ? 1 : 0;
    int tr = (r + 15) / 16;
    if (tr <= 1)
        launch_pg<1, 32>(c, alpha, X, ldx, k, C_dev, r, beta, Y, ldy, m, vec_ok);
    else if (tr <= 2)
        launch_pg<2, 32>(c, alpha, X, ldx, k, C_dev, r, beta, Y, ldy, m, vec_ok);
    else if (tr <= 4)
        launch_pg<4, 32>(c, alpha, X, ldx, k, C_dev, r, beta, Y, ldy, m, vec_ok);
    else if (tr <= 8)
        launch_pg<8, 32>(c, alpha, X, ldx, k, C_dev, r, beta, Y, ldy, m, vec_ok);
    else
        launch_pg<16, 16>(c, alpha, X, ldx, k, C_dev, r, beta, Y, ldy, m, vec_ok);
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

// ---- the plain wide GEMM of a basis rotation through rocBLAS -----------------------------------------------------------------------
// P2 = P Q with m = 1M rows, k and r in the hundreds, is a plain DGEMM, compute-bound: rocBLAS runs it at 44 TFLOP/s where
// k_panel_gemm<8, 32> (built around the bandwidth-bound narrow shapes) reaches 30 -- 3.9 instead of 5.8 ms per restart at k = 324,
// r = 268.  librocblas is resolved with dlopen (a library GEMM is what the platform's BLAS is for; everything else in this file is
// hand-written); creating the handle takes 0.3 s, so a solver that is going to rotate asks for it up front
// (rails_ctx_enable_library_gemm), not in the middle of a solve.  RAILS_WIDE_GEMM=own keeps the hand-written kernel.
namespace {
struct RocblasApi {
    void *handle = nullptr;
    int (*create)(void **) = nullptr;
    int (*destroy)(void *) = nullptr;
    int (*set_stream)(void *, hipStream_t) = nullptr;
    int (*dgemm)(void *, int, int, int, int, int, const double *, const double *, int, const double *, int, const double *, double *, int) = nullptr;
    bool tried = false, ok = false;
};
RocblasApi g_rocblas;
std::mutex g_rocblas_mutex;

bool load_rocblas()
{
    std::lock_guard<std::mutex> lock(g_rocblas_mutex);
    if (g_rocblas.tried) return g_rocblas.ok;
    g_rocblas.tried = true;
    // opt-in since round 3: the hand-written one-pass kernel (k_panel_gemm_wide) is the default
    const char *e = getenv("RAILS_WIDE_GEMM");
    if (!e || strcmp(e, "rocblas")) return false;
    const char *names[] = {getenv("RAILS_ROCBLAS_LIB"), "librocblas.so", "/opt/rocm/lib/librocblas.so", "librocblas.so.5", "librocblas.so.4"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        if ((g_rocblas.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!g_rocblas.handle) return false;
    g_rocblas.create = (int (*)(void **))dlsym(g_rocblas.handle, "rocblas_create_handle");
    g_rocblas.destroy = (int (*)(void *))dlsym(g_rocblas.handle, "rocblas_destroy_handle");
    g_rocblas.set_stream = (int (*)(void *, hipStream_t))dlsym(g_rocblas.handle, "rocblas_set_stream");
    g_rocblas.dgemm = (decltype(g_rocblas.dgemm))dlsym(g_rocblas.handle, "rocblas_dgemm");
    g_rocblas.ok = g_rocblas.create && g_rocblas.destroy && g_rocblas.set_stream && g_rocblas.dgemm;
    return g_rocblas.ok;
}
constexpr int ROCBLAS_OP_N = 111, ROCBLAS_OP_T = 112;
} // namespace

// One handle per process (for the device of the first context that asks).  rocblas_create_handle takes 0.3 s and is NOT done behind the
// caller's back on another thread (tried: a native program that starts using the GPU at once crashed now and then while rocBLAS was
// loading its code objects): whoever wants the library path asks for it at a convenient moment -- bench.py when it sets up.  The
// coordinate-space back end used to ask at the second restart a process saw; in a cold process that was a stall of seconds (the
// library and its kernels come from disk) for 2 ms saved per restart, so it no longer does.
namespace {
struct LibraryGemm {
    std::atomic<int> state{0}; // 0 not asked for, 2 ready, 3 not available
    int device = -1;
    void *handle = nullptr;
    std::mutex use; // creation; set_stream + dgemm of one caller at a time
};
LibraryGemm g_libgemm;
} // namespace

extern "C" int rails_ctx_enable_library_gemm(rails_ctx *c)
{
    RAILS_REQUIRE(c, "rails_ctx_enable_library_gemm: null context");
    std::lock_guard<std::mutex> lock(g_libgemm.use);
    if (g_libgemm.state.load() != 0) return RAILS_OK;
    g_libgemm.device = c->device;
    void *h = nullptr;
    if (!load_rocblas() || hipSetDevice(c->device) != hipSuccess || g_rocblas.create(&h) != 0 || !h) {
        g_libgemm.state = 3;
        return RAILS_OK; // without the library the hand-written kernel does the work
    }
    g_libgemm.handle = h;
    g_libgemm.state = 2;
    return RAILS_OK;
}

// 1 when rails_panel_gemm_wide on this context goes through the library from now on
extern "C" int rails_ctx_library_gemm_ready(const rails_ctx *c) { return c && g_libgemm.state.load() == 2 && g_libgemm.device == c->device ? 1 : 0; }

void rails_library_gemm_release(rails_ctx *) {}

// Y[:, yc0:yc0+r] = beta * Y + alpha * X[:, xc0:xc0+k] * C for any r: C goes to the device in ONE upload and the product is launched in
// slices of 128 output columns (the faster tile shape) without the host waiting in between -- rails_panel_gemm re-uses one staging
// buffer per call, so a loop over it makes the host wait for every slice's kernel but the last (the basis rotation P <- P Q of the
// coordinate-space back end: 2.6 ms of host time per restart).  X and Y must be different panels or disjoint windows.
extern "C" int rails_panel_gemm_wide(rails_ctx *c, double alpha, const rails_panel *X, int xc0, int k, const double *C_host, int ldc, int r,
                                     double beta, rails_panel *Y, int yc0)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    rails_slow_guard slow__(c, "rails_panel_gemm_wide", k, r);
    RAILS_REQUIRE(c && X && Y, "rails_panel_gemm_wide: null argument");
    RAILS_REQUIRE(k >= 0 && r >= 0 && xc0 >= 0 && yc0 >= 0 && xc0 + k <= X->cap && yc0 + r <= Y->cap,
                  "rails_panel_gemm_wide: windows [%d,%d) / [%d,%d) outside capacities %d / %d", xc0, xc0 + k, yc0, yc0 + r, X->cap, Y->cap);
    RAILS_REQUIRE(X->m == Y->m, "rails_panel_gemm_wide: row mismatch %lld vs %lld", (long long)X->m, (long long)Y->m);
    RAILS_REQUIRE(k == 0 || r == 0 || (C_host && ldc >= k), "rails_panel_gemm_wide: bad coefficient matrix (ldc %d < %d)", ldc, k);
    if (X->d == Y->d) RAILS_REQUIRE((xc0 + k <= yc0) || (yc0 + r <= xc0), "rails_panel_gemm_wide: overlapping windows of one panel");
    if (r == 0 || X->m == 0) return RAILS_OK;
    if (k == 0) {
        if (beta == 0.0) return rails_panel_fill(c, Y, yc0, r, 0.0);
        if (beta == 1.0) return RAILS_OK;
        return rails_panel_scale(c, Y, yc0, r, beta);
    }
    size_t n = (size_t)k * r;
    RAILS_TRY(rails_small_reserve(c, n * sizeof(double)));
    RAILS_TRY(rails_pinned_begin_write(c, n * sizeof(double)));
    for (int j = 0; j < r; ++j) memcpy(c->pinned + (size_t)j * k, C_host + (size_t)j * ldc, sizeof(double) * k);
    RAILS_HIP_CHECK(hipMemcpyAsync(c->small, c->pinned, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RAILS_TRY(rails_pinned_end_write(c));
    if (k >= 64 && r >= 64 && X->m < 0x7fffffffLL && rails_ctx_library_gemm_ready(c)) {
        // (opt-in, RAILS_WIDE_GEMM=rocblas + rails_ctx_enable_library_gemm) row-major panels are column-major matrices transposed:
        // Y' (r x m, ld) = alpha C' (r x k) X' (k x m, ld) + beta Y'
        const int m32 = (int)X->m;
        std::lock_guard<std::mutex> lock(g_libgemm.use);
        int rc = g_rocblas.set_stream(g_libgemm.handle, c->stream);
        if (rc == 0) rc = g_rocblas.dgemm(g_libgemm.handle, ROCBLAS_OP_T, ROCBLAS_OP_N, r, m32, k, &alpha, c->small, k, X->d + xc0, X->ld, &beta, Y->d + yc0, Y->ld);
        if (rc == 0) return RAILS_OK;
        rails_set_error("rails_panel_gemm_wide: rocblas_dgemm failed with status %d", rc);
        return RAILS_EHIP;
    }
    // slices of equal width, at most 288 columns (18 tiles of 16: what a wave's registers hold), X read once per slice
    static const int wide_env = getenv("RAILS_PANEL_GEMM_WIDE") ? atoi(getenv("RAILS_PANEL_GEMM_WIDE")) : 1;
    const int vec_ok = ((((uintptr_t)(X->d + xc0)) & 15) == 0 && (X->ld % 2) == 0) ? 1 : 0;
    if (wide_env && r > 64 && k >= 32) {
        const int nslices = (r + 287) / 288, width = ((r + nslices - 1) / nslices + 15) / 16 * 16;
        const size_t ct_doubles = (size_t)((k + 31) / 32 * 32) * (16 * 18 + 4);
        RAILS_TRY(rails_ws_reserve(c, (size_t)nslices * ct_doubles * sizeof(double)));
        int slice = 0;
        for (int j0 = 0; j0 < r; j0 += width, ++slice) {
            const int nc = std::min(width, r - j0), tr = (nc + 31) / 32 * 2;
            const double *Cj = c->small + (size_t)j0 * k;
            double *Yj = Y->d + yc0 + j0, *ct = c->ws + (size_t)slice * ct_doubles;
            switch (tr) {
            case 2: case 4: case 6: RAILS_TRY((launch_pg_wide<6>(c, alpha, X->d + xc0, X->ld, k, Cj, nc, beta, Yj, Y->ld, X->m, vec_ok, ct))); break;
            case 8: RAILS_TRY((launch_pg_wide<8>(c, alpha, X->d + xc0, X->ld, k, Cj, nc, beta, Yj, Y->ld, X->m, vec_ok, ct))); break;
            case 10: RAILS_TRY((launch_pg_wide<10>(c, alpha, X->d + xc0, X->ld, k, Cj, nc, beta, Yj, Y->ld, X->m, vec_ok, ct))); break;
            case 12: RAILS_TRY((launch_pg_wide<12>(c, alpha, X->d + xc0, X->ld, k, Cj, nc, beta, Yj, Y->ld, X->m, vec_ok, ct))); break;
            case 14: RAILS_TRY((launch_pg_wide<14>(c, alpha, X->d + xc0, X->ld, k, Cj, nc, beta, Yj, Y->ld, X->m, vec_ok, ct))); break;
            case 16: RAILS_TRY((launch_pg_wide<16>(c, alpha, X->d + xc0, X->ld, k, Cj, nc, beta, Yj, Y->ld, X->m, vec_ok, ct))); break;
            default: RAILS_TRY((launch_pg_wide<18>(c, alpha, X->d + xc0, X->ld, k, Cj, nc, beta, Yj, Y->ld, X->m, vec_ok, ct))); break;
            }
        }
        RAILS_HIP_CHECK(hipGetLastError());
        return RAILS_OK;
    }
    for (int j0 = 0; j0 < r; j0 += 128) {
        const int nc = std::min(128, r - j0);
        RAILS_TRY(rails_panel_gemm_dev(c, alpha, X->d + xc0, X->ld, k, c->small + (size_t)j0 * k, nc, beta, Y->d + yc0 + j0, Y->ld, X->m));
    }
    return RAILS_OK;
}

extern "C" int rails_panel_gemm(rails_ctx *c, double alpha, const rails_panel *X, int xc0, int k, const double *C_host, int ldc,
                                int r, double beta, rails_panel *Y, int yc0)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    rails_slow_guard slow__(c, "rails_panel_gemm", k, r);
    RAILS_REQUIRE(c && X && Y, "rails_panel_gemm: null argument");
    RAILS_REQUIRE(k >= 0 && r >= 0 && xc0 >= 0 && yc0 >= 0 && xc0 + k <= X->cap && yc0 + r <= Y->cap,
                  "rails_panel_gemm: windows [%d,%d) / [%d,%d) outside capacities %d / %d", xc0, xc0 + k, yc0, yc0 + r, X->cap, Y->cap);
    RAILS_REQUIRE(X->m == Y->m, "rails_panel_gemm: row mismatch %lld vs %lld", (long long)X->m, (long long)Y->m);
    RAILS_REQUIRE(r <= 256, "rails_panel_gemm: r = %d > 256 output columns per call", r);
    RAILS_REQUIRE(k == 0 || r == 0 || (C_host && ldc >= k), "rails_panel_gemm: bad coefficient matrix (ldc %d < %d)", ldc, k);
    if (X->d == Y->d) {
        bool same = (xc0 == yc0);
        bool disjoint = (xc0 + k <= yc0) || (yc0 + r <= xc0);
        RAILS_REQUIRE(same || disjoint, "rails_panel_gemm: partially overlapping windows of one panel");
    }
    if (r == 0 || X->m == 0) return RAILS_OK;
    if (k == 0) {
        if (beta == 0.0) return rails_panel_fill(c, Y, yc0, r, 0.0);
        if (beta == 1.0) return RAILS_OK;
        return rails_panel_scale(c, Y, yc0, r, beta);
    }
    size_t n = (size_t)k * r;
    // asynchronous: the coefficient block goes through the pinned staging buffer (the next host write into it waits for this
    // upload, rails_pinned_begin_write) into the context's small device buffer (re-used in stream order)
    RAILS_TRY(rails_small_reserve(c, n * sizeof(double)));
    RAILS_TRY(rails_pinned_begin_write(c, n * sizeof(double)));
    for (int j = 0; j < r; ++j) memcpy(c->pinned + (size_t)j * k, C_host + (size_t)j * ldc, sizeof(double) * k);
    RAILS_HIP_CHECK(hipMemcpyAsync(c->small, c->pinned, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RAILS_TRY(rails_pinned_end_write(c));
    return rails_panel_gemm_dev(c, alpha, X->d + xc0, X->ld, k, c->small, r, beta, Y->d + yc0, Y->ld, X->m);
}
