// lanczos.hip -- fused residual Lanczos (src/LyapunovSolver.hpp:367-447) for gfx950.
//
// Lanczos on the implicit symmetric operator  R = AV T MV^T + MV T AV^T + B B^T  (MV == V when
// M = I).  The reference makes 4 passes over the m x k panels per step (V^T q, AV Z, AV^T q, V Z)
// plus B, alpha, beta and axpy passes.  Here ONE pass over P = [AV MV B] does a whole step:
//
//   with c = P^T q_i known (from the previous pass):   g = [T c_MV ; T c_AV ; c_B]
//        alpha_i = q_i^T R q_i = c_AV.g_AV + c_MV.g_MV + c_B.c_B           (no pass needed)
//   pass i, per row:   r = P_row . g - alpha_i q_i - beta_{i-1} q_{i-1}     (= un-normalised q_{i+1})
//                      c' += P_row * r ;  rr += r*r                          (row-local, same pass)
//   then               beta_i = sqrt(rr),  c_{next} = c' / beta_i.
//
// HBM traffic per step = (2k + p + ~4) * m * 8 bytes: the compulsory single read of the panels.
// The Lanczos vectors live in a column-major side buffer (one contiguous vector per step) so the
// per-row scalar traffic is coalesced.  alpha, beta and the breakdown test (beta < 1e-14,
// :419-426) stay on the device; the host reads H back once at the end.
#include "rails_internal.h"

#include <algorithm>
#include <cmath>

struct rails_lanczos_state {
    double *Qc = nullptr; // (L+2) vectors of length mpad, column-major
    size_t qc_bytes = 0;
    int64_t m = 0, mpad = 0;
    int steps = 0;
    int L = 0;
    double *small = nullptr; // T, coefficients, reduced sums, state, alphas, betas
    size_t small_bytes = 0;
};

// one state per context: the Lanczos vectors of the context's last run (several contexts may live in one process,
// e.g. one per thread in the single-GPU rank emulation of tests/test_gpu_partition.py)
static rails_lanczos_state &lz_state(rails_ctx *c)
{
    if (!c->lz) c->lz = new rails_lanczos_state();
    return *static_cast<rails_lanczos_state *>(c->lz);
}

namespace {

typedef double v2f64 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// 64-lane sum, result uniform; fixed association: quads, 8s, 16s by DPP, then the four rows.
__device__ __forceinline__ double wave_sum(double x)
{
    x += dpp_f64<0xB1>(x);  // quad_perm [1,0,3,2]
    x += dpp_f64<0x4E>(x);  // quad_perm [2,3,0,1]
    x += dpp_f64<0x141>(x); // row_half_mirror
    x += dpp_f64<0x140>(x); // row_mirror
    double s0 = readlane_f64(x, 0), s1 = readlane_f64(x, 16), s2 = readlane_f64(x, 32), s3 = readlane_f64(x, 48);
    return (s0 + s1) + (s2 + s3);
}

struct LzArgs {
    const double *AV;
    int ldav;
    const double *MV;
    int ldmv;
    const double *B;
    int ldb;
    int k, p;
    int av_room, mv_room, b_room; // doubles readable from the window start to the end of the padded row
    int64_t m, mpad;
    double *Qc;
    int step;             // -1 = init pass (r := raw q_0)
    const double *coef;   // [k (for AV) | k (for MV) | p (for B)]
    const double *state;  // alpha, beta_prev, inv_beta, done, steps
    double *partial;      // [nblocks][2k+p+1]
};

// lane l owns column pairs (2(l+64c), 2(l+64c)+1), c < NCH, of AV and of MV, and pair l of B.
template <int NCH, int U>
__global__ __launch_bounds__(256) void k_lanczos_pass(LzArgs a)
{
    __shared__ double red[4][(4 * NCH + 2) * 64 + 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool init = a.step < 0;
    const double done = a.state[3];
    const int ncoef = 2 * a.k + a.p + 1;
    double *myp = a.partial + (int64_t)blockIdx.x * ncoef;
    if (done != 0.0) {
        for (int i = threadIdx.x; i < ncoef; i += 256) myp[i] = 0.0;
        return;
    }
    const double alpha = init ? 0.0 : a.state[0];
    const double betap = init ? 0.0 : a.state[1];
    const double invb = init ? 1.0 : a.state[2];

    v2f64 gav[NCH], gmv[NCH], gb;
    bool ok0[NCH], ok1[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int col = 2 * (lane + 64 * c);
        ok0[c] = col < a.k;
        ok1[c] = col + 1 < a.k;
        gav[c].x = (!init && ok0[c]) ? a.coef[col] : 0.0;
        gav[c].y = (!init && ok1[c]) ? a.coef[col + 1] : 0.0;
        gmv[c].x = (!init && ok0[c]) ? a.coef[a.k + col] : 0.0;
        gmv[c].y = (!init && ok1[c]) ? a.coef[a.k + col + 1] : 0.0;
    }
    const bool bok0 = 2 * lane < a.p, bok1 = 2 * lane + 1 < a.p;
    // clamped (always valid, 16-B aligned) column offsets for the unconditional loads: lanes past the last column pair
    // re-read that pair (same cache line as their neighbour: no extra HBM traffic), never the padding beyond it
    int cav_off[NCH], cmv_off[NCH];
    {
        const int klast = a.k > 1 ? ((a.k - 1) & ~1) : 0;
        const int lim_av = klast < a.av_room - 2 ? klast : a.av_room - 2;
        const int lim_mv = klast < a.mv_room - 2 ? klast : a.mv_room - 2;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = 2 * (lane + 64 * c);
            cav_off[c] = col < lim_av ? col : lim_av;
            cmv_off[c] = col < lim_mv ? col : lim_mv;
        }
    }
    const int plast = a.p > 1 ? ((a.p - 1) & ~1) : 0;
    const int lim_b = plast < a.b_room - 2 ? plast : a.b_room - 2;
    const int cb_off = 2 * lane < lim_b ? 2 * lane : lim_b;
    gb.x = (!init && bok0) ? a.coef[2 * a.k + 2 * lane] : 0.0;
    gb.y = (!init && bok1) ? a.coef[2 * a.k + 2 * lane + 1] : 0.0;
    v2f64 cav[NCH], cmv[NCH], cb;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        cav[c] = (v2f64){0.0, 0.0};
        cmv[c] = (v2f64){0.0, 0.0};
    }
    cb = (v2f64){0.0, 0.0};
    double rr = 0.0;

    double *q_cur = a.Qc + (int64_t)(init ? 0 : a.step) * a.mpad;
    const double *q_prev = a.Qc + (int64_t)((init || a.step == 0) ? 0 : a.step - 1) * a.mpad;
    double *q_next = a.Qc + (int64_t)(a.step + 1) * a.mpad;
    const bool have_prev = (!init && a.step > 0);

    const int64_t ngroups = a.mpad / 64;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
        const int64_t row0 = grp * 64;
        double qn = q_cur[row0 + lane] * invb; // rows >= m hold zeros
        double qm = have_prev ? q_prev[row0 + lane] : 0.0;
        if (!init) q_cur[row0 + lane] = qn; // store the normalised q_i
        double rvec = 0.0;
        const int nrows = (int)((a.m - row0) < 64 ? (a.m - row0) : 64);
        // U rows per trip of the loop: their 2*NCH+1 row loads are all issued before the first reduction, the
        // U wave reductions are independent chains (rows are independent of each other)
        for (int j0 = 0; j0 < nrows; j0 += U) {
            // all row loads are UNCONDITIONAL (clamped row / column, values masked afterwards with selects): a load
            // under a lane-dependent branch makes hipcc wait vmcnt(0) at the join and serialises the loads
            v2f64 xav[U][NCH], xmv[U][NCH], xb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jr = (j0 + u) < nrows ? (j0 + u) : (nrows - 1);
                const int64_t row = row0 + jr;
                const double *pav = a.AV + row * a.ldav;
                const double *pmv = a.MV + row * a.ldmv;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    xav[u][c] = *reinterpret_cast<const v2f64 *>(pav + cav_off[c]);
                    xmv[u][c] = *reinterpret_cast<const v2f64 *>(pmv + cmv_off[c]);
                }
                xb[u] = *reinterpret_cast<const v2f64 *>(a.B + row * a.ldb + cb_off);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    xav[u][c].x = ok0[c] ? xav[u][c].x : 0.0;
                    xav[u][c].y = ok1[c] ? xav[u][c].y : 0.0;
                    xmv[u][c].x = ok0[c] ? xmv[u][c].x : 0.0;
                    xmv[u][c].y = ok1[c] ? xmv[u][c].y : 0.0;
                }
                xb[u].x = bok0 ? xb[u].x : 0.0;
                xb[u].y = bok1 ? xb[u].y : 0.0;
            }
            double r[U];
            if (init) {
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = ((j0 + u) < nrows) ? readlane_f64(qn, (j0 + u) & 63) : 0.0;
            } else {
                double t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    double tt = 0.0;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        tt = __builtin_fma(xav[u][c].x, gav[c].x, tt);
                        tt = __builtin_fma(xav[u][c].y, gav[c].y, tt);
                        tt = __builtin_fma(xmv[u][c].x, gmv[c].x, tt);
                        tt = __builtin_fma(xmv[u][c].y, gmv[c].y, tt);
                    }
                    tt = __builtin_fma(xb[u].x, gb.x, tt);
                    tt = __builtin_fma(xb[u].y, gb.y, tt);
                    t[u] = tt;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) t[u] = wave_sum(t[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int jj = (j0 + u) & 63;
                    const double qi = readlane_f64(qn, jj);
                    const double qmi = readlane_f64(qm, jj);
                    r[u] = ((j0 + u) < nrows) ? (t[u] - alpha * qi - betap * qmi) : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    cav[c].x = __builtin_fma(xav[u][c].x, r[u], cav[c].x);
                    cav[c].y = __builtin_fma(xav[u][c].y, r[u], cav[c].y);
                    cmv[c].x = __builtin_fma(xmv[u][c].x, r[u], cmv[c].x);
                    cmv[c].y = __builtin_fma(xmv[u][c].y, r[u], cmv[c].y);
                }
                cb.x = __builtin_fma(xb[u].x, r[u], cb.x);
                cb.y = __builtin_fma(xb[u].y, r[u], cb.y);
                rr = __builtin_fma(r[u], r[u], rr);
                rvec = (lane == j0 + u) ? r[u] : rvec;
            }
        }
        if (!init) q_next[row0 + lane] = rvec;
    }

    // block reduction in a fixed order: wave 0 += wave 1, 2, 3
    double *mine = red[wave];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        mine[(4 * c + 0) * 64 + lane] = cav[c].x;
        mine[(4 * c + 1) * 64 + lane] = cav[c].y;
        mine[(4 * c + 2) * 64 + lane] = cmv[c].x;
        mine[(4 * c + 3) * 64 + lane] = cmv[c].y;
    }
    mine[(4 * NCH + 0) * 64 + lane] = cb.x;
    mine[(4 * NCH + 1) * 64 + lane] = cb.y;
    if (lane == 0) mine[(4 * NCH + 2) * 64] = rr;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int slot = (4 * c + e) * 64 + lane;
                double s = ((red[0][slot] + red[1][slot]) + red[2][slot]) + red[3][slot];
                int col = 2 * (lane + 64 * c) + (e & 1);
                if (col < a.k) myp[(e < 2 ? 0 : a.k) + col] = s;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int slot = (4 * NCH + e) * 64 + lane;
            double s = ((red[0][slot] + red[1][slot]) + red[2][slot]) + red[3][slot];
            int col = 2 * lane + e;
            if (col < a.p) myp[2 * a.k + col] = s;
        }
        if (lane == 0) {
            int slot = (4 * NCH + 2) * 64;
            myp[2 * a.k + a.p] = ((red[0][slot] + red[1][slot]) + red[2][slot]) + red[3][slot];
        }
    }
}

// out[e] = sum over the blocks' partials, fixed order (16 interleaved strands, then strands 0..15)
__global__ __launch_bounds__(1024) void k_lz_reduce(const double *__restrict__ partial, int nblocks, int n, double *__restrict__ out)
{
    __shared__ double sh[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    double s = 0.0;
    if (e < n)
        for (int t = ty; t < nblocks; t += 16) s += partial[(int64_t)t * n + e];
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < n) {
        double r = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) r += sh[g][tx];
        out[e] = r;
    }
}

// One block.  sums = [c'_AV (k) | c'_MV (k) | c'_B (p) | rr] (already all-reduced).
// Writes the coefficients for the next pass, alpha/beta bookkeeping and the breakdown flag.
__global__ __launch_bounds__(1024) void k_lz_small(const double *__restrict__ sums, const double *__restrict__ T, int k, int p, int step,
                                                   double *__restrict__ coef, double *__restrict__ state, double *__restrict__ alphas,
                                                   double *__restrict__ betas)
{
    __shared__ double sh[1024];
    __shared__ double part1[4][256], part2[4][256];
    if (state[3] != 0.0) return;
    const int tid = threadIdx.x;
    const int jr = tid & 255, q = tid >> 8; // output row within a group of 256, quarter of the inner index
    const double rr = sums[2 * k + p];
    const double beta = sqrt(rr);
    const bool init = step < 0;
    if (!init && beta < 1e-14) { // src/LyapunovSolver.hpp:419-426
        if (tid == 0) {
            betas[step] = beta;
            state[3] = 1.0;
            state[4] = (double)(step + 1);
        }
        return;
    }
    const double inv = 1.0 / beta;
    // g_AV = T (c_MV * inv),  g_MV = T (c_AV * inv),  g_B = c_B * inv; the inner index is split over 4 strands that
    // are summed in a fixed order
    const int l0 = (int)(((int64_t)k * q) / 4), l1 = (int)(((int64_t)k * (q + 1)) / 4);
    for (int j0 = 0; j0 < k; j0 += 256) {
        const int j = j0 + jr;
        double s1 = 0.0, s2 = 0.0;
        if (j < k)
            for (int l = l0; l < l1; ++l) {
                const double t = T[j + (int64_t)l * k];
                s1 = __builtin_fma(t, sums[k + l] * inv, s1);
                s2 = __builtin_fma(t, sums[l] * inv, s2);
            }
        part1[q][jr] = s1;
        part2[q][jr] = s2;
        __syncthreads();
        if (q == 0 && j < k) {
            coef[j] = ((part1[0][jr] + part1[1][jr]) + part1[2][jr]) + part1[3][jr];
            coef[k + j] = ((part2[0][jr] + part2[1][jr]) + part2[2][jr]) + part2[3][jr];
        }
        __syncthreads();
    }
    for (int j = tid; j < p; j += 1024) coef[2 * k + j] = sums[2 * k + j] * inv;
    __threadfence_block();
    __syncthreads();
    // alpha_next = c_AV.g_AV + c_MV.g_MV + c_B.c_B
    double part = 0.0;
    for (int j = tid; j < k; j += 1024) {
        part = __builtin_fma(sums[j] * inv, coef[j], part);
        part = __builtin_fma(sums[k + j] * inv, coef[k + j], part);
    }
    for (int j = tid; j < p; j += 1024) {
        double cbv = sums[2 * k + j] * inv;
        part = __builtin_fma(cbv, cbv, part);
    }
    sh[tid] = part;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        const double alpha_next = sh[0];
        state[0] = alpha_next;
        state[1] = init ? 0.0 : beta;
        state[2] = inv;
        alphas[step + 1] = alpha_next;
        if (!init) {
            betas[step] = beta;
            state[4] = (double)(step + 1);
        }
    }
}

// Out[row, oc0 + j] = sum_l Qc[l][row] * S[l + j*lds],  j < w (<= 16 per launch chunk)
__global__ __launch_bounds__(256) void k_lz_vectors(const double *__restrict__ Qc, int64_t mpad, int64_t m, int steps,
                                                    const double *__restrict__ S, int lds, int w, double *__restrict__ Out, int ldo)
{
    extern __shared__ double Ss[]; // steps x 16
    for (int idx = threadIdx.x; idx < steps * 16; idx += blockDim.x) {
        int l = idx / 16, j = idx % 16;
        Ss[idx] = (j < w) ? S[l + (int64_t)j * lds] : 0.0;
    }
    __syncthreads();
    int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    for (int l = 0; l < steps; ++l) {
        double q = Qc[(int64_t)l * mpad + row];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_fma(q, Ss[l * 16 + j], acc[j]);
    }
    double *o = Out + row * ldo;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < w) o[j] = acc[j];
}

__global__ void k_lz_random(double *__restrict__ q, int64_t m, int64_t mpad, uint64_t seed, uint64_t stream, int64_t row0);

__device__ __forceinline__ uint64_t sm64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void k_lz_random(double *__restrict__ q, int64_t m, int64_t mpad, uint64_t seed, uint64_t stream, int64_t row0)
{
    uint64_t hs = sm64(seed ^ sm64(stream * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < mpad; r += (int64_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (r < m) {
            uint64_t h = sm64(hs ^ sm64((uint64_t)(row0 + r) * 0x9E3779B97F4A7C15ull + 1));
            double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
            v = 2.0 * u - 1.0;
        }
        q[r] = v;
    }
}

template <int NCH, int U>
void launch_pass_u(rails_ctx *c, const LzArgs &a, int *nblocks_io, bool size_only)
{
    if (size_only) { // grid = resident blocks only (every block walks the row groups with a grid stride)
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_lanczos_pass<NCH, U>, 256, 0) != hipSuccess || occ < 1) occ = 2;
        if (occ > 4) occ = 4;
        *nblocks_io = c->num_cu * occ;
        return;
    }
    RAILS_LAUNCH((k_lanczos_pass<NCH, U>), dim3(*nblocks_io), dim3(256), 0, c->stream, a);
}

int lz_unroll()
{
    static int u = -1;
    if (u < 0) {
        const char *e = getenv("RAILS_LZ_UNROLL");
        u = e ? atoi(e) : 4;
        if (u != 1 && u != 2 && u != 4) u = 4;
    }
    return u;
}

void launch_pass(rails_ctx *c, const LzArgs &a, int nch, int *nblocks_io, bool size_only)
{
    int u = lz_unroll();
    if (nch >= 3 && u > 2) u = 2;
#define RAILS_LZ_CASE(N, UU) \
    if (nch == N && u == UU) return launch_pass_u<N, UU>(c, a, nblocks_io, size_only);
    RAILS_LZ_CASE(1, 1) RAILS_LZ_CASE(1, 2) RAILS_LZ_CASE(1, 4) RAILS_LZ_CASE(2, 1) RAILS_LZ_CASE(2, 2) RAILS_LZ_CASE(2, 4)
    RAILS_LZ_CASE(3, 1) RAILS_LZ_CASE(3, 2) RAILS_LZ_CASE(4, 1) RAILS_LZ_CASE(4, 2)
#undef RAILS_LZ_CASE
}

} // namespace

static int lz_run(rails_ctx *c, const rails_panel *AV, int avc0, const rails_panel *MV, int mvc0, int k, const double *T_host, int ldt,
                  const rails_panel *B, int bc0, int p, int L, double *H_host, int ldh, int *steps_out, double *start_sums_host)
{
    const bool only_start = (start_sums_host != nullptr);
    RAILS_REQUIRE(c && AV && MV && B && (only_start || (H_host && steps_out)), "rails_resid_lanczos: null argument");
    RAILS_REQUIRE(k >= 0 && p >= 0 && L >= 1 && (only_start || ldh >= L + 1), "rails_resid_lanczos: bad sizes k=%d p=%d L=%d ldh=%d", k, p, L, ldh);
    RAILS_REQUIRE(avc0 >= 0 && avc0 + k <= AV->cap && mvc0 >= 0 && mvc0 + k <= MV->cap && bc0 >= 0 && bc0 + p <= B->cap,
                  "rails_resid_lanczos: column windows outside capacity");
    RAILS_REQUIRE(AV->m == MV->m && AV->m == B->m, "rails_resid_lanczos: row mismatch");
    RAILS_REQUIRE(((avc0 | mvc0 | bc0) & 1) == 0, "rails_resid_lanczos: column windows must start at even columns");
    RAILS_REQUIRE(k <= 512 && p <= 128, "rails_resid_lanczos: fused kernel supports k <= 512, p <= 128 (got %d, %d)", k, p);
    RAILS_REQUIRE(k == 0 || only_start || (T_host && ldt >= k), "rails_resid_lanczos: bad T");
    const int64_t m = AV->m;
    const int64_t mpad = (m + 63) / 64 * 64;
    rails_lanczos_state &S = lz_state(c);
    // Lanczos vectors
    size_t qbytes = (size_t)(L + 2) * (size_t)std::max<int64_t>(mpad, 64) * sizeof(double);
    if (qbytes > S.qc_bytes) {
        RAILS_HIP_CHECK(rails_stream_sync(c));
        if (S.Qc) RAILS_HIP_CHECK(hipFree(S.Qc));
        S.Qc = nullptr;
        S.qc_bytes = 0;
        hipError_t e = hipMalloc((void **)&S.Qc, qbytes);
        if (e != hipSuccess) {
            rails_set_error("rails_resid_lanczos: hipMalloc(%zu) failed: %s", qbytes, hipGetErrorString(e));
            return RAILS_ENOMEM;
        }
        S.qc_bytes = qbytes;
    }
    S.m = m;
    S.mpad = std::max<int64_t>(mpad, 64);
    S.L = L;
    S.steps = 0;
    const int ncoef = 2 * k + p + 1;
    int64_t ngroups = S.mpad / 64;
    const int nch = std::min(4, std::max(1, (k + 127) / 128));
    int nblocks = 0;
    {
        LzArgs dummy;
        launch_pass(c, dummy, nch, &nblocks, true);
    }
    nblocks = (int)std::min<int64_t>((ngroups + 3) / 4, (int64_t)nblocks);
    if (nblocks < 1) nblocks = 1;
    // small device block: T | coef | sums | state(8) | alphas(L+2) | betas(L+2)
    size_t nsmall = (size_t)k * k + ncoef + ncoef + 8 + 2 * (size_t)(L + 2);
    if (nsmall * sizeof(double) > S.small_bytes) {
        RAILS_HIP_CHECK(rails_stream_sync(c));
        if (S.small) RAILS_HIP_CHECK(hipFree(S.small));
        S.small = nullptr;
        RAILS_HIP_CHECK(hipMalloc((void **)&S.small, nsmall * sizeof(double) * 2));
        S.small_bytes = nsmall * sizeof(double) * 2;
    }
    double *dT = S.small;
    double *dcoef = dT + (size_t)k * k;
    double *dsums = dcoef + ncoef;
    double *dstate = dsums + ncoef;
    double *dalpha = dstate + 8;
    double *dbeta = dalpha + (L + 2);
    RAILS_TRY(rails_ws_reserve(c, (size_t)nblocks * ncoef * sizeof(double)));
    RAILS_TRY(rails_pinned_begin_write(c, std::max<size_t>((size_t)k * k, (size_t)(2 * (L + 2) + 8)) * sizeof(double)));
    // T -> device (contiguous k x k)
    if (!only_start) {
        for (int j = 0; j < k; ++j) memcpy(c->pinned + (size_t)j * k, T_host + (size_t)j * ldt, sizeof(double) * k);
        if (k) RAILS_HIP_CHECK(hipMemcpyAsync(dT, c->pinned, (size_t)k * k * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    RAILS_HIP_CHECK(hipMemsetAsync(dcoef, 0, (nsmall - (size_t)k * k) * sizeof(double), c->stream));
    // start vector: Q.random() consumes one RNG stream (src/LyapunovSolver.hpp:374)
    {
        uint64_t stream = c->next_stream++;
        int grid = (int)std::min<int64_t>((S.mpad + 255) / 256, (int64_t)c->num_cu * 8);
        RAILS_LAUNCH(k_lz_random, dim3(grid), dim3(256), 0, c->stream, S.Qc, m, S.mpad, c->seed, stream, c->row0);
    }
    LzArgs a;
    a.AV = AV->d + avc0;
    a.ldav = AV->ld;
    a.MV = MV->d + mvc0;
    a.ldmv = MV->ld;
    a.B = B->d + bc0;
    a.ldb = B->ld;
    a.k = k;
    a.p = p;
    a.av_room = AV->ld - avc0;
    a.mv_room = MV->ld - mvc0;
    a.b_room = B->ld - bc0;
    a.m = m;
    a.mpad = S.mpad;
    a.Qc = S.Qc;
    a.coef = dcoef;
    a.state = dstate;
    a.partial = c->ws;
    for (int step = -1; step < (only_start ? 0 : L); ++step) {
        a.step = step;
        launch_pass(c, a, nch, &nblocks, false);
        RAILS_LAUNCH(k_lz_reduce, dim3((ncoef + 63) / 64), dim3(1024), 0, c->stream, c->ws, nblocks, ncoef, dsums);
        RAILS_TRY(rails_allreduce_dev(c, dsums, (size_t)ncoef));
        if (only_start) break;
        RAILS_LAUNCH(k_lz_small, dim3(1), dim3(1024), 0, c->stream, dsums, dT, k, p, step, dcoef, dstate, dalpha, dbeta);
    }
    RAILS_HIP_CHECK(hipGetLastError());
    if (only_start) { // [AV^T q0 | MV^T q0 | B^T q0 | q0^T q0], q0 kept (raw) as Lanczos vector 0
        RAILS_TRY(rails_pinned_reserve(c, (size_t)ncoef * sizeof(double)));
        RAILS_HIP_CHECK(hipMemcpyAsync(c->pinned, dsums, (size_t)ncoef * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RAILS_HIP_CHECK(rails_stream_sync(c));
        memcpy(start_sums_host, c->pinned, (size_t)ncoef * sizeof(double));
        S.steps = 1;
        c->n_lanczos_start++;
        return RAILS_OK;
    }
    // read back state, alphas, betas (contiguous)
    size_t nback = 8 + 2 * (size_t)(L + 2);
    RAILS_HIP_CHECK(hipMemcpyAsync(c->pinned, dstate, nback * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RAILS_HIP_CHECK(rails_stream_sync(c));
    const double *hstate = c->pinned, *halpha = c->pinned + 8, *hbeta = halpha + (L + 2);
    const bool broke = hstate[3] != 0.0;
    int steps = broke ? (int)hstate[4] : L;
    for (int j = 0; j <= L; ++j)
        for (int i = 0; i <= L; ++i) H_host[i + (size_t)j * ldh] = 0.0; // H = 0.0 (:377)
    for (int i = 0; i < steps; ++i) {
        H_host[i + (size_t)i * ldh] = halpha[i]; // :407
        bool last_broke = broke && (i == steps - 1);
        if (!last_broke) { // :428-429
            H_host[(i + 1) + (size_t)i * ldh] = hbeta[i];
            H_host[i + (size_t)(i + 1) * ldh] = hbeta[i];
        }
    }
    S.steps = steps;
    c->n_lanczos++;
    *steps_out = steps;
    return RAILS_OK;
}

extern "C" int rails_resid_lanczos(rails_ctx *c, const rails_panel *AV, int avc0, const rails_panel *MV, int mvc0, int k,
                                   const double *T_host, int ldt, const rails_panel *B, int bc0, int p, int L, double *H_host,
                                   int ldh, int *steps_out)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    return lz_run(c, AV, avc0, MV, mvc0, k, T_host, ldt, B, bc0, p, L, H_host, ldh, steps_out, nullptr);
}

extern "C" int rails_lanczos_start(rails_ctx *c, const rails_panel *AV, int avc0, const rails_panel *MV, int mvc0, int k,
                                   const rails_panel *B, int bc0, int p, double *sums_host)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    RAILS_REQUIRE(sums_host, "rails_lanczos_start: null output");
    return lz_run(c, AV, avc0, MV, mvc0, k, nullptr, 0, B, bc0, p, 1, nullptr, 0, nullptr, sums_host);
}

extern "C" int rails_lanczos_vectors(rails_ctx *c, const double *S_host, int lds, int w, rails_panel *Out, int oc0)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    RAILS_REQUIRE(c && Out, "rails_lanczos_vectors: null argument");
    rails_lanczos_state &S = lz_state(c);
    RAILS_REQUIRE(S.Qc && S.steps > 0, "rails_lanczos_vectors: no Lanczos run to take vectors from");
    RAILS_REQUIRE(w >= 0 && oc0 >= 0 && oc0 + w <= Out->cap, "rails_lanczos_vectors: columns [%d,%d) outside capacity %d", oc0, oc0 + w,
                  Out->cap);
    RAILS_REQUIRE(Out->m == S.m, "rails_lanczos_vectors: row mismatch %lld vs %lld", (long long)Out->m, (long long)S.m);
    RAILS_REQUIRE(w == 0 || (S_host && lds >= S.steps), "rails_lanczos_vectors: bad coefficient matrix");
    if (w == 0 || S.m == 0) return RAILS_OK;
    const int steps = S.steps;
    size_t n = (size_t)steps * w;
    RAILS_TRY(rails_small_reserve(c, n * sizeof(double)));
    RAILS_TRY(rails_pinned_begin_write(c, n * sizeof(double)));
    for (int j = 0; j < w; ++j) memcpy(c->pinned + (size_t)j * steps, S_host + (size_t)j * lds, sizeof(double) * steps);
    RAILS_HIP_CHECK(hipMemcpyAsync(c->small, c->pinned, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    for (int j0 = 0; j0 < w; j0 += 16) {
        int wc = std::min(16, w - j0);
        RAILS_LAUNCH(k_lz_vectors, dim3((unsigned)((S.m + 255) / 256)), dim3(256), (size_t)steps * 16 * sizeof(double), c->stream,
                           S.Qc, S.mpad, S.m, steps, c->small + (size_t)j0 * steps, steps, wc, Out->d + oc0 + j0, Out->ld);
    }
    RAILS_HIP_CHECK(hipGetLastError());
    RAILS_HIP_CHECK(rails_stream_sync(c));
    return RAILS_OK;
}

extern "C" int rails_lanczos_release(rails_ctx *c)
{
    if (!c || !c->lz) return RAILS_OK;
    rails_lanczos_state *S = static_cast<rails_lanczos_state *>(c->lz);
    rails_stream_sync(c);
    if (S->Qc) hipFree(S->Qc);
    if (S->small) hipFree(S->small);
    delete S;
    c->lz = nullptr;
    return RAILS_OK;
}
