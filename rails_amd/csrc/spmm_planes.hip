// spmm_planes.hip -- Y = A X for structured-grid stencils (7- and 27-point patterns in natural ordering) on gfx950.
//
// Replaces `A_ * W` of the reference (src/LyapunovSolver.hpp:146; Epetra_CrsMatrix::Apply behind src/Epetra_OperatorWrapper.cpp:87) for
// the operators of BASELINE configs[1] and configs[3].  HBM-bound fp64 streaming work: no MFMA here.
//
// The box kernel (spmm.hip, k_spmm_tiled_reg) stages the halo box of a 4 x 4 x 4 tile per column chunk: every X row crosses the CUs'
// load path 3.4 times.  This kernel turns the sweep kernel's idea (spmm_sweep.hip: X streams once, the partial sums stay put) onto
// the grid: a workgroup owns a PX x PY patch of grid columns and walks along z.  Per step ONE plane of X (patch + halo, whole panel
// rows of up to 128 columns = 1 KiB) arrives in LDS by LDS-DMA, two planes ahead of its use, and contributes to the three output planes
// z-1, z, z+1, whose partial sums live in registers: an X row is loaded (PX+2)(PY+2) / (PX PY) = 1.9 times instead of 3.4, every LDS read
// feeds up to six multiply-adds, and A is read once -- as a stream of per-(grid point, X plane) coefficient records that the waves
// fetch with SCALAR loads: a wave owns RW x-consecutive rows at all columns (lane = two columns), so a coefficient is wave-uniform and
// sits in SGPRs: no broadcasts, no per-lane value registers, no schedule.
//
// Result: every row's nonzeros meet the same chain of fused multiply-adds in column order as in the row-gather kernel (planes z-1, z,
// z+1, inside a plane y-1, y, y+1, inside a line x-1, x, x+1), so the product is bitwise the row-gather kernel's except for the sign
// of an exact zero (products with halo rows outside the grid are +0 terms).
#include "rails_internal.h"

#include <algorithm>

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int PL_REC = 28; // doubles per coefficient record (27 used; 224 B: whole 16-byte pieces for the LDS-DMA, pairs for the reads)

struct PlanesArgs {
    int gx, gy, gz;
    int ldx, ldy, nc;
    int npx, npy, nseg, seglen;
    int z_lo, z_hi; // output planes of this launch (the whole grid, or the interior planes of a z-slab whose end planes have ghost columns)
    int wg_per_xcd, nwg;
    int y_vec; // 1: Y's rows are 16-byte aligned (one store per lane and row), 0: a window on an odd column (two 8-byte stores)
};

// LPR lanes own a row (two columns each): 64 / LPR rows per wave-instruction.  Narrow panels (the in-loop A * W at Expand size 16 or 32,
// 64-column blocks) put G = 64 / LPR y-lines of the patch side by side in a wave, so that no lane idles: group g of wave (wx, wy) has the
// RW rows x = x0 + wx RW + r of line y = y0 + wy G + g.
template <int LPR, int RW, int WX, int WY, bool CROSS>
__global__ __launch_bounds__(64 * WX * WY) void k_spmm_planes(const double *__restrict__ X, double *__restrict__ Y,
                                                              const double *__restrict__ cf /* [gx*gy*gz][PL_REC]: coefficients by (grid column (x, y), X plane z): 9 of output plane z+1, 9 of z, 9 of z-1 */,
                                                              const double *__restrict__ zero /* 1 KiB of zeros: source of halo rows outside the grid */, PlanesArgs a)
{
    constexpr int G = 64 / LPR, NW = WX * WY, PX = RW * WX, PY = WY * G, HX = PX + 2, HY = PY + 2;
    constexpr int RB = LPR * 16;                    // bytes of a staged X row
    constexpr int HXP = (HX + G - 1) / G * G;       // rows of a halo line in LDS: whole LDS-DMA pieces (G rows = 1 KiB each)
    constexpr int PPL = HXP / G, NXS = HY * PPL;    // pieces per line, per plane
    constexpr int LINE_B = PX * PL_REC * 8;          // bytes of coefficient records per patch line
    constexpr int NCF = (PY * LINE_B + 1023) / 1024; // LDS-DMA instructions per plane for the patch's records
    constexpr int NS = NXS + NCF;                    // requests per plane: slot i goes to wave i % NW
    constexpr int NPW = (NS + NW - 1) / NW;
    constexpr int XBUF = NXS * 1024, CBUF = NCF * 1024; // X planes: 2 buffers; coefficient records: 3 (read during the whole step)
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, gl = lane / LPR, ll = lane % LPR;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wx = w % WX, wy = w / WX;
    const int b = blockIdx.x;
    const int id = a.wg_per_xcd > 0 ? (b & 7) * a.wg_per_xcd + (b >> 3) : b; // XCD x walks a contiguous run of patches (speed only)
    if (id >= a.nwg) return;
    const int px = id % a.npx, t = id / a.npx, py = t % a.npy, seg = t / a.npy;
    const int x0 = px * PX, y0 = py * PY;
    const int zs = a.z_lo + seg * a.seglen, ze = min(a.z_hi, zs + a.seglen);
    const int col0 = blockIdx.y * (2 * LPR), ncc = min(2 * LPR, a.nc - col0);
    const bool lane_on = 2 * ll < ncc;
    const int64_t plane_x = (int64_t)a.gx * a.gy * a.ldx;  // doubles per plane of X
    const int64_t plane_c = (int64_t)a.gx * a.gy * PL_REC; // ... of the records

    // The requests of a plane: NXS pieces of its halo rows of X (piece i: the G rows hx = (i % PPL) G + g of halo line i / PPL, one per
    // lane group; a lane's source is its two columns of its row, or the zero page for rows outside the grid; lanes beyond the panel's
    // width re-read the first two columns, into LDS words nobody stores from) and NCF pieces of 1 KiB of the patch's coefficient records
    // (PY lines of PX records, laid out line after line: per-lane offsets from the plane's first record, clamped into the grid -- what
    // lies beyond belongs to rows that are never stored).  All inline assembly: the compiler does not count LDS-DMA (it would wait
    // vmcnt(0) before every LDS read after one), the waits below do.
    int soff[NPW]; // X pieces: this lane's offset within a plane (doubles), -1 = outside the grid; coefficient pieces: its byte offset from the plane's first record
    int my_requests = 0;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int i = w + NW * k;
        soff[k] = -1;
        if (i < NXS) {
            const int hx = (i % PPL) * G + gl, hy = i / PPL;
            const int x = x0 - 1 + hx, y = y0 - 1 + hy;
            soff[k] = (hx < HX && x >= 0 && x < a.gx && y >= 0 && y < a.gy) ? (y * a.gx + x) * a.ldx + (lane_on ? 2 * ll : 0) : -1;
            ++my_requests;
        } else if (i < NS) {
            const int o = (i - NXS) * 1024 + lane * 16; // byte within the patch's block of records
            int ly = o / LINE_B, lo = o % LINE_B;
            int y = y0 + ly;
            y = y < a.gy ? y : a.gy - 1;
            const int avail = (a.gx - x0) * PL_REC * 8; // bytes of this line that exist
            lo = lo < avail ? lo : avail - 16;
            soff[k] = (y * a.gx + x0) * PL_REC * 8 + lo;
            ++my_requests;
        }
    }
    my_requests = __builtin_amdgcn_readfirstlane(my_requests);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const double *Xc = X + col0;
    auto stage = [&](int zp, int s) { // plane zp is consumed in step s: X buffer s % 2, record buffer s % 3
        const bool plane_ok = zp >= 0 && zp < a.gz && zp <= ze; // (planes beyond the segment's last halo plane: not needed)
        const int zc = zp < 0 ? 0 : (zp < a.gz ? zp : a.gz - 1);
        const double *Xz = Xc + (int64_t)zc * plane_x;
        const double *Cz = cf + (int64_t)zc * plane_c;
        const uint32_t xdst = lds_base + (uint32_t)((s & 1) * XBUF), cdst = lds_base + (uint32_t)(2 * XBUF + (s % 3) * CBUF);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int i = w + NW * k;
            if (i >= NS) continue;
            uint32_t keep;
            if (i >= NXS) {
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"((uint32_t)soff[k]), "s"(Cz), "s"(cdst + (uint32_t)((i - NXS) * 1024)) : "memory");
            } else {
                const double *src = (plane_ok && soff[k] >= 0) ? Xz + soff[k] : zero + 2 * lane;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(xdst + (uint32_t)(i * 1024)) : "memory");
            }
        }
    };

    // this lane's rows: x = xw + r (r < RW) at y = yw
    const int xw = x0 + wx * RW, yw = y0 + wy * G + gl;
    const bool row_line_ok = yw < a.gy && lane_on;
    double2_t acc[3][RW], out[RW];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[j][r] = (double2_t){0.0, 0.0};
#pragma unroll
    for (int r = 0; r < RW; ++r) out[r] = (double2_t){0.0, 0.0};
    const double *xs_base = lds + ((size_t)((wy * G + gl) * HXP + wx * RW)) * (RB / 8) + 2 * ll;
    const double *cs_base = lds + (size_t)(2 * XBUF) / 8 + (size_t)((wy * G + gl) * PX + wx * RW) * PL_REC;
    auto store_out = [&](int zo) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (xw + r < a.gx && row_line_ok) {
                double *dst = Y + ((int64_t)(zo * a.gy + yw) * a.gx + xw + r) * a.ldy + col0 + 2 * ll;
                if (a.y_vec)
                    *reinterpret_cast<double2_t *>(dst) = out[r];
                else {
                    dst[0] = out[r].x;
                    dst[1] = out[r].y;
                }
            }
    };

    stage(zs - 1, 0);
    stage(zs, 1);
    const int nsteps = ze - zs + 2;
    for (int s = 0; s < nsteps; ++s) {
        const int zp = zs - 1 + s;
        // plane zp has landed once all but this wave's youngest requests (those of plane zp + 1) are done; then everybody's has
        if (my_requests == NPW)
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NPW - 1) : "memory");
        __builtin_amdgcn_s_barrier();
        double2_t xs[3][RW + 2];
        const double *xb = xs_base + (size_t)(s & 1) * (XBUF / 8);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int cx = 0; cx < RW + 2; ++cx) xs[dy][cx] = *reinterpret_cast<const double2_t *>(xb + (dy * HXP + cx) * (RB / 8));
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        __builtin_amdgcn_s_barrier(); // every wave has its rows of this X buffer in registers: the buffer is free for plane zp + 2
        // the plane completed one step ago goes out first, so that the wait above counts LDS-DMA requests only
        if (s >= 3) store_out(zp - 2);
        stage(zp + 2, s + 2);
        if (zp >= 0 && zp < a.gz) {
            const double *cb = cs_base + (size_t)(s % 3) * (CBUF / 8);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                // the row's 27 coefficients for this X plane, the same words for every lane of the row: LDS broadcasts, two per read
                double2_t c2[PL_REC / 2];
#pragma unroll
                for (int q = 0; q < (CROSS ? 12 : 14); ++q) c2[q] = *reinterpret_cast<const double2_t *>(cb + r * PL_REC + 2 * q);
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            if (CROSS && ((j != 1 && (dy != 1 || dx != 1)) || (j == 1 && dy != 1 && dx != 1))) continue;
                            const int e = j * 9 + dy * 3 + dx;
                            const double cv = (e & 1) ? c2[e / 2].y : c2[e / 2].x;
                            acc[j][r].x = __builtin_fma(cv, xs[dy][r + dx].x, acc[j][r].x);
                            acc[j][r].y = __builtin_fma(cv, xs[dy][r + dx].y, acc[j][r].y);
                        }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            out[r] = acc[2][r];
            acc[2][r] = acc[1][r];
            acc[1][r] = acc[0][r];
            acc[0][r] = (double2_t){0.0, 0.0};
        }
    }
    store_out(ze - 1); // completed in the last step
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); // the two planes requested beyond the segment land before the LDS is given back
}

} // namespace

struct rails_planes_plan {
    bool ok = false;
    bool cross = false;
    bool last_interior = false; // the last product ran the interior planes only (row-partitioned operator, rails_spmm_planes_interior)
    int gx = 0, gy = 0, gz = 0;
    int z_lo = 0, z_hi = 0; // output planes the records are complete for: all of them, or a z-slab's planes without ghost columns
    double *cf = nullptr;
    double *zero = nullptr;
};

void rails_planes_release(rails_csr *A)
{
    if (!A->planes) return;
    if (A->planes->cf) hipFree(A->planes->cf);
    if (A->planes->zero) hipFree(A->planes->zero);
    delete A->planes;
    A->planes = nullptr;
}

// The coefficient records of a complete 7- or 27-point stencil, or nothing: every entry of every row has to be a neighbour of the
// row's grid point (no wrap-around), no neighbour inside the grid may be missing (a zero coefficient in its place would multiply an X
// row the matrix does not reference -- a non-finite value there must not leak into the product), no column twice.  One thread per row
// on the device (the CSR arrays are there already): 0.2 ms per million rows, so the plan costs a product or two and every operator
// that qualifies can have it from its first product on.
namespace {
__global__ __launch_bounds__(256) void k_planes_build(int64_t m, int gx, int gy, int gz, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      const double *__restrict__ val, double *__restrict__ cf, unsigned *__restrict__ flags, int64_t v_lo, int64_t v_hi)
{
    // rows [v_lo, v_hi) must be complete stencil rows; the others (the end planes of a z-slab: they have ghost columns and are not
    // computed from the records) only contribute the entries that multiply local X rows
    const int64_t plane = (int64_t)gx * gy;
    unsigned f = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(r % gx), y = (int)((r / gx) % gy), z = (int)(r / plane);
        const int ex = 1 + (x > 0) + (x < gx - 1), ey = 1 + (y > 0) + (y < gy - 1), ez = 1 + (z > 0) + (z < gz - 1);
        const int64_t p0 = rowptr[r], p1 = rowptr[r + 1];
        const bool checked = r >= v_lo && r < v_hi;
        if (checked && p1 - p0 != ex * ey * ez) f |= 4u;     // not the complete 27-point neighbourhood
        if (checked && p1 - p0 != ex + ey + ez - 2) f |= 8u; // not the complete 7-point one
        int64_t prev = -1;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t cc = col[p];
            if (checked && (cc <= prev || cc >= m)) f |= 1u; // (sorted, no duplicates: with the count above every neighbour is there exactly once)
            prev = cc;
            if (cc >= m) continue;
            const int dx = (int)(cc % gx) - x, dy = (int)((cc / gx) % gy) - y, dz = (int)(cc / plane) - z;
            if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) {
                if (checked) f |= 1u;
                continue;
            }
            if (checked && (dx != 0) + (dy != 0) + (dz != 0) > 1) f |= 2u;
            // the entry multiplies X row (x+dx, y+dy, z+dz): record of grid column (x, y) at X plane z + dz, output plane slot dz + 1
            cf[((int64_t)(z + dz) * plane + (int64_t)y * gx + x) * PL_REC + (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)] = val[p];
        }
    }
    if (f) atomicOr(flags, f);
}
} // namespace

static int planes_build(rails_ctx *c, rails_csr *A)
{
    A->planes = new rails_planes_plan();
    rails_planes_plan *P = A->planes;
    if (A->rect || A->nnz == 0 || (A->n_ghost == 0 && A->m != A->ncols_ext)) return RAILS_OK;
    int64_t gx = 0, gy = 0, gz = 0;
    if (!rails_detect_grid(A, &gx, &gy, &gz)) return RAILS_OK;
    if (gx * gy * gz != A->m || gx > 0x7fff || gy > 0x7fff || gz > 0x3fffffff) return RAILS_OK;
    // a row block with ghost columns: a z-slab whose interior rows (rails_csr_set_halo) are whole planes
    int64_t v_lo = 0, v_hi = A->m;
    if (A->n_ghost > 0) {
        v_lo = A->int_lo;
        v_hi = A->int_hi;
        if (v_lo % (gx * gy) || v_hi % (gx * gy) || v_hi - v_lo < gx * gy) return RAILS_OK;
    }
    const size_t cf_bytes = (size_t)A->m * PL_REC * sizeof(double);
    unsigned *flags = nullptr;
    RAILS_HIP_CHECK(hipMalloc((void **)&P->cf, cf_bytes));
    RAILS_HIP_CHECK(hipMalloc((void **)&P->zero, 1024 + 64));
    c->n_dev_alloc += 2;
    flags = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(P->zero) + 1024);
    RAILS_HIP_CHECK(hipMemsetAsync(P->cf, 0, cf_bytes, c->stream));
    RAILS_HIP_CHECK(hipMemsetAsync(P->zero, 0, 1024 + 64, c->stream));
    const int grid = (int)std::min<int64_t>((A->m + 255) / 256, (int64_t)c->num_cu * 16);
    RAILS_LAUNCH(k_planes_build, dim3(grid), dim3(256), 0, c->stream, A->m, (int)gx, (int)gy, (int)gz, A->rowptr, A->col, A->val, P->cf, flags, v_lo, v_hi);
    unsigned f = 0;
    RAILS_HIP_CHECK(hipMemcpyAsync(&f, flags, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    RAILS_HIP_CHECK(rails_stream_sync(c));
    const bool cross = !(f & 2u);
    if ((f & 1u) || (cross ? (f & 8u) : (f & 4u))) { // not a complete stencil: the records are of no use
        hipFree(P->cf);
        P->cf = nullptr;
        return RAILS_OK;
    }
    P->gx = (int)gx;
    P->gy = (int)gy;
    P->gz = (int)gz;
    P->cross = cross;
    P->z_lo = (int)(v_lo / (gx * gy));
    P->z_hi = (int)(v_hi / (gx * gy));
    P->ok = true;
    return RAILS_OK;
}

static int planes_env(const char *name, int def)
{
    const char *e = getenv(name);
    return e ? atoi(e) : def;
}

template <int LPR, int RW, int WX, int WY>
static int planes_launch(rails_ctx *c, const rails_planes_plan *P, const double *X, int ldx, double *Y, int ldy, int nc, hipStream_t st)
{
    const int y_vec = ((((uintptr_t)Y) & 15) == 0 && ldy % 2 == 0) ? 1 : 0;
    constexpr int G = 64 / LPR, NW = WX * WY, PX = RW * WX, PY = WY * G, HXP = (PX + 2 + G - 1) / G * G, NXS = (PY + 2) * (HXP / G);
    constexpr int NCF = (PY * PX * PL_REC * 8 + 1023) / 1024;
    static_assert((2 * NXS + 3 * NCF) * 1024 <= 160 * 1024, "patch too large for the LDS");
    PlanesArgs a;
    a.gx = P->gx;
    a.gy = P->gy;
    a.gz = P->gz;
    a.ldx = ldx;
    a.ldy = ldy;
    a.nc = nc;
    a.y_vec = y_vec;
    a.npx = (P->gx + PX - 1) / PX;
    a.npy = (P->gy + PY - 1) / PY;
    const int nchunks = (nc + 2 * LPR - 1) / (2 * LPR);
    // z segments: the workgroups of a launch run in rounds of one per CU; a segment of len planes costs len + 2 steps (+ ~2 of start-up)
    static const int env_seg = planes_env("RAILS_PLANES_SEG", 0);
    const int nz = P->z_hi - P->z_lo;
    a.z_lo = P->z_lo;
    a.z_hi = P->z_hi;
    int best_len = nz;
    double best = 1e300;
    for (int nseg = 1; nseg <= std::min(nz, 256); ++nseg) {
        const int len = (nz + nseg - 1) / nseg, ns = (nz + len - 1) / len;
        const double wgs = (double)a.npx * a.npy * ns * nchunks;
        const double rounds = std::ceil(wgs / (double)c->num_cu);
        const double cost = rounds * (len + 4);
        if (cost < best - 1e-9) {
            best = cost;
            best_len = len;
        }
    }
    if (env_seg > 0) best_len = std::min(env_seg, nz);
    a.seglen = best_len;
    a.nseg = (nz + best_len - 1) / best_len;
    a.nwg = a.npx * a.npy * a.nseg;
    a.wg_per_xcd = a.nwg >= 64 ? (a.nwg + 7) / 8 : 0;
    const unsigned grid = a.wg_per_xcd ? (unsigned)a.wg_per_xcd * 8u : (unsigned)a.nwg;
    const size_t lds = (size_t)(2 * NXS + 3 * NCF) * 1024;
    if (P->cross) {
        RAILS_HIP_CHECK(hipFuncSetAttribute((const void *)k_spmm_planes<LPR, RW, WX, WY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (st == c->stream)
            RAILS_LAUNCH((k_spmm_planes<LPR, RW, WX, WY, true>), dim3(grid, (unsigned)nchunks), dim3(64 * NW), lds, c->stream, X, Y, P->cf, P->zero, a);
        else
            hipLaunchKernelGGL((k_spmm_planes<LPR, RW, WX, WY, true>), dim3(grid, (unsigned)nchunks), dim3(64 * NW), lds, st, X, Y, P->cf, P->zero, a);
    } else {
        RAILS_HIP_CHECK(hipFuncSetAttribute((const void *)k_spmm_planes<LPR, RW, WX, WY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (st == c->stream)
            RAILS_LAUNCH((k_spmm_planes<LPR, RW, WX, WY, false>), dim3(grid, (unsigned)nchunks), dim3(64 * NW), lds, c->stream, X, Y, P->cf, P->zero, a);
        else
            hipLaunchKernelGGL((k_spmm_planes<LPR, RW, WX, WY, false>), dim3(grid, (unsigned)nchunks), dim3(64 * NW), lds, st, X, Y, P->cf, P->zero, a);
    }
    return RAILS_OK;
}

static int planes_dispatch(rails_ctx *c, const rails_planes_plan *P, const double *X, int ldx, double *Y, int ldy, int nc, hipStream_t st)
{
    // lanes per row by the panel's width; patch shapes by what the LDS holds (two X planes + three planes of records)
    static const int shape = planes_env("RAILS_PLANES_SHAPE", 0);
    if (nc <= 16)
        RAILS_TRY((planes_launch<8, 2, 4, 2>(c, P, X, ldx, Y, ldy, nc, st)));
    else if (nc <= 32)
        RAILS_TRY((planes_launch<16, 2, 4, 3>(c, P, X, ldx, Y, ldy, nc, st)));
    else if (nc <= 64)
        RAILS_TRY((planes_launch<32, 2, 4, 4>(c, P, X, ldx, Y, ldy, nc, st)));
    else if (shape == 1)
        RAILS_TRY((planes_launch<64, 4, 2, 4>(c, P, X, ldx, Y, ldy, nc, st)));
    else if (shape == 2)
        RAILS_TRY((planes_launch<64, 3, 2, 5>(c, P, X, ldx, Y, ldy, nc, st)));
    else if (shape == 3)
        RAILS_TRY((planes_launch<32, 2, 4, 4>(c, P, X, ldx, Y, ldy, nc, st)));
    else
        RAILS_TRY((planes_launch<64, 2, 4, 4>(c, P, X, ldx, Y, ldy, nc, st)));
    return RAILS_OK;
}

static bool planes_shape_ok(const rails_planes_plan *P, int ldx, int ldy, int nc, bool aligned)
{
    if (!P->ok || !aligned || nc < 2 || (nc & 1)) return false;
    return (int64_t)P->gx * P->gy * std::max(std::max(ldx, ldy), PL_REC) * 8 < 0x7fffffffLL; // 32-bit offsets inside a plane
}

// *done tells whether the kernel computed the product; `build` = the plan may be made now (one pass over the matrix on the device)
int rails_spmm_planes(rails_ctx *c, rails_csr *A, const double *X, int ldx, double *Y, int ldy, int nc, bool aligned, bool build, bool *done)
{
    *done = false;
    if (A->n_ghost > 0) return RAILS_OK; // (row blocks with ghost columns: rails_spmm_planes_interior)
    if (!A->planes) {
        if (!build) return RAILS_OK;
        RAILS_TRY(planes_build(c, A));
    }
    rails_planes_plan *P = A->planes;
    if (!planes_shape_ok(P, ldx, ldy, nc, aligned)) return RAILS_OK;
    RAILS_TRY(planes_dispatch(c, P, X, ldx, Y, ldy, nc, c->stream));
    P->last_interior = false;
    A->last_kernel = "k_spmm_planes";
    *done = true;
    return RAILS_OK;
}

// The interior planes of a row block that is a z-slab of a grid stencil (BASELINE configs[3]'s partition): the planes without ghost
// columns, on stream st -- while the ghost planes travel (rails_spmm).  The two end planes are the caller's (row kernels, after the exchange).
int rails_spmm_planes_interior(rails_ctx *c, rails_csr *A, const double *X, int ldx, double *Y, int ldy, int nc, bool aligned, hipStream_t st, bool *done)
{
    *done = false;
    static const int on = planes_env("RAILS_SPMM_PLANES", 1);
    if (!on || A->n_ghost == 0 || A->variant != 0) return RAILS_OK;
    if (!A->planes) RAILS_TRY(planes_build(c, A));
    rails_planes_plan *P = A->planes;
    P->last_interior = false;
    if (!planes_shape_ok(P, ldx, ldy, nc, aligned)) return RAILS_OK;
    RAILS_TRY(planes_dispatch(c, P, X, ldx, Y, ldy, nc, st));
    P->last_interior = true;
    c->n_spmm_planes++;
    *done = true;
    return RAILS_OK;
}

bool planes_last_interior(const rails_csr *A) { return A->planes && A->planes->last_interior; }
