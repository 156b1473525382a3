// spmm_sweep.hip -- Y = A X for banded patterns on gfx950: the "sweep" kernel.
//
// Replaces `A_ * W` of the reference (src/LyapunovSolver.hpp:146) where the row-gather kernel of spmm.hip is bound by
// the gather itself: with a row-major panel every nonzero moves one 1-KiB row of X from L2 to a CU (27.6 GB at m = 1M,
// 27 nonzeros per row, 128 columns, against 2.4 GB of algorithmic traffic), and for |j - i| <= 4096 the window of rows
// a workgroup gathers from is far larger than LDS.  Here the roles are turned around:
//
//   * X streams ONCE per XCD.  The rows are cut into 8 parts (one per XCD); all 32 workgroups of a part sweep the same
//     X rows in the same order at the same pace, 256 rows per step, so a row is fetched from HBM by whichever workgroup
//     asks first and served to the others by that XCD's L2.
//   * a workgroup keeps a ring of 5 x 256 X rows, 16 columns (one 128-B line) wide, in LDS (160 KiB), filled by LDS-DMA
//     one step ahead of the lanes, and owns the 16 x R partial sums of one block of R rows in registers: the partial sums,
//     not the X rows, are what stays put while the window slides by.  8 column chunks x 4 row-block phases = 32
//     workgroups per part.
//   * the order in which the nonzeros meet the ring is fixed per matrix on the host (sweep_plan.cpp): per step and row
//     group a number of lock-step trips; in one trip each of a wave's 8 slots (8 lanes = one row) reads the 128-B ring row
//     of its next nonzero and adds value x row into its partial sums.  Trips come in units of four (four ring rows in
//     flight per lane, no per-trip control flow); (value, ring row) pairs arrive as a dense stream, two pairs per lane and
//     batch of 16 trips, and a slot's pair is handed to its 8 lanes with DPP shifts and quad broadcasts.
//
// Every row's nonzeros are consumed in ascending column order with one fused multiply-add each, exactly like the
// row-gather kernel: the two kernels give bit-identical results.
#include "rails_internal.h"
#include "sweep_plan.h"

#include <map>

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int SEG = 256;   // X rows per step
constexpr int NSEG = 5;    // ring segments
constexpr int RING_BYTES = SEG * NSEG * 128;

struct SweepArgs {
    int64_t ldx, ldg, ldy, m, ncols;
    int parts, n_chunks, phases;
    int ablate; // experiments only (RAILS_SWEEP_ABLATE): 1 = no LDS-DMA after the first step, 2 = no trips (results are wrong either way)
};

// quad broadcast of quad lane T; shifts by four lanes inside a row of 16 (= two slots) that only write one quad of each slot
#define RAILS_BCQ(x, T) __builtin_amdgcn_update_dpp(0, (int)(x), (T) * 0x55, 0xf, 0xf, true)
#define RAILS_LO_TO_HI(x) __builtin_amdgcn_update_dpp((int)(x), (int)(x), 0x114 /* row_shr:4 */, 0xf, 0xa, false)
#define RAILS_HI_TO_LO(x) __builtin_amdgcn_update_dpp((int)(x), (int)(x), 0x104 /* row_shl:4 */, 0xf, 0x5, false)

// The schedule pointers are separate __restrict__ kernel arguments: what is read through them is never written by the kernel.
template <int W, int G>
__global__ __launch_bounds__(W * 64) void k_spmm_sweep(SweepArgs a, const int64_t *__restrict__ part_row0, const int64_t *__restrict__ sweep0_,
                                                       const int32_t *__restrict__ nsteps_, const int64_t *__restrict__ hdr_off,
                                                       const int64_t *__restrict__ batch_off, const int64_t *__restrict__ flush_off,
                                                       const uint8_t *__restrict__ codes, const double *__restrict__ vals,
                                                       const uint16_t *__restrict__ offs, const int32_t *__restrict__ flush_rows,
                                                       const double *__restrict__ X, const double *__restrict__ Xg, double *__restrict__ Y)
{
    static_assert(G <= RAILS_SWEEP_CODES && RAILS_SWEEP_CODES == 64, "one code byte per lane");
    static_assert((SEG / 8) % W == 0, "every wave issues the same number of LDS-DMA instructions per step");
    __shared__ __attribute__((aligned(128))) unsigned char ring[RING_BYTES];
    const int part = (int)(blockIdx.x % (unsigned)a.parts);
    const int local = (int)(blockIdx.x / (unsigned)a.parts);
    const int chunk = local % a.n_chunks;
    const int phase = local / a.n_chunks;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int slot = lane >> 3, q = lane & 7;
    const int lane_off = q * 16;
    const int64_t prog = ((int64_t)part * a.phases + phase) * W + wave;
    const uint8_t *cp = codes + hdr_off[prog] + lane; // lane g reads the byte of group g of a step's record
    const int32_t *fp = flush_rows + flush_off[prog];
    const int64_t b0 = batch_off[prog];
    // a batch = 16 trips of the wave: lane q of a slot holds the slot's (value, ring row) of trips q and q + 8
    const double2_t *vp = reinterpret_cast<const double2_t *>(vals) + (b0 * 8 + slot) * 8 + q;
    const uint32_t *op = reinterpret_cast<const uint32_t *>(offs) + (b0 * 8 + slot) * 8 + q;
    const int64_t row_end = part_row0[part + 1];
    const int64_t sweep0 = sweep0_[part];
    const int nsteps = nsteps_[part];
    const int col0 = chunk * 16;

    // one step of X rows into ring segment `seg`: SEG / 8 LDS-DMA instructions of 8 rows x 128 B, dealt over the waves
    auto stage = [&](int k, int seg) {
        for (int j = wave; j < SEG / 8; j += W) {
            int64_t xr = sweep0 + (int64_t)k * SEG + j * 8 + (lane >> 3);
            xr = xr < 0 ? 0 : (xr >= a.ncols ? a.ncols - 1 : xr);
            const double *src = (xr < a.m ? X + xr * a.ldx : Xg + (xr - a.m) * a.ldg) + col0 + q * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(ring + (seg * SEG + j * 8) * 128), 16, 0, 0);
        }
    };

    double2_t acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (double2_t){0.0, 0.0};

    stage(0, 0);
    // Two batches, A and B, used in turn (four units each) and each requested while the other is in use, sixteen trips ahead;
    // they are never copied, so no wait sits right behind a load.
    double2_t Av = vp[0], Bv = (double2_t){0.0, 0.0};
    uint32_t Ao = op[0], Bo = 0;
    int64_t bnext = 1;
    int ub = 0, fl = 0, seg = 0;
    uint32_t code_v = cp[0];
    for (int k = 0; k < nsteps; ++k) {
        // step k's rows have landed for every wave, and every wave is done reading the segment refilled next
        __syncthreads();
        const uint32_t codes_k = code_v;
        const int seg_next = seg + 1 == NSEG ? 0 : seg + 1;
        if (k + 1 < nsteps) {
            code_v = cp[(int64_t)(k + 1) * RAILS_SWEEP_CODES];
            if (!(a.ablate & 1)) stage(k + 1, seg_next);
        }
        seg = seg_next;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)codes_k, g);
            for (int u = (a.ablate & 2) ? 0 : (int)(code & 0x7fu); u > 0; --u) {
                // the unit's four trips sit in one quad of every slot: hand them to the other quad
                int lo, hi, ro;
                switch (ub) {
                case 0:
                    lo = RAILS_LO_TO_HI(__double2loint(Av.x));
                    hi = RAILS_LO_TO_HI(__double2hiint(Av.x));
                    ro = RAILS_LO_TO_HI((Ao & 0xffffu) << 7);
                    break;
                case 1:
                    lo = RAILS_HI_TO_LO(__double2loint(Av.x));
                    hi = RAILS_HI_TO_LO(__double2hiint(Av.x));
                    ro = RAILS_HI_TO_LO((Ao & 0xffffu) << 7);
                    break;
                case 2:
                    lo = RAILS_LO_TO_HI(__double2loint(Av.y));
                    hi = RAILS_LO_TO_HI(__double2hiint(Av.y));
                    ro = RAILS_LO_TO_HI((Ao >> 16) << 7);
                    break;
                case 3:
                    lo = RAILS_HI_TO_LO(__double2loint(Av.y));
                    hi = RAILS_HI_TO_LO(__double2hiint(Av.y));
                    ro = RAILS_HI_TO_LO((Ao >> 16) << 7);
                    break;
                case 4:
                    lo = RAILS_LO_TO_HI(__double2loint(Bv.x));
                    hi = RAILS_LO_TO_HI(__double2hiint(Bv.x));
                    ro = RAILS_LO_TO_HI((Bo & 0xffffu) << 7);
                    break;
                case 5:
                    lo = RAILS_HI_TO_LO(__double2loint(Bv.x));
                    hi = RAILS_HI_TO_LO(__double2hiint(Bv.x));
                    ro = RAILS_HI_TO_LO((Bo & 0xffffu) << 7);
                    break;
                case 6:
                    lo = RAILS_LO_TO_HI(__double2loint(Bv.y));
                    hi = RAILS_LO_TO_HI(__double2hiint(Bv.y));
                    ro = RAILS_LO_TO_HI((Bo >> 16) << 7);
                    break;
                default:
                    lo = RAILS_HI_TO_LO(__double2loint(Bv.y));
                    hi = RAILS_HI_TO_LO(__double2hiint(Bv.y));
                    ro = RAILS_HI_TO_LO((Bo >> 16) << 7);
                    break;
                }
                // request the batch after next once the one in use has been touched (its wait then covers nothing younger)
                if (ub == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    Bv = vp[bnext * 64];
                    Bo = op[bnext * 64];
                    ++bnext;
                } else if (ub == 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    Av = vp[bnext * 64];
                    Ao = op[bnext * 64];
                    ++bnext;
                }
                ub = (ub + 1) & 7;
                // four ring rows in flight per lane, then four multiply-adds per partial sum in trip (= column) order: the same
                // chain of fused multiply-adds as the row-gather kernel
                const double2_t x0 = *reinterpret_cast<const double2_t *>(ring + (RAILS_BCQ(ro, 0) + lane_off));
                const double2_t x1 = *reinterpret_cast<const double2_t *>(ring + (RAILS_BCQ(ro, 1) + lane_off));
                const double2_t x2 = *reinterpret_cast<const double2_t *>(ring + (RAILS_BCQ(ro, 2) + lane_off));
                const double2_t x3 = *reinterpret_cast<const double2_t *>(ring + (RAILS_BCQ(ro, 3) + lane_off));
                const double v0 = __hiloint2double(RAILS_BCQ(hi, 0), RAILS_BCQ(lo, 0));
                const double v1 = __hiloint2double(RAILS_BCQ(hi, 1), RAILS_BCQ(lo, 1));
                const double v2 = __hiloint2double(RAILS_BCQ(hi, 2), RAILS_BCQ(lo, 2));
                const double v3 = __hiloint2double(RAILS_BCQ(hi, 3), RAILS_BCQ(lo, 3));
                acc[g].x = __builtin_fma(v0, x0.x, acc[g].x);
                acc[g].y = __builtin_fma(v0, x0.y, acc[g].y);
                acc[g].x = __builtin_fma(v1, x1.x, acc[g].x);
                acc[g].y = __builtin_fma(v1, x1.y, acc[g].y);
                acc[g].x = __builtin_fma(v2, x2.x, acc[g].x);
                acc[g].y = __builtin_fma(v2, x2.y, acc[g].y);
                acc[g].x = __builtin_fma(v3, x3.x, acc[g].x);
                acc[g].y = __builtin_fma(v3, x3.y, acc[g].y);
            }
            if (code & 0x80u) {
                const int64_t row = (int64_t)__builtin_amdgcn_readfirstlane(fp[fl++]) + slot;
                if (row < row_end) *reinterpret_cast<double2_t *>(Y + row * a.ldy + col0 + q * 2) = acc[g];
                acc[g] = (double2_t){0.0, 0.0};
            }
        }
    }
}

struct DevPlan {
    rails_sweep_plan host; // kept for its small arrays and statistics (the big arrays are released after the upload)
    int64_t *part_row0 = nullptr, *sweep0 = nullptr, *hdr_off = nullptr, *batch_off = nullptr, *flush_off = nullptr;
    int32_t *nsteps = nullptr, *flush_rows = nullptr;
    uint8_t *codes = nullptr;
    double *vals = nullptr;
    uint16_t *offs = nullptr;
    bool ok = false;
};

template <typename T>
int up(rails_ctx *c, T **dst, const std::vector<T> &src)
{
    RAILS_HIP_CHECK(hipMalloc((void **)dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    c->n_dev_alloc++;
    if (!src.empty()) RAILS_HIP_CHECK(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return RAILS_OK;
}

void free_plan(DevPlan *d)
{
    if (!d) return;
    hipFree(d->part_row0);
    hipFree(d->sweep0);
    hipFree(d->hdr_off);
    hipFree(d->batch_off);
    hipFree(d->flush_off);
    hipFree(d->nsteps);
    hipFree(d->flush_rows);
    hipFree(d->codes);
    hipFree(d->vals);
    hipFree(d->offs);
    delete d;
}

} // namespace

// plans of an operator, one per number of column chunks (kept in rails_csr::sweep_plans as an opaque pointer)
struct rails_sweep_cache {
    std::map<int, DevPlan *> by_chunks;
};

void rails_sweep_release(rails_csr *A)
{
    if (!A->sweep) return;
    for (auto &kv : A->sweep->by_chunks) free_plan(kv.second);
    delete A->sweep;
    A->sweep = nullptr;
}

// the geometry the kernel is instantiated for
static constexpr int SWEEP_W = 8, SWEEP_G = 44;

// *done = true when the product was computed here.  force: fail instead of declining.
int rails_spmm_sweep(rails_ctx *c, rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, bool aligned,
                     bool force, bool *done)
{
    *done = false;
    const int n_chunks = nc / 16;
    const bool shape_ok = aligned && nc % 16 == 0 && n_chunks >= 1 && n_chunks <= 32 && 32 % n_chunks == 0 && c->num_cu >= 256 && A->n_ghost == 0 &&
                          A->ncols_ext < 0x7fffffffLL;
    if (!shape_ok) {
        RAILS_REQUIRE(!force, "rails_spmm: the sweep kernel needs a multiple of 16 columns that divides 512, even column offsets, 256 CUs and no ghost rows");
        return RAILS_OK;
    }
    if (!A->sweep) A->sweep = new rails_sweep_cache();
    DevPlan *&d = A->sweep->by_chunks[n_chunks];
    if (!d) {
        d = new DevPlan();
        rails_sweep_params prm;
        prm.waves = SWEEP_W;
        prm.groups = SWEEP_G;
        prm.seg_rows = SEG;
        prm.nseg = NSEG;
        prm.parts = 8;
        prm.phases = 32 / n_chunks;
        if (rails_sweep_plan_build(prm, A->m, A->ncols_ext, A->h_rowptr.data(), A->h_col.data(), A->h_val.data(), d->host)) {
            RAILS_TRY(up(c, &d->part_row0, d->host.part_row0));
            RAILS_TRY(up(c, &d->sweep0, d->host.sweep0));
            RAILS_TRY(up(c, &d->nsteps, d->host.nsteps));
            RAILS_TRY(up(c, &d->hdr_off, d->host.hdr_off));
            RAILS_TRY(up(c, &d->batch_off, d->host.batch_off));
            RAILS_TRY(up(c, &d->flush_off, d->host.flush_off));
            RAILS_TRY(up(c, &d->codes, d->host.codes));
            RAILS_TRY(up(c, &d->vals, d->host.vals));
            RAILS_TRY(up(c, &d->offs, d->host.offs));
            RAILS_TRY(up(c, &d->flush_rows, d->host.flush_rows));
            RAILS_HIP_CHECK(hipStreamSynchronize(c->stream));
            std::vector<double>().swap(d->host.vals);
            std::vector<uint16_t>().swap(d->host.offs);
            std::vector<uint8_t>().swap(d->host.codes);
            d->ok = true;
        }
    }
    if (!d->ok) {
        RAILS_REQUIRE(!force, "rails_spmm: the sweep kernel does not fit this operator: %s", d->host.why.c_str());
        return RAILS_OK;
    }
    // worth it only where the lock-step trips are reasonably full and a row block re-uses what it stages
    if (!force && (d->host.efficiency < 0.5 || d->host.staged_rows_per_row > 8.0)) return RAILS_OK;
    SweepArgs a;
    a.ldx = ldx;
    a.ldg = ldg;
    a.ldy = ldy;
    a.m = A->m;
    a.ncols = A->ncols_ext;
    a.parts = 8;
    a.n_chunks = n_chunks;
    a.phases = 32 / n_chunks;
    static const int ablate = getenv("RAILS_SWEEP_ABLATE") ? atoi(getenv("RAILS_SWEEP_ABLATE")) : 0;
    a.ablate = ablate;
    hipLaunchKernelGGL((k_spmm_sweep<SWEEP_W, SWEEP_G>), dim3(256), dim3(SWEEP_W * 64), 0, c->stream, a, d->part_row0, d->sweep0, d->nsteps, d->hdr_off,
                       d->batch_off, d->flush_off, d->codes, d->vals, d->offs, d->flush_rows, X, Xg, Y);
    RAILS_HIP_CHECK(hipGetLastError());
    A->last_kernel = "k_spmm_sweep";
    c->n_spmm_sweep++;
    *done = true;
    return RAILS_OK;
}

extern "C" int rails_csr_sweep_stats(rails_csr *A, int nc, double *out)
{
    RAILS_REQUIRE(A && out, "rails_csr_sweep_stats: null argument");
    out[0] = out[1] = out[2] = out[3] = 0.0;
    if (!A->sweep || nc % 16) return RAILS_OK;
    auto it = A->sweep->by_chunks.find(nc / 16);
    if (it == A->sweep->by_chunks.end() || !it->second->ok) return RAILS_OK;
    out[0] = it->second->host.efficiency;
    out[1] = it->second->host.staged_rows_per_row;
    out[2] = (double)it->second->host.trips;
    out[3] = 1.0;
    return RAILS_OK;
}
