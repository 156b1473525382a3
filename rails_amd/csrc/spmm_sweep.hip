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
    int layout;  // experiments only (RAILS_SWEEP_LAYOUT): 1 = every lane reads 32 contiguous bytes of a ring row
    int ablate; // experiments only (RAILS_SWEEP_ABLATE): 1 = no LDS-DMA after the first step, 2 = no trips, 4 = no barriers; 16 / 32 / 48 = the
                // builds without ring-row reads / multiply-adds / stream waits (results are wrong in every case)
};

// ---- register plan -------------------------------------------------------------------------------------------------
// The compiler gets v0-v23 (amdgpu_num_vgpr(24)): per-step work only (LDS-DMA addresses, the step's record, the rare flush).
// Everything above is addressed by the inline assembly below only:
//   v[24 .. 24 + 4 G)     the partial sums: group g = v[24 + 4 g .. 24 + 4 g + 3] (two doubles per lane), read and written through
//                         M0-relative register indexing (s_set_gpr_idx_on), so that ONE copy of the unit's code serves every
//                         group: with the groups as a register array indexed by constants the loop over groups has to be
//                         unrolled, and 44 copies of a unit's code (80 KB) thrash the instruction cache (4.0 ms per product)
//   v[200:215]            the four ring rows of a unit; v[216:219] two broadcast values; v220-v222 the unit's (value lo, value hi,
//                         ring row << 7) pairs; v223, v254 ring addresses
//   v[224 + 4 j .. 227 + 4 j], v[248 + j], j = 0..5   six batches of (value, ring row) pairs, used in turn; a batch is requested
//                         five batches (80 trips) before its first use and waited for with a counted vmcnt.  They land where they
//                         are read from: a compiler-managed destination gets copied, behind vmcnt(0), right after the load is issued.
// The unit loop of a step is ONE assembly block: compiled from C++ the same loop took 44 vector + 42 scalar instructions and
// 10 branches per unit, half of the time waiting (rocprofv3, profiles/): here ~31 + ~25 with the waits counted per ring row.
constexpr int SWEEP_ACC0 = 24;
#define RAILS_SWEEP_RESERVED "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

// (the assembly names registers above the compiler's share and M0 in its clobber lists on purpose)
#pragma clang diagnostic ignored "-Winline-asm"
#define RAILS_SW_NAME k_spmm_sweep
#define RAILS_SW_ABLATE 0 /* the product kernel: no experiment switches */
#define RAILS_SW_PIPELINED 1
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n\t"
#define RAILS_SW_FMA(TEXT) TEXT
#define RAILS_SW_VMWAIT(TEXT) TEXT
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
// the same kernel for schedules whose entries are half units (rails_sweep_params::entry_trips == 2): every half of a unit of the code
// fetches its own entry, so that the two halves may serve different groups
#define RAILS_SW_NAME k_spmm_sweep_h2
#define RAILS_SW_ABLATE 0
#define RAILS_SW_PIPELINED 0
#define RAILS_SW_ENTRY_TRIPS 2
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n\t"
#define RAILS_SW_FMA(TEXT) TEXT
#define RAILS_SW_VMWAIT(TEXT) TEXT
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
// ---- experiment builds (timings only: their results are wrong by construction).  Compiled only with -DRAILS_SWEEP_EXPERIMENTS
// (`make EXPERIMENTS=1`, what scripts/gpu_sweep*.sh use); the shipped library has none of them and refuses RAILS_SWEEP_ABLATE / RAILS_SWEEP_LAYOUT.
#ifdef RAILS_SWEEP_EXPERIMENTS
// the full kernel with the run-time experiment switches (RAILS_SWEEP_ABLATE & 7: no LDS-DMA / no trips / no barriers)
#define RAILS_SW_NAME k_spmm_sweep_switches
#define RAILS_SW_ABLATE a.ablate
#define RAILS_SW_PIPELINED 1
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n\t"
#define RAILS_SW_FMA(TEXT) TEXT
#define RAILS_SW_VMWAIT(TEXT) TEXT
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
// experiment: the four trips of a unit as two separate halves (reads of two trips, their multiply-adds, then the next two)
#define RAILS_SW_NAME k_spmm_sweep_halves
#define RAILS_SW_ABLATE 0
#define RAILS_SW_PIPELINED 0
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n\t"
#define RAILS_SW_FMA(TEXT) TEXT
#define RAILS_SW_VMWAIT(TEXT) TEXT
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
// experiments: without the ring-row reads / the multiply-adds / the waits for the schedule stream
#define RAILS_SW_NAME k_spmm_sweep_noread
#define RAILS_SW_ABLATE a.ablate
#define RAILS_SW_PIPELINED 0
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) ""
#define RAILS_SW_FMA(TEXT) TEXT
#define RAILS_SW_VMWAIT(TEXT) TEXT
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
#define RAILS_SW_NAME k_spmm_sweep_nofma
#define RAILS_SW_ABLATE a.ablate
#define RAILS_SW_PIPELINED 0
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n\t"
#define RAILS_SW_FMA(TEXT) ""
#define RAILS_SW_VMWAIT(TEXT) TEXT
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
#define RAILS_SW_NAME k_spmm_sweep_nowait
#define RAILS_SW_ABLATE a.ablate
#define RAILS_SW_PIPELINED 0
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n\t"
#define RAILS_SW_FMA(TEXT) TEXT
#define RAILS_SW_VMWAIT(TEXT) ""
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
#define RAILS_SW_NAME k_spmm_sweep_noidx
#define RAILS_SW_ABLATE a.ablate
#define RAILS_SW_PIPELINED 0
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n\t"
#define RAILS_SW_FMA(TEXT) TEXT
#define RAILS_SW_VMWAIT(TEXT) TEXT
#define RAILS_SW_IDX(TEXT) ""
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
#define RAILS_SW_NAME k_spmm_sweep_bare
#define RAILS_SW_ABLATE a.ablate
#define RAILS_SW_PIPELINED 0
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) ""
#define RAILS_SW_FMA(TEXT) ""
#define RAILS_SW_VMWAIT(TEXT) ""
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX "_dpp"
#define RAILS_SW_QP(T) "quad_perm:[" #T "," #T "," #T "," #T "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#include "spmm_sweep_kernel.inc"
#define RAILS_SW_NAME k_spmm_sweep_nodpp
#define RAILS_SW_ABLATE a.ablate
#define RAILS_SW_PIPELINED 0
#define RAILS_SW_ENTRY_TRIPS 4
#define RAILS_SW_WAITA "s_waitcnt lgkmcnt(2)\n\t"
#define RAILS_SW_WAITB "s_waitcnt lgkmcnt(0)\n\t"
#define RAILS_SW_READ(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n\t"
#define RAILS_SW_FMA(TEXT) TEXT
#define RAILS_SW_VMWAIT(TEXT) TEXT
#define RAILS_SW_IDX(TEXT) TEXT
#define RAILS_SW_DPPSFX ""
#define RAILS_SW_QP(T) ""
#include "spmm_sweep_kernel.inc"
#endif // RAILS_SWEEP_EXPERIMENTS

struct DevPlan {
    rails_sweep_plan host; // kept for its small arrays and statistics (the big arrays are released after the upload)
    int64_t *part_row0 = nullptr, *sweep0 = nullptr, *hdr_off = nullptr, *batch_off = nullptr, *flush_off = nullptr;
    int32_t *nsteps = nullptr, *flush_rows = nullptr;
    uint32_t *codes = nullptr;
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
    std::map<int, int> asked; // products of that width seen in automatic mode before the schedule exists
    // row-partitioned operators: schedules of the INTERIOR rows [int_lo, int_hi) (no ghost columns: a rectangular operator over the local
    // X rows), whose product runs beside the halo exchange (rails_spmm_sweep_interior)
    std::map<int, DevPlan *> interior;
    std::map<int, int> asked_interior;
};

void rails_sweep_release(rails_csr *A)
{
    if (!A->sweep) return;
    for (auto &kv : A->sweep->by_chunks) free_plan(kv.second);
    for (auto &kv : A->sweep->interior) free_plan(kv.second);
    delete A->sweep;
    A->sweep = nullptr;
}

// the geometry the kernel is instantiated for
static constexpr int SWEEP_W = 8, SWEEP_G = 22;

// The schedule of A for n_chunks column chunks, built (host, a second per million rows) and copied to the device when missing.
static int ensure_plan(rails_ctx *c, rails_csr *A, int n_chunks, DevPlan **out, bool interior = false)
{
    if (!A->sweep) A->sweep = new rails_sweep_cache();
    DevPlan *&d = interior ? A->sweep->interior[n_chunks] : A->sweep->by_chunks[n_chunks];
    // the rows the schedule covers and the X rows they may reference: the whole (square) operator, or the interior rows over the local rows
    const int64_t plan_rows = interior ? A->int_hi - A->int_lo : A->m, plan_cols = interior ? A->m : A->ncols_ext;
    const int64_t *plan_rowptr = A->h_rowptr.data() + (interior ? A->int_lo : 0);
    if (!d) {
        d = new DevPlan();
        rails_sweep_params prm;
        prm.waves = SWEEP_W;
        prm.groups = SWEEP_G;
        prm.seg_rows = SEG;
        prm.nseg = NSEG;
        prm.parts = 8;
        prm.phases = 32 / n_chunks;
        if (getenv("RAILS_SWEEP_LEVEL")) prm.level = atoi(getenv("RAILS_SWEEP_LEVEL")); // experiments: 0 = no levelling of the waves
        if (getenv("RAILS_SWEEP_MIN_FILL")) prm.level_min_fill = atoi(getenv("RAILS_SWEEP_MIN_FILL"));
        if (getenv("RAILS_SWEEP_SLACK")) prm.level_slack = atoi(getenv("RAILS_SWEEP_SLACK"));
        if (getenv("RAILS_SWEEP_ENTRY_TRIPS")) prm.entry_trips = atoi(getenv("RAILS_SWEEP_ENTRY_TRIPS")) == 4 ? 4 : 2; // 4: entries of whole units (k_spmm_sweep)
        if (getenv("RAILS_SWEEP_ABLATE") && atoi(getenv("RAILS_SWEEP_ABLATE"))) prm.entry_trips = 4;                  // (the experiment builds take those)
        bool built = rails_sweep_plan_build(prm, plan_rows, plan_cols, plan_rowptr, A->h_col.data(), A->h_val.data(), d->host);
        if (!built && prm.entry_trips == 2) { // heavy rows: twice the entries do not fit a step's record; whole units may
            prm.entry_trips = 4;
            built = rails_sweep_plan_build(prm, plan_rows, plan_cols, plan_rowptr, A->h_col.data(), A->h_val.data(), d->host);
        }
        if (built) {
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
            RAILS_HIP_CHECK(rails_stream_sync(c));
            std::vector<double>().swap(d->host.vals);
            std::vector<uint16_t>().swap(d->host.offs);
            std::vector<uint32_t>().swap(d->host.codes);
            d->ok = true;
        }
    }
    *out = d;
    return RAILS_OK;
}

static bool sweep_shape_ok(const rails_ctx *c, const rails_csr *A, int nc, bool aligned)
{
    const int n_chunks = nc / 16;
    return aligned && nc % 16 == 0 && n_chunks >= 1 && n_chunks <= 32 && 32 % n_chunks == 0 && c->num_cu >= 256 && A->n_ghost == 0 &&
           A->ncols_ext < 0x7fffffffLL;
}

// Builds the schedule for nc columns ahead of the products (rails_csr_prepare): *fits says whether the kernel will take them.
int rails_sweep_prepare(rails_ctx *c, rails_csr *A, int nc, bool *fits)
{
    *fits = false;
    if (!sweep_shape_ok(c, A, nc, true)) return RAILS_OK;
    DevPlan *d = nullptr;
    RAILS_TRY(ensure_plan(c, A, nc / 16, &d));
    *fits = d->ok && d->host.efficiency >= 0.4 && d->host.staged_rows_per_row <= 8.0;
    return RAILS_OK;
}

// *done = true when the product was computed here.  force: fail instead of declining.  In automatic mode (!force) the schedule is not
// built by the first product that could use it: it costs about a thousand products' worth of the time it saves each of them, so the
// operator waits until it has been asked for RAILS_SWEEP_AFTER (16) products of that width -- a sign of a caller that streams panels
// through A -- or until rails_csr_prepare says so.
int rails_spmm_sweep(rails_ctx *c, rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, bool aligned,
                     bool force, bool *done)
{
    *done = false;
    const int n_chunks = nc / 16;
    if (!sweep_shape_ok(c, A, nc, aligned)) {
        RAILS_REQUIRE(!force, "rails_spmm: the sweep kernel needs a multiple of 16 columns that divides 512, even column offsets, 256 CUs and no ghost rows");
        return RAILS_OK;
    }
    if (!A->sweep) A->sweep = new rails_sweep_cache();
    if (!force && !A->sweep->by_chunks.count(n_chunks)) {
        static const int after = getenv("RAILS_SWEEP_AFTER") ? atoi(getenv("RAILS_SWEEP_AFTER")) : 16;
        if (++A->sweep->asked[n_chunks] < after) return RAILS_OK;
    }
    DevPlan *d = nullptr;
    RAILS_TRY(ensure_plan(c, A, n_chunks, &d));
    if (!d->ok) {
        RAILS_REQUIRE(!force, "rails_spmm: the sweep kernel does not fit this operator: %s", d->host.why.c_str());
        return RAILS_OK;
    }
    // worth it only where the lock-step trips are reasonably full (the units run early to level the waves count as trips: 0.52 on the
    // banded-random pattern) and a row block re-uses what it stages
    if (!force && (d->host.efficiency < 0.4 || d->host.staged_rows_per_row > 8.0)) return RAILS_OK;
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
    static const int layout = getenv("RAILS_SWEEP_LAYOUT") ? atoi(getenv("RAILS_SWEEP_LAYOUT")) : 0;
#ifndef RAILS_SWEEP_EXPERIMENTS
    // the experiment switches select kernels whose results are wrong by construction: a library built without them refuses the product
    // instead of returning one (a leftover export of a profiling script must not corrupt a solve)
    RAILS_REQUIRE(!ablate && !layout, "rails_spmm: RAILS_SWEEP_ABLATE / RAILS_SWEEP_LAYOUT are set, but this library has no experiment builds of the sweep kernel (make EXPERIMENTS=1)");
#endif
    a.ablate = ablate;
    a.layout = layout;
    const char *launched = d->host.p.entry_trips == 2 ? "k_spmm_sweep_h2" : "k_spmm_sweep";
#define RAILS_SWEEP_LAUNCH(K)                                                                                                                     \
    RAILS_LAUNCH((K<SWEEP_W, SWEEP_G>), dim3(256), dim3(SWEEP_W * 64), 0, c->stream, a, d->part_row0, d->sweep0, d->nsteps, d->hdr_off, \
                       d->batch_off, d->codes, d->vals, d->offs, X, Xg, Y)
    if (d->host.p.entry_trips == 2) {
        RAILS_REQUIRE(!ablate, "rails_spmm: the experiment builds of the sweep kernel take schedules with entries of four trips (RAILS_SWEEP_ENTRY_TRIPS)");
        RAILS_SWEEP_LAUNCH(k_spmm_sweep_h2);
    } else
#ifdef RAILS_SWEEP_EXPERIMENTS
    switch ((ablate >> 4) & 15) {
    case 1: RAILS_SWEEP_LAUNCH(k_spmm_sweep_noread); launched = "k_spmm_sweep_noread (experiment: wrong result)"; break;
    case 2: RAILS_SWEEP_LAUNCH(k_spmm_sweep_nofma); launched = "k_spmm_sweep_nofma (experiment: wrong result)"; break;
    case 3: RAILS_SWEEP_LAUNCH(k_spmm_sweep_nowait); launched = "k_spmm_sweep_nowait (experiment: wrong result)"; break;
    case 4: RAILS_SWEEP_LAUNCH(k_spmm_sweep_noidx); launched = "k_spmm_sweep_noidx (experiment: wrong result)"; break;
    case 5: RAILS_SWEEP_LAUNCH(k_spmm_sweep_nodpp); launched = "k_spmm_sweep_nodpp (experiment: wrong result)"; break;
    case 6: RAILS_SWEEP_LAUNCH(k_spmm_sweep_bare); launched = "k_spmm_sweep_bare (experiment: wrong result)"; break;
    case 7: RAILS_SWEEP_LAUNCH(k_spmm_sweep_halves); launched = "k_spmm_sweep_halves (experiment)"; break;
    default:
        if (ablate) {
            RAILS_SWEEP_LAUNCH(k_spmm_sweep_switches);
            launched = "k_spmm_sweep_switches (experiment: wrong result)";
        } else
            RAILS_SWEEP_LAUNCH(k_spmm_sweep);
        break;
    }
#else
        RAILS_SWEEP_LAUNCH(k_spmm_sweep);
#endif
    RAILS_HIP_CHECK(hipGetLastError());
    A->last_kernel = launched;
    c->n_spmm_sweep++;
    *done = true;
    return RAILS_OK;
}

// The interior rows of a row-partitioned operator (rails_csr_set_halo: rows [int_lo, int_hi) reference local X rows only) as a sweep of
// their own, on stream st while the ghost rows travel (rails_spmm).  prepare_only: build the schedule if the shape qualifies.
static bool sweep_interior_shape_ok(const rails_ctx *c, const rails_csr *A, int nc, bool aligned)
{
    const int n_chunks = nc / 16;
    if (!(aligned && nc % 16 == 0 && nc >= 64 && n_chunks <= 32 && 32 % n_chunks == 0 && c->num_cu >= 256 && A->n_ghost > 0 && !A->rect && A->m < 0x7fffffffLL)) return false;
    const int64_t phases = 32 / n_chunks, rows = A->int_hi - A->int_lo, part_rows = rows / 8;
    return A->window_rows_int > 0 && A->window_rows_int + 256 <= (phases - 1) * 2816 && rows >= 8 * phases * 2816 &&
           (double)phases * (double)(part_rows + A->window_rows_int + 1024) <= 8.0 * (double)part_rows;
}

int rails_sweep_prepare_interior(rails_ctx *c, rails_csr *A, int nc, bool *fits)
{
    *fits = false;
    if (!sweep_interior_shape_ok(c, A, nc, true)) return RAILS_OK;
    DevPlan *d = nullptr;
    RAILS_TRY(ensure_plan(c, A, nc / 16, &d, true));
    *fits = d->ok && d->host.efficiency >= 0.4 && d->host.staged_rows_per_row <= 8.0;
    return RAILS_OK;
}

int rails_spmm_sweep_interior(rails_ctx *c, rails_csr *A, const double *X, int ldx, double *Y, int ldy, int nc, bool aligned, hipStream_t st, bool *done)
{
    *done = false;
    if (A->variant != 0 || !sweep_interior_shape_ok(c, A, nc, aligned)) return RAILS_OK;
    const int n_chunks = nc / 16;
    if (!A->sweep) A->sweep = new rails_sweep_cache();
    if (!A->sweep->interior.count(n_chunks)) {
        static const int after = getenv("RAILS_SWEEP_AFTER") ? atoi(getenv("RAILS_SWEEP_AFTER")) : 16;
        if (++A->sweep->asked_interior[n_chunks] < after) return RAILS_OK;
    }
    DevPlan *d = nullptr;
    RAILS_TRY(ensure_plan(c, A, n_chunks, &d, true));
    if (!d->ok || d->host.efficiency < 0.4 || d->host.staged_rows_per_row > 8.0) return RAILS_OK;
    SweepArgs a;
    a.ldx = ldx;
    a.ldg = ldx;
    a.ldy = ldy;
    a.m = A->m;     // X rows below this index come from X: all of them (the interior rows have no ghost columns)
    a.ncols = A->m;
    a.parts = 8;
    a.n_chunks = n_chunks;
    a.phases = 32 / n_chunks;
    a.ablate = 0;
    a.layout = 0;
    double *Yi = Y + A->int_lo * ldy;
    if (d->host.p.entry_trips == 2)
        hipLaunchKernelGGL((k_spmm_sweep_h2<SWEEP_W, SWEEP_G>), dim3(256), dim3(SWEEP_W * 64), 0, st, a, d->part_row0, d->sweep0, d->nsteps, d->hdr_off, d->batch_off, d->codes,
                           d->vals, d->offs, X, X, Yi);
    else
        hipLaunchKernelGGL((k_spmm_sweep<SWEEP_W, SWEEP_G>), dim3(256), dim3(SWEEP_W * 64), 0, st, a, d->part_row0, d->sweep0, d->nsteps, d->hdr_off, d->batch_off, d->codes,
                           d->vals, d->offs, X, X, Yi);
    RAILS_HIP_CHECK(hipGetLastError());
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
