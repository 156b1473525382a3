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

// refill buffer (VREGS, OREG) with the next batch: scalar base addresses + per-lane 32-bit offsets
#define RAILS_RQ(N, VREGS, OREG) "rq" #N "_%=: global_load_dwordx4 " VREGS ", %[voff], %[vb]\n\tglobal_load_dword " OREG ", %[ooff], %[ob]\n\ts_branch rqd_%=\n\t"

// The schedule pointers are separate __restrict__ kernel arguments: what is read through them is never written by the kernel.
template <int W, int G>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_num_vgpr(24))) void k_spmm_sweep(
    SweepArgs a, const int64_t *__restrict__ part_row0, const int64_t *__restrict__ sweep0_, const int32_t *__restrict__ nsteps_,
    const int64_t *__restrict__ hdr_off, const int64_t *__restrict__ batch_off, const int64_t *__restrict__ flush_off,
    const uint16_t *__restrict__ codes, const double *__restrict__ vals, const uint16_t *__restrict__ offs,
    const int32_t *__restrict__ flush_rows, const double *__restrict__ X, const double *__restrict__ Xg, double *__restrict__ Y)
{
    static_assert(SWEEP_ACC0 + 4 * G <= 200, "partial sums must end below the unit's registers");
    static_assert(RAILS_SWEEP_CODES == 128, "two schedule entries per lane");
    static_assert((SEG / 8) % W == 0, "every wave issues the same number of LDS-DMA instructions per step");
    static_assert(SEG / 8 / W + 1 == 5, "the counted waits are written for 4 LDS-DMA instructions + 1 record load per step");
    __shared__ __attribute__((aligned(128))) unsigned char ring[RING_BYTES];
    const int part = (int)(blockIdx.x % (unsigned)a.parts);
    const int local = (int)(blockIdx.x / (unsigned)a.parts);
    const int chunk = local % a.n_chunks;
    const int phase = local / a.n_chunks;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int slot = lane >> 3, q = lane & 7;
    // byte offset of this lane's two columns inside a ring row, plus the ring's address in LDS
    const uint32_t lane_off = (uint32_t)(q * 16) + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;
    const int64_t prog = ((int64_t)part * a.phases + phase) * W + wave;
    const uint32_t *cp = reinterpret_cast<const uint32_t *>(codes + hdr_off[prog]) + lane; // lane i reads entries 2 i and 2 i + 1 of a step's record
    const int32_t *fp = flush_rows + flush_off[prog];
    const int64_t b0 = batch_off[prog];
    // a batch = 16 trips of the wave: lane q of a slot holds the slot's (value, ring row) of trips q and q + 8
    // the wave's stream: scalar base addresses + per-lane 32-bit byte offsets that advance by one batch per request
    const uint64_t vb = (uint64_t)(uintptr_t)(vals + b0 * 128), ob = (uint64_t)(uintptr_t)(offs + b0 * 128);
    uint32_t voff = (uint32_t)(slot * 8 + q) * 16u, ooff = (uint32_t)(slot * 8 + q) * 4u;
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

    // zero the partial sums (this statement also tells the compiler which registers the assembly owns)
    asm volatile(".set rails_i, %c0\n\t.rept %c1\n\tv_mov_b32 v[rails_i], 0\n\t.set rails_i, rails_i + 1\n\t.endr" : : "n"(SWEEP_ACC0), "n"(4 * G)
                 : RAILS_SWEEP_RESERVED);

    stage(0, 0);
    // batches 0..4 into buffers 0..4; buffer 5 gets batch 5 when buffer 0 comes into use
    asm volatile("global_load_dwordx4 v[224:227], %0, %2\n\tglobal_load_dword v248, %1, %3\n\t"
                 "global_load_dwordx4 v[228:231], %0, %2 offset:1024\n\tglobal_load_dword v249, %1, %3 offset:256\n\t"
                 "global_load_dwordx4 v[232:235], %0, %2 offset:2048\n\tglobal_load_dword v250, %1, %3 offset:512\n\t"
                 "global_load_dwordx4 v[236:239], %0, %2 offset:3072\n\tglobal_load_dword v251, %1, %3 offset:768\n\t"
                 "v_add_u32 %0, 0x1000, %0\n\tv_add_u32 %1, 0x400, %1\n\t"
                 "global_load_dwordx4 v[240:243], %0, %2\n\tglobal_load_dword v252, %1, %3\n\t"
                 "v_add_u32 %0, 0x400, %0\n\tv_add_u32 %1, 0x100, %1"
                 : "+v"(voff), "+v"(ooff) : "s"(vb), "s"(ob) : "memory");
    int ub = 0;   // unit of the six batches in use next (0..23)
    int tss = 8;  // batch turns since the last step start (the step's 5 vector-memory instructions are younger than requests older than that)
    int fl = 0, seg = 0;
    uint32_t rec_next = cp[0];
    for (int k = 0; k < nsteps; ++k) {
        // step k's rows have landed for every wave, and every wave is done reading the segment refilled next
        __syncthreads();
        const uint32_t rec = rec_next;
        const int seg_next = seg + 1 == NSEG ? 0 : seg + 1;
        if (k + 1 < nsteps) {
            rec_next = cp[(int64_t)(k + 1) * (RAILS_SWEEP_CODES / 2)];
            if (!(a.ablate & 1)) {
                stage(k + 1, seg_next);
                tss = 0;
            }
        }
        seg = seg_next;
        const int n = (a.ablate & 2) ? 0 : (__builtin_amdgcn_readlane((int)rec, 0) & 0xffff);
        int i = 1;
        while (i <= n) {
            // Units i.. of the step until one asks for a flush (or the record ends).  Per unit: the batch turn every fourth unit
            // (wait for the buffer coming into use, refill the one released), the unit's (value, ring row) pairs out of the batch
            // registers, handed to the other quad of each slot, four ring rows requested, and per ring row as it arrives two
            // fused multiply-adds into the group's partial sums: the same chain of fused multiply-adds, in column order, as
            // the row-gather kernel.
            int code, t0, t1;
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n"
                "loop_%=:\n\t"
                "s_lshr_b32 %[t0], %[i], 1\n\t"
                "v_readlane_b32 %[code], %[rec], %[t0]\n\t"
                "s_lshl_b32 %[t0], %[i], 4\n\t"
                "s_and_b32 %[t0], %[t0], 16\n\t"
                "s_lshr_b32 %[code], %[code], %[t0]\n\t"
                "s_and_b32 %[code], %[code], 0xffff\n\t"
                "s_add_u32 %[i], %[i], 1\n\t"
                "s_bitcmp1_b32 %[code], 7\n\t"
                "s_cbranch_scc1 tail_%=\n\t"
                // batch turn
                "s_and_b32 %[t0], %[ub], 3\n\t"
                "s_cmp_lg_u32 %[t0], 0\n\t"
                "s_cbranch_scc1 noturn_%=\n\t"
                "s_cmp_lt_u32 %[tss], 5\n\t"
                "s_cbranch_scc1 w13_%=\n\t"
                "s_waitcnt vmcnt(8)\n\t"
                "s_branch wd_%=\n"
                "w13_%=: s_waitcnt vmcnt(13)\n"
                "wd_%=:\n\t"
                "s_add_u32 %[tss], %[tss], 1\n\t"
                "s_lshr_b32 %[t0], %[ub], 2\n\t"
                "s_cmp_eq_u32 %[t0], 0\n\ts_cbranch_scc1 rq0_%=\n\t"
                "s_cmp_eq_u32 %[t0], 1\n\ts_cbranch_scc1 rq1_%=\n\t"
                "s_cmp_eq_u32 %[t0], 2\n\ts_cbranch_scc1 rq2_%=\n\t"
                "s_cmp_eq_u32 %[t0], 3\n\ts_cbranch_scc1 rq3_%=\n\t"
                "s_cmp_eq_u32 %[t0], 4\n\ts_cbranch_scc1 rq4_%=\n\t"
                "global_load_dwordx4 v[240:243], %[voff], %[vb]\n\tglobal_load_dword v252, %[ooff], %[ob]\n\ts_branch rqd_%=\n\t"
                RAILS_RQ(0, "v[244:247]", "v253") RAILS_RQ(1, "v[224:227]", "v248") RAILS_RQ(2, "v[228:231]", "v249")
                RAILS_RQ(3, "v[232:235]", "v250") RAILS_RQ(4, "v[236:239]", "v251")
                "rqd_%=:\n\t"
                "v_add_u32 %[voff], 0x400, %[voff]\n\tv_add_u32 %[ooff], 0x100, %[ooff]\n"
                "noturn_%=:\n\t"
                // the unit's pairs: value registers v[224 + 2 (ub / 2) ..], 16-bit half (ub / 2) % 2 of v[248 + ub / 4]
                "s_and_b32 %[t0], %[ub], -2\n\t"
                "s_lshr_b32 %[t1], %[ub], 2\n\t"
                "s_set_gpr_idx_on %[t0], 0x1\n\t"
                "v_mov_b32 v220, v224\n\t"
                "v_mov_b32 v221, v225\n\t"
                "s_set_gpr_idx_on %[t1], 0x1\n\t"
                "v_mov_b32 v222, v248\n\t"
                "s_set_gpr_idx_off\n\t"
                "s_lshl_b32 %[t0], %[ub], 3\n\t"
                "s_and_b32 %[t0], %[t0], 16\n\t"
                "v_bfe_u32 v222, v222, %[t0], 16\n\t"
                "v_lshlrev_b32 v222, 7, v222\n\t"
                // even units have them in the low quad of every slot, odd units in the high quad: hand them to the other quad
                // (VALU write -> DPP read of the same register: two wait states)
                "s_bitcmp1_b32 %[ub], 0\n\t"
                "s_cbranch_scc1 odd_%=\n\t"
                "s_nop 0\n\t"
                "v_mov_b32_dpp v220, v220 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_mov_b32_dpp v221, v221 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_mov_b32_dpp v222, v222 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                "s_branch pd_%=\n"
                "odd_%=:\n\t"
                "s_nop 0\n\t"
                "v_mov_b32_dpp v220, v220 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                "v_mov_b32_dpp v221, v221 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                "v_mov_b32_dpp v222, v222 row_shl:4 row_mask:0xf bank_mask:0x5\n"
                "pd_%=:\n\t"
                "s_add_u32 %[ub], %[ub], 1\n\t"
                "s_cmp_eq_u32 %[ub], 24\n\t"
                "s_cselect_b32 %[ub], 0, %[ub]\n\t"
                "s_and_b32 %[t0], %[code], 63\n\t"
                "s_lshl_b32 %[t0], %[t0], 2\n\t"
                // four ring rows
                "v_add_u32_dpp v223, v222, %[loff] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_add_u32_dpp v254, v222, %[loff] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "ds_read_b128 v[200:203], v223\n\t"
                "ds_read_b128 v[204:207], v254\n\t"
                "v_add_u32_dpp v223, v222, %[loff] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_add_u32_dpp v254, v222, %[loff] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "ds_read_b128 v[208:211], v223\n\t"
                "ds_read_b128 v[212:215], v254\n\t"
                // values of trips 0, 1 to every lane of the slot; multiply-adds as the rows arrive
                "v_mov_b32_dpp v216, v220 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v217, v221 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v218, v220 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v219, v221 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "s_set_gpr_idx_on %[t0], 0xc\n\t"
                "s_waitcnt lgkmcnt(3)\n\t"
                "v_fma_f64 v[%c[acc]:%c[acc]+1], v[216:217], v[200:201], v[%c[acc]:%c[acc]+1]\n\t"
                "v_fma_f64 v[%c[acc]+2:%c[acc]+3], v[216:217], v[202:203], v[%c[acc]+2:%c[acc]+3]\n\t"
                "s_waitcnt lgkmcnt(2)\n\t"
                "v_fma_f64 v[%c[acc]:%c[acc]+1], v[218:219], v[204:205], v[%c[acc]:%c[acc]+1]\n\t"
                "v_fma_f64 v[%c[acc]+2:%c[acc]+3], v[218:219], v[206:207], v[%c[acc]+2:%c[acc]+3]\n\t"
                "s_set_gpr_idx_off\n\t"
                "v_mov_b32_dpp v216, v220 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v217, v221 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v218, v220 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp v219, v221 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "s_set_gpr_idx_on %[t0], 0xc\n\t"
                "s_waitcnt lgkmcnt(1)\n\t"
                "v_fma_f64 v[%c[acc]:%c[acc]+1], v[216:217], v[208:209], v[%c[acc]:%c[acc]+1]\n\t"
                "v_fma_f64 v[%c[acc]+2:%c[acc]+3], v[216:217], v[210:211], v[%c[acc]+2:%c[acc]+3]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_fma_f64 v[%c[acc]:%c[acc]+1], v[218:219], v[212:213], v[%c[acc]:%c[acc]+1]\n\t"
                "v_fma_f64 v[%c[acc]+2:%c[acc]+3], v[218:219], v[214:215], v[%c[acc]+2:%c[acc]+3]\n\t"
                "s_set_gpr_idx_off\n"
                "tail_%=:\n\t"
                "s_bitcmp1_b32 %[code], 6\n\t"
                "s_cbranch_scc1 out_%=\n\t"
                "s_cmp_le_u32 %[i], %[n]\n\t"
                "s_cbranch_scc1 loop_%=\n\t"
                "s_mov_b32 %[code], 0\n"
                "out_%=:"
                : [i] "+s"(i), [ub] "+s"(ub), [tss] "+s"(tss), [voff] "+v"(voff), [ooff] "+v"(ooff), [code] "=&s"(code), [t0] "=&s"(t0), [t1] "=&s"(t1)
                : [rec] "v"(rec), [n] "s"(n), [loff] "v"(lane_off), [vb] "s"(vb), [ob] "s"(ob), [acc] "n"(SWEEP_ACC0)
                : "memory", "scc", "m0");
            if (code & 0x40) {
                // the group's block is done: its partial sums go to Y and start again from zero
                int s0, s1, s2, s3;
                asm volatile("s_set_gpr_idx_on %4, 0x1\n\tv_mov_b32 %0, v%c5\n\tv_mov_b32 %1, v%c6\n\tv_mov_b32 %2, v%c7\n\tv_mov_b32 %3, v%c8\n\t"
                             "s_set_gpr_idx_on %4, 0x8\n\tv_mov_b32 v%c5, 0\n\tv_mov_b32 v%c6, 0\n\tv_mov_b32 v%c7, 0\n\tv_mov_b32 v%c8, 0\n\t"
                             "s_set_gpr_idx_off"
                             : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                             : "s"((code & 63) * 4), "n"(SWEEP_ACC0), "n"(SWEEP_ACC0 + 1), "n"(SWEEP_ACC0 + 2), "n"(SWEEP_ACC0 + 3)
                             : "m0");
                const int64_t row = (int64_t)__builtin_amdgcn_readfirstlane(fp[fl++]) + slot;
                if (row < row_end) {
                    double2_t out;
                    out.x = __hiloint2double(s1, s0);
                    out.y = __hiloint2double(s3, s2);
                    *reinterpret_cast<double2_t *>(Y + row * a.ldy + col0 + q * 2) = out;
                }
            }
        }
    }
}

struct DevPlan {
    rails_sweep_plan host; // kept for its small arrays and statistics (the big arrays are released after the upload)
    int64_t *part_row0 = nullptr, *sweep0 = nullptr, *hdr_off = nullptr, *batch_off = nullptr, *flush_off = nullptr;
    int32_t *nsteps = nullptr, *flush_rows = nullptr;
    uint16_t *codes = nullptr;
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
            std::vector<uint16_t>().swap(d->host.codes);
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
