// sweep_plan.h -- host-side plan of the LDS-window ("sweep") SpMM, spmm_sweep.hip.
//
// The kernel computes Y = A X (replaces `A_ * W`, src/LyapunovSolver.hpp:146) for matrices whose rows reference a
// bounded window of columns (banded patterns).  Every workgroup streams a contiguous range of X rows ONCE through a
// ring of LDS segments, 16 columns wide, while the partial sums of the rows it owns live in registers; the order in
// which the nonzeros are consumed is fixed per matrix on the host (this file), so that the device code has no
// searching, no comparisons and no divergence: it interprets a schedule.
//
// Geometry (all sizes are parameters of the plan; the kernel instantiates W, G and fixes the rest):
//   parts    the rows are cut into `parts` contiguous ranges, one per XCD (8): the workgroups of a part sweep the
//            same X rows at the same time, so each X row is fetched from HBM once per part and served to the others
//            by that XCD's L2
//   chunks   column chunks of 16 (a 128-B line per X row); `phases` = workgroups per (part, chunk)
//   blocks   a part's rows are cut into blocks of R = G * W * 16 rows; block j belongs to phase j % phases; a
//            workgroup owns one block at a time: G groups x W waves x 16 slots, one row per slot (a quad of lanes),
//            16 partial sums per row, four per lane
//   steps    sweep position s = column - sweep0(part); positions [k*SEG, (k+1)*SEG) live in ring segment k % NSEG; they are
//            asked for at the start of step k - ahead and have arrived at the start of step k, during which the lanes may read
//            positions [(k-NSEG+1+ahead)*SEG, (k+1)*SEG)
//   trips    one trip = every slot of a wave consumes at most one nonzero of its row (lock step); trips come in units
//            of four (the kernel's code and the layout of the stream) and are scheduled in entries of `entry_trips` = 4 or 2
//            trips; per step and group the schedule gives the number of entries; a group's partial sums are written
//            to Y when its block is done
#ifndef RAILS_SWEEP_PLAN_H
#define RAILS_SWEEP_PLAN_H

#include <cstdint>
#include <string>
#include <vector>

constexpr int RAILS_SWEEP_CODES = 64; // 32-bit entries per (program, step) record, one per lane of a wave: a header, then up to 63 entries
// An entry = entry_trips (four or two) trips of a group (or a flush without trips): bits 0-7 = 8 x group (the group's first partial-sum register
// relative to the first group's); the header: bits 0-7 = number of entries, bits 16-31 = quarters of a unit's time (128 cycles each) to sit out first (pacing).  Flags (the kernel tests them together after a unit):
constexpr uint32_t RAILS_SWEEP_FLUSH = 0x100;         // the group's partial sums go to Y after this entry
constexpr uint32_t RAILS_SWEEP_NO_TRIPS = 0x200;      // the entry has no trips (rows without nonzeros left: only the flush)
constexpr uint32_t RAILS_SWEEP_LAST = 0x400;          // last entry of the step
constexpr uint32_t RAILS_SWEEP_NEXT_NO_TRIPS = 0x800; // the next entry has no trips (in the header: the first entry)
constexpr int RAILS_SWEEP_ROW_SHIFT = 12;             // entries with FLUSH: bits 12-31 = (first row of the flush - first row of the part) / 16

struct rails_sweep_params {
    int waves = 8;      // W: waves per workgroup
    int groups = 22;    // G: row groups per wave (each: 16 slots = 16 rows)
    int seg_rows = 256; // SEG: X rows per step
    int nseg = 5;       // ring segments
    int ahead = 1;      // segments being filled at any time: step k asks for the rows of step k + ahead (NSEG - ahead are readable)
    int parts = 8;      // row ranges (XCDs)
    int phases = 4;     // workgroups per (part, column chunk)
    int level = 2;      // 1: the waves of a workgroup run the same number of units in every step; 2: and the workgroups of a part's column
                        // chunk keep pace (pauses); 0: every unit as late as its X rows allow
    int level_slack = 32; // units of time a workgroup may be ahead of the slowest one of its chunk (level 2)
    int level_min_fill = 16; // a unit run early to level must consume at least this many nonzeros (of 4 x 16; scaled with entry_trips)
    int entry_trips = 2;     // trips per schedule entry: 2 = half a unit (the two halves of a unit of the kernel's code may belong to different
                             // groups: 17 % fewer trips on the benchmark pattern for 4 % more instructions per trip; twice the entries per
                             // step, so patterns with heavy rows may only fit with whole units); 4 = a whole unit per entry
};

struct rails_sweep_plan {
    rails_sweep_params p;
    int64_t m = 0, ncols = 0, nnz = 0;
    std::vector<int64_t> part_row0;          // [parts + 1]
    std::vector<int64_t> sweep0;             // [parts] column of sweep position 0
    std::vector<int32_t> nsteps;             // [parts]
    // one program per (part, phase, wave): index (part * phases + phase) * W + wave
    std::vector<int64_t> hdr_off;            // [programs] offset into codes
    std::vector<int64_t> batch_off;          // [programs] first batch (16 trips) in vals / offs
    std::vector<int64_t> flush_off;          // [programs] offset into flush_rows
    std::vector<uint32_t> codes;             // per (program, step) RAILS_SWEEP_CODES entries (see above), units in group order
    std::vector<double> vals;                // per batch of 16 trips = 4 units: [unit / 2][slot 16][quad lane 4][unit % 2]
                                             // (lane (slot, quad lane) holds the slot's trip 4 unit + quad lane of every unit)
    std::vector<uint16_t> offs;              // per batch: [slot 16][quad lane 4][unit 4]: ring row of the X row to read
    std::vector<int32_t> flush_rows;         // first row (of the wave's 8 x G rows: + g * 64 * ... see kernel) per flush
    // statistics
    int max_units_per_step = 0;              // the busiest (wave, step)
    int64_t trips = 0;                       // lock-step trips over all programs
    int64_t entries = 0;                     // = nnz when feasible
    double efficiency = 0.0;                 // nnz / (16 * trips)
    double staged_rows_per_row = 0.0;        // X rows staged per matrix row and chunk (1 + window / R for a band)
    std::string why;                         // reason when not feasible
};

// Builds the plan; returns false (plan.why says why) when the pattern does not fit the scheme with these parameters:
// a block's rows still need X rows that have not arrived when the rows of the next block of the same workgroup lose theirs.
bool rails_sweep_plan_build(const rails_sweep_params &prm, int64_t m, int64_t ncols, const int64_t *rowptr, const int32_t *col,
                            const double *val, rails_sweep_plan &plan);

#endif
