// sweep_plan.cpp -- builds the schedule the sweep SpMM kernel interprets (see sweep_plan.h).  Host code only.
#include "sweep_plan.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int SLOTS = 16; // rows of a group in one wave: a slot is a quad of lanes, each lane four of the row's 16 columns

// the schedule of one group: its units (four trips each, or a flush without trips) in execution order
struct GroupSchedule {
    std::vector<int> step;        // per unit: the step it runs in
    std::vector<int> earliest;    // per unit: the first step at which all its X rows are in the ring
    std::vector<uint8_t> flags;   // per unit: 1 = the group's partial sums go to Y afterwards, 2 = no trips
    std::vector<double> val;      // [units with trips][4][SLOTS]
    std::vector<uint16_t> off;
    std::vector<int32_t> flush;   // first row of the wave's slot octet, per flush, in order
};

struct Ctx {
    const rails_sweep_params *prm;
    const int64_t *rowptr;
    const int32_t *col;
    const double *val;
    int64_t r0, r1;     // rows of the part
    int64_t sweep0;
    int nsteps;
    int64_t R;          // rows per block
    int64_t nblocks;
};

// schedule of one (phase, wave, group) of a part; false = not feasible
bool schedule_group(const Ctx &c, int phase, int wave, int g, GroupSchedule &out, std::string &why)
{
    const rails_sweep_params &P = *c.prm;
    const int SEG = P.seg_rows, NSEG = P.nseg;
    const int ring = SEG * NSEG;
    out.step.clear();
    out.earliest.clear();
    out.flags.clear();
    out.val.clear();
    out.off.clear();
    out.flush.clear();
    int64_t bj = phase; // current block of this workgroup
    // state of the current block
    int64_t p[SLOTS], pe[SLOTS]; // next / end nonzero of each slot's row
    int64_t last_pos[SLOTS];
    bool loaded = false, any_valid = false;
    int64_t cur_row0 = 0;
    auto load_block = [&](int64_t j) {
        cur_row0 = c.r0 + j * c.R + (int64_t)g * P.waves * SLOTS + (int64_t)wave * SLOTS;
        any_valid = false;
        for (int s = 0; s < SLOTS; ++s) {
            const int64_t row = cur_row0 + s;
            last_pos[s] = -1;
            if (row < c.r1 && row < c.r0 + (j + 1) * c.R) {
                p[s] = c.rowptr[row];
                pe[s] = c.rowptr[row + 1];
                any_valid = true;
            } else
                p[s] = pe[s] = 0;
        }
        loaded = true;
    };
    // first sweep segment of the first nonzero of block j for this group (INT64_MAX: none)
    auto first_seg = [&](int64_t j) -> int64_t {
        if (j >= c.nblocks) return INT64_MAX;
        const int64_t row0 = c.r0 + j * c.R + (int64_t)g * P.waves * SLOTS + (int64_t)wave * SLOTS;
        int64_t best = INT64_MAX;
        for (int s = 0; s < SLOTS; ++s) {
            const int64_t row = row0 + s;
            if (row < c.r1 && row < c.r0 + (j + 1) * c.R && c.rowptr[row + 1] > c.rowptr[row])
                best = std::min<int64_t>(best, ((int64_t)c.col[c.rowptr[row]] - c.sweep0) / SEG);
        }
        return best;
    };
    int64_t next_first = INT64_MAX;
    for (int k = 0; k < c.nsteps; ++k) {
        if (!loaded) {
            if (bj >= c.nblocks) break;
            load_block(bj);
            // the next block WITH nonzeros bounds how long this one may take
            next_first = INT64_MAX;
            for (int64_t j = bj + P.phases; j < c.nblocks && next_first == INT64_MAX; j += P.phases) next_first = first_seg(j);
        }
        const int64_t hi = (int64_t)(k + 1) * SEG;
        const int64_t lo = (int64_t)(k - NSEG + 2) * SEG;                            // readable: [lo, hi)
        const int64_t lo_next = (k == c.nsteps - 1) ? INT64_MAX : lo + SEG;          // gone after this step
        int forced = 0, remaining = 0, ready = INT_MAX;
        bool all_available = true;
        for (int s = 0; s < SLOTS; ++s) {
            int f = 0, av = 0;
            for (int64_t q = p[s]; q < pe[s]; ++q) {
                const int64_t pos = (int64_t)c.col[q] - c.sweep0;
                if (pos < lo) {
                    why = "a nonzero lost its X row before it was consumed (columns of a row not sorted?)";
                    return false;
                }
                if (pos >= hi) break;
                ++av;
                if (pos < lo_next) ++f;
            }
            forced = std::max(forced, f);
            if (pe[s] > p[s]) ready = std::min(ready, av);
            remaining = std::max<int>(remaining, (int)(pe[s] - p[s]));
            if (pe[s] > p[s] && (int64_t)c.col[pe[s] - 1] - c.sweep0 >= hi) all_available = false;
        }
        // the block has to be done before the first X row of the workgroup's next block leaves the ring
        const bool must_finish = next_first != INT64_MAX && (int64_t)k >= next_first + NSEG - 3;
        // what has to go now, or more where every slot with work left can fill whole units (no idle slots: spreads bursts for free)
        int T = std::max(forced, ready == INT_MAX ? 0 : ready / 4 * 4);
        if (must_finish || remaining == 0) {
            if (!all_available) {
                why = "the column window of a row block is wider than (phases - 1) blocks";
                return false;
            }
            T = remaining;
        }
        // trips come in units of four (the kernel's unit: four ring rows in flight, then four multiply-adds per lane)
        const int units = (T + 3) / 4;
        T = units * 4;
        if (units > 127) {
            why = "more than 508 nonzeros of one row inside one ring of X rows";
            return false;
        }
        for (int u = 0; u < units; ++u) {
            double v[4][SLOTS];
            int64_t o[4][SLOTS];
            int64_t first_real = -1, newest = -1;
            for (int t = 0; t < 4; ++t)
                for (int s = 0; s < SLOTS; ++s) {
                    v[t][s] = 0.0;
                    o[t][s] = -1;
                    if (p[s] < pe[s]) {
                        const int64_t pos = (int64_t)c.col[p[s]] - c.sweep0;
                        if (pos < hi) {
                            v[t][s] = c.val[p[s]];
                            o[t][s] = pos;
                            ++p[s];
                            if (first_real < 0) first_real = pos;
                            newest = std::max(newest, pos);
                        }
                    }
                }
            // an idle slot multiplies an X row by zero: the row of its own previous nonzero while that is still in the ring, else
            // the first row this unit reads anyway (both are there at whatever step the unit ends up running)
            for (int t = 0; t < 4; ++t)
                for (int s = 0; s < SLOTS; ++s) {
                    if (o[t][s] >= 0)
                        last_pos[s] = o[t][s];
                    else
                        o[t][s] = (last_pos[s] >= 0 && last_pos[s] >= lo) ? last_pos[s] : first_real;
                    out.val.push_back(v[t][s]);
                    out.off.push_back((uint16_t)(o[t][s] % ring));
                }
            out.step.push_back(k);
            out.earliest.push_back((int)(newest / SEG));
            out.flags.push_back(0);
        }
        bool done = true;
        for (int s = 0; s < SLOTS; ++s) done = done && p[s] == pe[s];
        if (done) {
            if (any_valid) {
                if (units == 0) { // rows without (remaining) nonzeros: a flush of its own, tied to this step
                    out.step.push_back(k);
                    out.earliest.push_back(k);
                    out.flags.push_back(2);
                }
                out.flags.back() |= 1;
                out.flush.push_back((int32_t)cur_row0);
            }
            loaded = false;
            bj += P.phases;
        }
    }
    if (loaded || bj < c.nblocks) {
        // blocks left over after the last step: only possible for rows without nonzeros beyond the swept range
        why = "row blocks left after the last sweep step";
        return false;
    }
    return true;
}

} // namespace

bool rails_sweep_plan_build(const rails_sweep_params &prm, int64_t m, int64_t ncols, const int64_t *rowptr, const int32_t *col,
                            const double *val, rails_sweep_plan &plan)
{
    plan = rails_sweep_plan();
    plan.p = prm;
    plan.m = m;
    plan.ncols = ncols;
    plan.nnz = rowptr[m];
    const int W = prm.waves, G = prm.groups, P = prm.phases, SEG = prm.seg_rows;
    if (W < 1 || G < 1 || G > 64 || P < 1 || prm.parts < 1 || prm.nseg < 3 || SEG < 8 || SEG % 8 || (int64_t)SEG * prm.nseg > 65536) {
        plan.why = "bad parameters";
        return false;
    }
    const int64_t R = (int64_t)G * W * SLOTS;
    plan.part_row0.resize(prm.parts + 1);
    for (int x = 0; x <= prm.parts; ++x) plan.part_row0[x] = m * x / prm.parts;
    plan.sweep0.assign(prm.parts, 0);
    plan.nsteps.assign(prm.parts, 1);
    const int64_t nprog = (int64_t)prm.parts * P * W;
    plan.hdr_off.assign(nprog, 0);
    plan.batch_off.assign(nprog, 0);
    plan.flush_off.assign(nprog, 0);
    // rows must list their columns in ascending order (the sweep consumes them in that order)
    for (int64_t i = 0; i < m; ++i)
        for (int64_t q = rowptr[i] + 1; q < rowptr[i + 1]; ++q)
            if (col[q] < col[q - 1]) {
                plan.why = "columns of a row are not sorted";
                return false;
            }
    int64_t staged = 0;
    std::vector<std::vector<GroupSchedule>> all(W, std::vector<GroupSchedule>(G));
    for (int x = 0; x < prm.parts; ++x) {
        Ctx c;
        c.prm = &prm;
        c.rowptr = rowptr;
        c.col = col;
        c.val = val;
        c.r0 = plan.part_row0[x];
        c.r1 = plan.part_row0[x + 1];
        c.R = R;
        c.nblocks = (c.r1 - c.r0 + R - 1) / R;
        int64_t cmin = INT64_MAX, cmax = -1;
        for (int64_t i = c.r0; i < c.r1; ++i)
            if (rowptr[i + 1] > rowptr[i]) {
                cmin = std::min<int64_t>(cmin, col[rowptr[i]]);
                cmax = std::max<int64_t>(cmax, col[rowptr[i + 1] - 1]);
            }
        if (cmax < 0) cmin = cmax = 0;
        c.sweep0 = cmin;
        // at least one step per block of a workgroup, so that blocks of rows without nonzeros still get their zeros written
        const int64_t by_blocks = (c.nblocks + P - 1) / P;
        // NSEG - 2 steps beyond the last X row, so that the last nonzeros are consumed at the pace of all the others
        c.nsteps = (int)std::max<int64_t>((cmax - cmin) / SEG + 1 + (prm.nseg - 2), by_blocks);
        plan.sweep0[x] = c.sweep0;
        plan.nsteps[x] = c.nsteps;
        staged += (int64_t)c.nsteps * SEG * P;
        for (int ph = 0; ph < P; ++ph) {
            for (int w = 0; w < W; ++w)
                for (int g = 0; g < G; ++g)
                    if (!schedule_group(c, ph, w, g, all[w][g], plan.why)) return false;
            // Level the work of the workgroup's waves step by step: all of them meet at a barrier every step, so a step costs what
            // its busiest wave costs (unlevelled: the sum over the steps of the busiest wave's units is 1.32 x the mean wave's).  The
            // schedule above runs a unit as late as its X rows allow; it may run earlier, down to the step in which the last of its
            // rows arrives, as long as the units of its group stay in order.  From the last step backwards, a wave above the step's
            // mean over the waves hands the first unit of a group to the step before.
            if (prm.level) {
                std::vector<std::vector<int>> load(W, std::vector<int>(c.nsteps, 0));
                for (int w = 0; w < W; ++w)
                    for (int g = 0; g < G; ++g)
                        for (size_t u = 0; u < all[w][g].step.size(); ++u)
                            if (!(all[w][g].flags[u] & 2)) ++load[w][all[w][g].step[u]];
                std::vector<size_t> head(G);
                for (int k = c.nsteps - 1; k >= 1; --k) {
                    int sum = 0;
                    for (int w = 0; w < W; ++w) sum += load[w][k];
                    const int target = (sum + W - 1) / W;
                    for (int w = 0; w < W; ++w) {
                        if (load[w][k] <= target) continue;
                        std::vector<GroupSchedule> &gs = all[w];
                        for (int g = 0; g < G; ++g) {
                            size_t u = 0;
                            while (u < gs[g].step.size() && gs[g].step[u] < k) ++u;
                            head[g] = u;
                        }
                        bool moved = true;
                        while (load[w][k] > target && moved) {
                            moved = false;
                            for (int g = 0; g < G && load[w][k] > target; ++g) {
                                const size_t u = head[g];
                                if (u >= gs[g].step.size() || gs[g].step[u] != k || (gs[g].flags[u] & 2) || gs[g].earliest[u] > k - 1) continue;
                                gs[g].step[u] = k - 1;
                                ++head[g];
                                --load[w][k];
                                ++load[w][k - 1];
                                moved = true;
                            }
                        }
                    }
                }
            }
            for (int w = 0; w < W; ++w) {
                const int64_t prog = ((int64_t)x * P + ph) * W + w;
                std::vector<GroupSchedule> &gs = all[w];
                // serialise: per step a record of RAILS_SWEEP_CODES 16-bit entries: [0] = n, then one entry per unit of four trips (or per
                // flush without trips): group | flush after << 6 | no trips << 7, in group order; the trips in the same order, 16
                // trips per batch: lane q of a slot holds trips q and q + 8 of the batch
                plan.hdr_off[prog] = (int64_t)plan.codes.size();
                plan.codes.resize(plan.codes.size() + (size_t)c.nsteps * RAILS_SWEEP_CODES, (uint16_t)0);
                uint16_t *h = plan.codes.data() + plan.hdr_off[prog];
                plan.batch_off[prog] = (int64_t)(plan.vals.size() / 256);
                plan.flush_off[prog] = (int64_t)plan.flush_rows.size();
                std::vector<size_t> up(G, 0), tp(G, 0), fp(G, 0); // per group: next unit / next unit with trips / next flush
                int64_t trip = 0;                                  // trips of this wave so far
                for (int k = 0; k < c.nsteps; ++k) {
                    int n = 0;
                    uint16_t *rec = h + (size_t)k * RAILS_SWEEP_CODES;
                    for (int g = 0; g < G; ++g)
                        while (up[g] < gs[g].step.size() && gs[g].step[up[g]] == k) {
                            const uint8_t fl = gs[g].flags[up[g]++];
                            if (n + 1 > RAILS_SWEEP_CODES - 1) {
                                plan.why = "more than 127 units of one wave in one step (step " + std::to_string(k) + " of " + std::to_string(c.nsteps) + ", part " + std::to_string(x) + ")";
                                return false;
                            }
                            rec[1 + n++] = (uint16_t)(g | ((fl & 1) ? 0x40 : 0) | ((fl & 2) ? 0x80 : 0));
                            if (!(fl & 2)) {
                                for (int t = 0; t < 4; ++t, ++trip) {
                                    const int64_t b = plan.batch_off[prog] + trip / 16;
                                    if ((size_t)(b + 1) * 256 > plan.vals.size()) {
                                        plan.vals.resize((size_t)(b + 1) * 256, 0.0);
                                        plan.offs.resize((size_t)(b + 1) * 256, 0);
                                    }
                                    // trip tt of the batch = unit tt / 4, quad lane tt % 4: lane (slot, quad lane) holds four values, one per
                                    // unit -- units 0, 1 in the first KiB of the batch, 2, 3 in the second (one 16-byte load each) -- and
                                    // four 16-bit ring rows (one 8-byte load)
                                    const int tt = (int)(trip % 16), unit = tt / 4, ql = tt % 4;
                                    for (int s = 0; s < SLOTS; ++s) {
                                        const size_t lane = (size_t)s * 4 + ql;
                                        plan.vals[(size_t)b * 256 + (size_t)(unit / 2) * 128 + lane * 2 + (unit % 2)] = gs[g].val[(tp[g] * 4 + t) * SLOTS + s];
                                        plan.offs[(size_t)b * 256 + lane * 4 + unit] = gs[g].off[(tp[g] * 4 + t) * SLOTS + s];
                                    }
                                }
                                ++tp[g];
                            }
                            if (fl & 1) plan.flush_rows.push_back(gs[g].flush[fp[g]++]);
                        }
                    rec[0] = (uint16_t)n;
                    plan.max_units_per_step = std::max(plan.max_units_per_step, n);
                }
                plan.trips += trip;
            }
        }
    }
    // spare batches at the very end: the kernel requests batches up to six ahead of the trips it runs
    plan.vals.resize(plan.vals.size() + 8 * 256, 0.0);
    plan.offs.resize(plan.offs.size() + 8 * 256, 0);
    plan.entries = plan.nnz;
    plan.efficiency = plan.trips ? (double)plan.nnz / ((double)SLOTS * (double)plan.trips) : 1.0;
    plan.staged_rows_per_row = m ? (double)staged / (double)m : 0.0;
    return true;
}

// ---- C ABI (include/rails_hip.h): host-side access to the schedule for tests and diagnostics ----
#include "rails_hip.h"

void rails_set_error(const char *fmt, ...);

extern "C" int rails_sweep_plan_create(int64_t m, int64_t ncols, const int64_t *rowptr, const int32_t *col, const double *val,
                                       const int *params, rails_sweep_plan **out)
{
    if (!rowptr || !out || m < 0 || (rowptr[m] > 0 && (!col || !val))) {
        rails_set_error("rails_sweep_plan_create: bad argument");
        return RAILS_EINVAL;
    }
    rails_sweep_params prm;
    if (params) {
        prm.waves = params[0];
        prm.groups = params[1];
        prm.seg_rows = params[2];
        prm.nseg = params[3];
        prm.parts = params[4];
        prm.phases = params[5];
        if (getenv("RAILS_SWEEP_LEVEL")) prm.level = atoi(getenv("RAILS_SWEEP_LEVEL"));
    }
    rails_sweep_plan *pl = new rails_sweep_plan();
    if (!rails_sweep_plan_build(prm, m, ncols, rowptr, col, val, *pl)) {
        rails_set_error("rails_sweep_plan_create: pattern does not fit the sweep scheme: %s", pl->why.c_str());
        delete pl;
        return RAILS_EINVAL;
    }
    *out = pl;
    return RAILS_OK;
}

extern "C" int rails_sweep_plan_destroy(rails_sweep_plan *pl)
{
    delete pl;
    return RAILS_OK;
}

extern "C" int rails_sweep_plan_info(const rails_sweep_plan *pl, int64_t *iinfo, double *dinfo)
{
    if (!pl || !iinfo || !dinfo) return RAILS_EINVAL;
    iinfo[0] = pl->p.waves;
    iinfo[1] = pl->p.groups;
    iinfo[2] = pl->p.seg_rows;
    iinfo[3] = pl->p.nseg;
    iinfo[4] = pl->p.parts;
    iinfo[5] = pl->p.phases;
    iinfo[6] = RAILS_SWEEP_CODES;
    iinfo[7] = pl->trips;
    iinfo[8] = pl->nnz;
    iinfo[9] = (int64_t)(pl->vals.size() / 256);
    iinfo[11] = SLOTS;
    iinfo[10] = pl->max_units_per_step;
    dinfo[0] = pl->efficiency;
    dinfo[1] = pl->staged_rows_per_row;
    return RAILS_OK;
}

extern "C" int rails_sweep_plan_array(const rails_sweep_plan *pl, int which, const void **ptr, int64_t *count)
{
    if (!pl || !ptr || !count) return RAILS_EINVAL;
    switch (which) {
    case 0: *ptr = pl->part_row0.data(); *count = (int64_t)pl->part_row0.size(); break;
    case 1: *ptr = pl->sweep0.data(); *count = (int64_t)pl->sweep0.size(); break;
    case 2: *ptr = pl->nsteps.data(); *count = (int64_t)pl->nsteps.size(); break;
    case 3: *ptr = pl->hdr_off.data(); *count = (int64_t)pl->hdr_off.size(); break;
    case 4: *ptr = pl->batch_off.data(); *count = (int64_t)pl->batch_off.size(); break;
    case 5: *ptr = pl->flush_off.data(); *count = (int64_t)pl->flush_off.size(); break;
    case 6: *ptr = pl->codes.data(); *count = (int64_t)pl->codes.size(); break;
    case 7: *ptr = pl->vals.data(); *count = (int64_t)pl->vals.size(); break;
    case 8: *ptr = pl->offs.data(); *count = (int64_t)pl->offs.size(); break;
    case 9: *ptr = pl->flush_rows.data(); *count = (int64_t)pl->flush_rows.size(); break;
    default: return RAILS_EINVAL;
    }
    return RAILS_OK;
}
