// sweep_plan.cpp -- builds the schedule the sweep SpMM kernel interprets (see sweep_plan.h).  Host code only.
#include "sweep_plan.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int SLOTS = 16; // rows of a group in one wave: a slot is a quad of lanes, each lane four of the row's 16 columns

// the schedule of one group: its entries (entry_trips trips each, or a flush without trips) in execution order ("unit" below = entry)
struct GroupSchedule {
    std::vector<int> step;        // per unit: the step it runs in
    std::vector<uint8_t> flags;   // per unit: 1 = the group's partial sums go to Y afterwards, 2 = no trips
    std::vector<int32_t> flush;   // first row of the wave's slot octet, per flush, in order
};

// The (value, ring row) pairs of one wave, in the order and layout in which the kernel consumes them: the groups of a wave are run
// group by group within a step and step by step (schedule_part), which is the order of the wave's entries in its records, so every
// trip goes straight to its place -- no per-group copy to be gathered later (a cold build is bound by the memory it touches).
// Trip tt of a batch of 16 = unit tt / 4, quad lane tt % 4: lane (slot, quad lane) holds four values, one per unit -- units 0, 1 in
// the first KiB of the batch, 2, 3 in the second (one 16-byte load each) -- and four 16-bit ring rows (one 8-byte load).
struct WaveStream {
    std::vector<double> vals;
    std::vector<uint16_t> offs;
    int64_t trip = 0;
    void emit(const double *v, const int64_t *o, int ring)
    {
        const int64_t b = trip / 16;
        if ((size_t)(b + 1) * 256 > vals.size()) {
            vals.resize((size_t)(b + 1) * 256, 0.0);
            offs.resize((size_t)(b + 1) * 256, 0);
        }
        const int tt = (int)(trip % 16), unit = tt / 4, ql = tt % 4;
        double *vb = vals.data() + (size_t)b * 256 + (size_t)(unit / 2) * 128 + (unit % 2);
        uint16_t *ob = offs.data() + (size_t)b * 256 + unit;
        for (int s = 0; s < SLOTS; ++s) {
            const size_t lane = (size_t)s * 4 + ql;
            vb[lane * 2] = v[s];
            ob[lane * 4] = (uint16_t)(o[s] % ring);
        }
        ++trip;
    }
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

// One group of a wave (SLOTS rows of every block of the wave's workgroup) on its way through the sweep.  Per step: look() says what
// has arrived and what cannot wait; run() forms the units (every slot takes up to four of its arrived nonzeros per unit, in column
// order) and moves on to the workgroup's next block when the rows are done.
struct GroupRun {
    const Ctx *c = nullptr;
    int phase = 0, wave = 0, g = 0;
    GroupSchedule out;
    WaveStream *stream = nullptr;  // the wave's stream of (value, ring row) pairs
    int64_t bj = 0;                // current block of this workgroup
    int64_t p[SLOTS], pe[SLOTS];   // next / end nonzero of each slot's row
    int64_t last_pos[SLOTS];
    bool loaded = false, any_valid = false;
    int64_t cur_row0 = 0, next_first = INT64_MAX;
    // what look() found for the current step
    int need = 0;                  // trips that have to run in this step
    int full = 0;                  // trips every slot with work left can fill (no idle slots: free of waste)
    int avail[SLOTS];              // arrived nonzeros per slot
    int64_t lo = 0, hi = 0;

    int64_t row0_of(int64_t j) const { return c->r0 + j * c->R + (int64_t)g * c->prm->waves * SLOTS + (int64_t)wave * SLOTS; }

    void start(const Ctx &ctx, int ph, int w, int group)
    {
        c = &ctx;
        phase = ph;
        wave = w;
        g = group;
        out = GroupSchedule();
        bj = ph;
        loaded = false;
        need = full = 0;
    }

    void load_block(int64_t j)
    {
        cur_row0 = row0_of(j);
        any_valid = false;
        for (int s = 0; s < SLOTS; ++s) {
            const int64_t row = cur_row0 + s;
            last_pos[s] = -1;
            if (row < c->r1 && row < c->r0 + (j + 1) * c->R) {
                p[s] = c->rowptr[row];
                pe[s] = c->rowptr[row + 1];
                any_valid = true;
            } else
                p[s] = pe[s] = 0;
        }
        loaded = true;
    }

    // first sweep segment of the first nonzero of block j for this group (INT64_MAX: none)
    int64_t first_seg(int64_t j) const
    {
        if (j >= c->nblocks) return INT64_MAX;
        const int64_t row0 = row0_of(j);
        int64_t best = INT64_MAX;
        for (int s = 0; s < SLOTS; ++s) {
            const int64_t row = row0 + s;
            if (row < c->r1 && row < c->r0 + (j + 1) * c->R && c->rowptr[row + 1] > c->rowptr[row])
                best = std::min<int64_t>(best, ((int64_t)c->col[c->rowptr[row]] - c->sweep0) / c->prm->seg_rows);
        }
        return best;
    }

    bool look(int k, std::string &why)
    {
        const rails_sweep_params &P = *c->prm;
        const int SEG = P.seg_rows, NSEG = P.nseg;
        need = full = 0;
        for (int s = 0; s < SLOTS; ++s) avail[s] = 0;
        if (!loaded) {
            if (bj >= c->nblocks) return true;
            load_block(bj);
            // the next block WITH nonzeros bounds how long this one may take
            next_first = INT64_MAX;
            for (int64_t j = bj + P.phases; j < c->nblocks && next_first == INT64_MAX; j += P.phases) next_first = first_seg(j);
        }
        hi = (int64_t)(k + 1) * SEG;
        lo = (int64_t)(k - NSEG + 1 + P.ahead) * SEG;                                 // readable: [lo, hi)
        const int64_t lo_next = (k == c->nsteps - 1) ? INT64_MAX : lo + SEG;          // gone after this step
        int forced = 0, remaining = 0, ready = INT_MAX;
        bool all_available = true;
        for (int s = 0; s < SLOTS; ++s) {
            int f = 0, av = 0;
            for (int64_t q = p[s]; q < pe[s]; ++q) {
                const int64_t pos = (int64_t)c->col[q] - c->sweep0;
                if (pos < lo) {
                    why = "a nonzero lost its X row before it was consumed (columns of a row not sorted?)";
                    return false;
                }
                if (pos >= hi) break;
                ++av;
                if (pos < lo_next) ++f;
            }
            avail[s] = av;
            forced = std::max(forced, f);
            if (pe[s] > p[s]) ready = std::min(ready, av);
            remaining = std::max<int>(remaining, (int)(pe[s] - p[s]));
            if (pe[s] > p[s] && (int64_t)c->col[pe[s] - 1] - c->sweep0 >= hi) all_available = false;
        }
        // the block has to be done before the first X row of the workgroup's next block leaves the ring
        const bool must_finish = next_first != INT64_MAX && (int64_t)k >= next_first + NSEG - 2 - P.ahead;
        need = forced;
        const int UT = c->prm->entry_trips;
        full = ready == INT_MAX ? 0 : ready / UT * UT;
        if (must_finish || remaining == 0) {
            if (!all_available) {
                why = "the column window of a row block is wider than (phases - 1) blocks";
                return false;
            }
            need = remaining;
        }
        return true;
    }

    // nonzeros unit number n of this step would consume (n = 0: the first)
    int fill_of_unit(int n) const
    {
        const int UT = c->prm->entry_trips;
        int f = 0;
        for (int s = 0; s < SLOTS; ++s) f += std::min(std::max(avail[s] - UT * n, 0), UT);
        return f;
    }

    bool run(int k, int units, std::string &why)
    {
        if (!loaded) return true;
        const rails_sweep_params &P = *c->prm;
        const int SEG = P.seg_rows, NSEG = P.nseg;
        const int ring = SEG * NSEG, UT = P.entry_trips;
        if (units > 127) {
            why = "more than 127 entries of one group in one step (hundreds of nonzeros of one row inside one ring of X rows)";
            return false;
        }
        for (int u = 0; u < units; ++u) {
            double v[4][SLOTS];
            int64_t o[4][SLOTS];
            int64_t first_real = -1;
            for (int t = 0; t < UT; ++t)
                for (int s = 0; s < SLOTS; ++s) {
                    v[t][s] = 0.0;
                    o[t][s] = -1;
                    if (p[s] < pe[s]) {
                        const int64_t pos = (int64_t)c->col[p[s]] - c->sweep0;
                        if (pos < hi) {
                            v[t][s] = c->val[p[s]];
                            o[t][s] = pos;
                            ++p[s];
                            if (first_real < 0) first_real = pos;
                        }
                    }
                }
            if (first_real < 0) break; // nothing left that has arrived
            // An idle slot multiplies an X row by zero.  Which row matters for LDS: a `ds_read_b128` serves 16 lanes (four slots) per
            // cycle -- lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), i.e. slots {0, 3, 5, 6},
            // {1, 2, 4, 7} and the same + 8 -- and the two slots of a group that read the same half of their ring rows first (slot parity:
            // 0 with 6, 3 with 5, 2 with 4, 1 with 7) collide when their rows have the same parity: one extra cycle for the group
            // (measured 1.75 cycles per group = 1 + P(either pair collides) = 1 + 3/4).  An idle slot that reads its PARTNER's row asks
            // for the very same words -- a broadcast, no collision: with 37 % of the slots idle the expectation drops to ~1.36.
            // (RAILS_SWEEP_IDLE_FILL=0: the row of the slot's own previous nonzero while that is still in the ring, as before.)
            static const int partner_fill = getenv("RAILS_SWEEP_IDLE_FILL") ? atoi(getenv("RAILS_SWEEP_IDLE_FILL")) : 1;
            static const int PARTNER[16] = {6, 7, 4, 5, 2, 3, 0, 1, 14, 15, 12, 13, 10, 11, 8, 9};
            for (int t = 0; t < UT; ++t) {
                bool real[SLOTS];
                for (int s = 0; s < SLOTS; ++s) real[s] = o[t][s] >= 0;
                for (int s = 0; s < SLOTS; ++s) {
                    if (real[s]) {
                        last_pos[s] = o[t][s];
                        continue;
                    }
                    const int q = partner_fill == 2 ? (s ^ 2) : PARTNER[s]; // (2: experiment -- four consecutive slots served together)
                    if (partner_fill && (real[q] || q < s))
                        o[t][s] = o[t][q]; // the partner's row (its own if it has work, else what it was given: q < s is assigned already)
                    else
                        o[t][s] = (last_pos[s] >= 0 && last_pos[s] >= lo) ? last_pos[s] : first_real;
                }
            }
            for (int t = 0; t < UT; ++t) stream->emit(v[t], o[t], ring);
            out.step.push_back(k);
            out.flags.push_back(0);
        }
        bool done = true;
        for (int s = 0; s < SLOTS; ++s) done = done && p[s] == pe[s];
        if (done) {
            if (any_valid) {
                if (out.step.empty() || out.step.back() != k || (out.flags.back() & 1)) { // nothing ran in this step for this block: a flush of its own
                    out.step.push_back(k);
                    out.flags.push_back(2);
                }
                out.flags.back() |= 1;
                out.flush.push_back((int32_t)cur_row0);
            }
            loaded = false;
            bj += P.phases;
        }
        return true;
    }

    bool left_over() const { return loaded || bj < c->nblocks; }
};

// The schedule of one workgroup (W waves x G groups) of a part.  All its waves meet at a barrier every step, so a step costs what its
// busiest wave costs.  Per step: every group runs what cannot wait (X rows about to leave the ring, the end of a block) and what is
// free of waste (units without idle slots); the busiest wave's number of units is then what the step costs anyway, and every other wave
// fills up to it with the units that consume the most of what has arrived -- work that would otherwise be forced on it in a later step,
// where it might be the busiest one.  (Every unit run as late as possible: the sum over the steps of the busiest wave's units is 1.37 x
// the mean wave's.)
// level 2 (the default) also paces the `phases` workgroups of a part's column chunk, which read the same X rows: a workgroup whose
// steps have cost less time so far than another's sits the difference out at the start of the step (pause[phase * nsteps + step], in
// quarters of a unit's time), so that an X row fetched by one of them is still in the L2 when the others ask for it (left alone they drift apart by more
// steps than the L2 holds: 3.9 GB fetched per product instead of 1.7 GB).  Costs no time: the product takes as long as its slowest
// workgroup either way.  runs[phase * W + wave][group].
bool schedule_part(const Ctx &c, std::vector<std::vector<GroupRun>> &runs, std::vector<WaveStream> &streams, std::vector<int> &pause, std::string &why)
{
    const rails_sweep_params &P = *c.prm;
    const int W = P.waves, G = P.groups, NW = P.phases * W, UT = P.entry_trips;
    const int cap = RAILS_SWEEP_CODES - 2;
    // room for a wave's stream: its rows' nonzeros at trips about half full (growing by doubling copies and faults in twice the memory)
    const double per_row = c.r1 > c.r0 ? (double)(c.rowptr[c.r1] - c.rowptr[c.r0]) / (double)(c.r1 - c.r0) : 0.0;
    const size_t guess = (size_t)((double)(c.r1 - c.r0) / NW * per_row * 2.0) + 4096;
    for (int x = 0; x < NW; ++x) {
        streams[x] = WaveStream();
        streams[x].vals.reserve(guess);
        streams[x].offs.reserve(guess);
        for (int g = 0; g < G; ++g) {
            runs[x][g].start(c, x / W, x % W, g);
            runs[x][g].stream = &streams[x];
        }
    }
    std::vector<int> units((size_t)NW * G), load(NW);
    pause.assign((size_t)P.phases * c.nsteps, 0);
    std::vector<int64_t> spent(P.phases, 0); // units of the busiest wave, summed over the steps so far: the workgroup's time
    for (int k = 0; k < c.nsteps; ++k) {
        for (int x = 0; x < NW; ++x) {
            load[x] = 0;
            for (int g = 0; g < G; ++g) {
                GroupRun &r = runs[x][g];
                if (!r.look(k, why)) return false;
                units[(size_t)x * G + g] = (std::max(r.need, r.full) + UT - 1) / UT;
                load[x] += units[(size_t)x * G + g];
            }
        }
        // the busiest wave of a workgroup sets what the step costs that workgroup; the other waves fill up to it
        int64_t furthest = 0;
        for (int ph = 0; ph < P.phases; ++ph) {
            int level = 0;
            for (int w = 0; w < W; ++w) level = std::max(level, load[ph * W + w]);
            for (int x = ph * W; x < (ph + 1) * W; ++x)
                while (P.level && load[x] < level) {
                    int best = -1, best_fill = P.level_min_fill * UT / 4 - 1;
                    for (int g = 0; g < G; ++g) {
                        const int f = runs[x][g].fill_of_unit(units[(size_t)x * G + g]);
                        if (f > best_fill) {
                            best = g;
                            best_fill = f;
                        }
                    }
                    if (best < 0) break;
                    ++units[(size_t)x * G + best];
                    ++load[x];
                }
            // what the step costs the workgroup, in quarters of a unit's time: measured, 1.24 us for a step without units (the time its X rows
            // take to arrive), 2.2 us for 5.8 units at 0.26 us each
            spent[ph] += std::max(19, UT * level + 11);
            furthest = std::max(furthest, spent[ph]);
        }
        // a workgroup whose steps have cost less than another's so far is ahead of it in time: it sits out the difference (beyond the slack)
        for (int ph = 0; ph < P.phases; ++ph) {
            const int64_t ahead = P.level >= 2 ? furthest - 4 * P.level_slack - spent[ph] : 0;
            pause[(size_t)ph * c.nsteps + k] = (int)std::min<int64_t>(std::max<int64_t>(ahead, 0), 0xffff);
            spent[ph] += pause[(size_t)ph * c.nsteps + k];
        }
        for (int x = 0; x < NW; ++x) {
            if (load[x] + G > cap) {
                why = "more units of one wave in one step than a step's record holds";
                return false;
            }
            for (int g = 0; g < G; ++g)
                if (!runs[x][g].run(k, units[(size_t)x * G + g], why)) return false;
        }
    }
    for (int x = 0; x < NW; ++x)
        for (int g = 0; g < G; ++g)
            if (runs[x][g].left_over()) {
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
    if ((prm.entry_trips != 4 && prm.entry_trips != 2) || W < 1 || G < 1 || G > 31 || P < 1 || prm.parts < 1 || prm.ahead < 1 || prm.nseg < prm.ahead + 2 || SEG < 8 || SEG % 8 || (int64_t)SEG * prm.nseg > 65536) {
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
    // The parts are independent: each is planned by a thread of its own into a plan of its own, and the pieces are joined afterwards.
    const std::vector<int64_t> part_row0 = plan.part_row0;
    auto plan_part = [&](int x, rails_sweep_plan &plan, std::vector<WaveStream> &streams, int64_t &staged) -> bool {
        std::vector<std::vector<GroupRun>> runs((size_t)P * W, std::vector<GroupRun>(G));
        std::vector<int> pause;
        streams.assign((size_t)P * W, WaveStream());
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
        // NSEG - 1 - ahead steps beyond the last X row, so that the last nonzeros are consumed at the pace of all the others
        c.nsteps = (int)std::max<int64_t>((cmax - cmin) / SEG + 1 + (prm.nseg - 1 - prm.ahead), by_blocks);
        plan.sweep0[x] = c.sweep0;
        plan.nsteps[x] = c.nsteps;
        staged += (int64_t)c.nsteps * SEG * P;
        if (!schedule_part(c, runs, streams, pause, plan.why)) return false; // (this part)
        int64_t batches = 0; // batches of 16 trips of the waves before this one
        for (int ph = 0; ph < P; ++ph) {
            for (int w = 0; w < W; ++w) {
                const int64_t prog = ((int64_t)x * P + ph) * W + w;
                std::vector<GroupRun> &gr = runs[(size_t)ph * W + w];
                // serialise: per step a record of RAILS_SWEEP_CODES 32-bit entries (sweep_plan.h): a header with the number of entries, then
                // one entry per unit (or per flush without trips), in group order; the trips are in the wave's stream in the same order
                plan.hdr_off[prog] = (int64_t)plan.codes.size();
                plan.codes.resize(plan.codes.size() + (size_t)c.nsteps * RAILS_SWEEP_CODES, 0u);
                uint32_t *h = plan.codes.data() + plan.hdr_off[prog];
                plan.batch_off[prog] = batches; // (within this part; the streams are joined afterwards)
                plan.flush_off[prog] = (int64_t)plan.flush_rows.size();
                std::vector<size_t> up(G, 0), fp(G, 0); // per group: next unit / next flush
                int64_t trip = 0;                       // trips of this wave so far
                for (int k = 0; k < c.nsteps; ++k) {
                    int n = 0;
                    uint32_t *rec = h + (size_t)k * RAILS_SWEEP_CODES;
                    for (int g = 0; g < G; ++g)
                        while (up[g] < gr[g].out.step.size() && gr[g].out.step[up[g]] == k) {
                            const uint8_t fl = gr[g].out.flags[up[g]++];
                            if (n + 1 > RAILS_SWEEP_CODES - 1) {
                                plan.why = "more than 63 entries of one wave in one step (step " + std::to_string(k) + " of " + std::to_string(c.nsteps) + ", part " + std::to_string(x) + ")";
                                return false; // (this part)
                            }
                            rec[1 + n++] = (uint32_t)(g * 8) | ((fl & 1) ? RAILS_SWEEP_FLUSH : 0u) | ((fl & 2) ? RAILS_SWEEP_NO_TRIPS : 0u);
                            if (!(fl & 2)) trip += prm.entry_trips;
                            if (fl & 1) {
                                // the rows the flush writes: sixteen from this one on, as a multiple of 16 from the part's first row in the entry
                                const int64_t row = gr[g].out.flush[fp[g]++], rel = (row - c.r0) / 16;
                                if ((row - c.r0) % 16 || rel >= (int64_t)1 << 20) {
                                    plan.why = "a part has more rows than a record entry can name (16M)";
                                    return false; // (this part)
                                }
                                rec[n] |= (uint32_t)rel << RAILS_SWEEP_ROW_SHIFT;
                                plan.flush_rows.push_back((int32_t)row);
                            }
                        }
                    // what the kernel branches on after a unit: anything but "go on with the next entry"
                    rec[0] = (uint32_t)n | ((uint32_t)pause[(size_t)ph * c.nsteps + k] << 16);
                    if (n) rec[n] |= RAILS_SWEEP_LAST;
                    for (int e = 1; e <= n; ++e)
                        if (rec[e] & RAILS_SWEEP_NO_TRIPS) rec[e - 1] |= RAILS_SWEEP_NEXT_NO_TRIPS;
                    plan.max_units_per_step = std::max(plan.max_units_per_step, n);
                }
                if (trip != streams[(size_t)ph * W + w].trip) {
                    plan.why = "internal: the records of a wave do not add up to its stream";
                    return false; // (this part)
                }
                plan.trips += trip;
                batches += (trip + 15) / 16;
            }
        }
        return true;
    };
    std::vector<rails_sweep_plan> piece(prm.parts);
    std::vector<std::vector<WaveStream>> piece_streams(prm.parts);
    std::vector<int64_t> piece_staged(prm.parts, 0);
    std::vector<char> piece_ok(prm.parts, 0);
    {
        const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)prm.parts, (int64_t)std::thread::hardware_concurrency(), 8, plan.nnz / 200000 + 1}));
        std::atomic<int> next(0);
        auto work = [&]() {
            for (int x = next++; x < prm.parts; x = next++) {
                rails_sweep_plan &q = piece[x];
                q.part_row0 = part_row0;
                q.sweep0.assign(prm.parts, 0);
                q.nsteps.assign(prm.parts, 1);
                q.hdr_off.assign(nprog, 0);
                q.batch_off.assign(nprog, 0);
                q.flush_off.assign(nprog, 0);
                piece_ok[x] = plan_part(x, q, piece_streams[x], piece_staged[x]) ? 1 : 0;
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    }
    int64_t staged = 0;
    {
        size_t nc = 0, nv = 0, nf = 0;
        for (int x = 0; x < prm.parts; ++x) {
            nc += piece[x].codes.size(), nf += piece[x].flush_rows.size();
            for (WaveStream const &ws : piece_streams[x]) nv += (size_t)((ws.trip + 15) / 16) * 256;
        }
        plan.codes.reserve(nc);
        plan.vals.reserve(nv + 8 * 256);
        plan.offs.reserve(nv + 8 * 256);
        plan.flush_rows.reserve(nf);
    }
    for (int x = 0; x < prm.parts; ++x) {
        rails_sweep_plan &q = piece[x];
        if (!piece_ok[x]) {
            plan.why = q.why;
            return false;
        }
        const int64_t codes0 = (int64_t)plan.codes.size(), batch0 = (int64_t)(plan.vals.size() / 256), flush0 = (int64_t)plan.flush_rows.size();
        for (int64_t prog = (int64_t)x * P * W; prog < (int64_t)(x + 1) * P * W; ++prog) {
            plan.hdr_off[prog] = codes0 + q.hdr_off[prog];
            plan.batch_off[prog] = batch0 + q.batch_off[prog];
            plan.flush_off[prog] = flush0 + q.flush_off[prog];
        }
        plan.sweep0[x] = q.sweep0[x];
        plan.nsteps[x] = q.nsteps[x];
        plan.codes.insert(plan.codes.end(), q.codes.begin(), q.codes.end());
        for (WaveStream &ws : piece_streams[x]) { // whole batches: a wave's next batch starts at a batch boundary
            const size_t n = (size_t)((ws.trip + 15) / 16) * 256;
            plan.vals.insert(plan.vals.end(), ws.vals.begin(), ws.vals.begin() + n);
            plan.offs.insert(plan.offs.end(), ws.offs.begin(), ws.offs.begin() + n);
            ws = WaveStream();
        }
        plan.flush_rows.insert(plan.flush_rows.end(), q.flush_rows.begin(), q.flush_rows.end());
        plan.trips += q.trips;
        plan.max_units_per_step = std::max(plan.max_units_per_step, q.max_units_per_step);
        staged += piece_staged[x];
        q = rails_sweep_plan();
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
        prm.ahead = params[6];
        if (params[7] == 2 || params[7] == 4) prm.entry_trips = params[7];
        if (getenv("RAILS_SWEEP_LEVEL")) prm.level = atoi(getenv("RAILS_SWEEP_LEVEL")); // experiments: 0 = no levelling of the waves
        if (getenv("RAILS_SWEEP_MIN_FILL")) prm.level_min_fill = atoi(getenv("RAILS_SWEEP_MIN_FILL"));
        if (getenv("RAILS_SWEEP_SLACK")) prm.level_slack = atoi(getenv("RAILS_SWEEP_SLACK"));
    }
    if ((!params || params[7] == 0) && getenv("RAILS_SWEEP_ENTRY_TRIPS")) prm.entry_trips = atoi(getenv("RAILS_SWEEP_ENTRY_TRIPS")) == 4 ? 4 : 2;
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
    iinfo[12] = pl->p.ahead;
    iinfo[13] = pl->p.entry_trips;
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
