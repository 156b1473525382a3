// SubspaceWrappers.hpp -- a second HIP back end for Solver<Matrix, MultiVector, DenseMatrix>: multivectors held as COORDINATES in
// an orthonormal basis P that lives on the device.
//
// Every vector the RAILS loop ever forms lies in span[B, random start vectors, A*(earlier vectors)].  This back end keeps ONE
// orthonormal basis P (m x dim, row-partitioned device panel) of that span and represents every multivector by its dim x n
// coefficient matrix on the host (replicated on every rank).  Then
//   * dot, norm, axpy, scale, `* DenseMatrix`, views, orthogonalize() -- everything the solver and its residual Lanczos do
//     between two operator applications -- are small host operations on coefficients (exact images of the reference's
//     m-dimensional operations, because P is orthonormal);
//   * device work happens only where NEW directions enter: `A * W` (materialise W = P*Wc, CSR SpMM straight into P's tail,
//     orthonormalise the tail against P: block CGS2 + CholQR2, coefficients by bookkeeping) and `random()` (fill, project);
//   * after a restart the live coefficient matrices span fewer directions than P holds: compress() re-bases P on an
//     orthonormal basis of their column space (pivoted Householder QR on the host, one panel GEMM on the device).
// Per RAILS trip that is 4 passes over the m x dim basis (materialise W, first Gram, update + second Gram fused, second update; the Lanczos start vector of
// the next trip rides in the A*W block) instead of the 21 passes over [AV V B] of the fused Lanczos kernel plus the projection /
// orthogonalisation passes of the direct back end (HipWrappers.hpp), and 5 all-reduces instead of ~31.
//
// The classes satisfy the same duck-typed contract as HipMultiVectorWrapper / HipOperatorWrapper, so the
// unmodified solver template -- the reference's member-by-member sequence, src/LyapunovSolver.hpp:100-482 -- runs on them.
// A mass matrix is one more SubspaceOperator (M * W is absorbed like A * W); a warm start absorbs the caller's V.
#ifndef RAILS_SUBSPACEWRAPPERS_HPP
#define RAILS_SUBSPACEWRAPPERS_HPP

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <vector>

#include "rails/HipWrappers.hpp"

namespace rails
{

// coefficient block: column-major, ld rows of capacity (rows past the basis dimension are zero), ncap columns
struct CoefStore {
    std::vector<double> c;
    int ld = 0, ncap = 0;
    bool in_basis = true; // false: a plain small replicated matrix with ld rows (B'W and friends)
    CoefStore(int ld_, int ncap_, bool in_basis_) : c((size_t)std::max(ld_, 1) * std::max(ncap_, 1), 0.0), ld(ld_), ncap(std::max(ncap_, 1)), in_basis(in_basis_) {}
    double *col(int j) { return c.data() + (size_t)j * ld; }
    const double *col(int j) const { return c.data() + (size_t)j * ld; }
};

class SubspaceBasis
{
public:
    rails_ctx *ctx;
    int64_t m_local, m_global;
    HipMultiVectorWrapper P, P2; // basis panel and the ping-pong panel used by compress()
    int dim = 0;
    int row_cap; // leading dimension of every coefficient store
    std::vector<std::weak_ptr<CoefStore>> live;
    long n_absorb = 0, n_absorb_cols = 0, n_single = 0, n_compress = 0, n_materialise = 0, n_dropped = 0, n_prefetched = 0, n_second_round = 0, n_delicate = 0, n_reprojected = 0;
    bool failed = false;
    long n_replaced = 0; // columns of a multivector that lay in the span of their predecessors and were replaced by random directions (orthogonalize)
    // wall-clock split (seconds); with RAILS_SUBSPACE_PROFILE=1 the device is synchronised around every part so that the numbers
    // are those of the part itself
    double t_materialise = 0, t_absorb = 0, t_qr = 0, t_rotate = 0, t_recoef = 0;
    bool profile_sync = getenv("RAILS_SUBSPACE_PROFILE") != nullptr;
    bool trace = getenv("RAILS_SUBSPACE_TRACE") != nullptr;
    double reorth_survival = getenv("RAILS_SUBSPACE_REORTH") ? atof(getenv("RAILS_SUBSPACE_REORTH")) : 0.5;
    // RAILS_SUBSPACE_SLOW_MS=x (with RAILS_SUBSPACE_PROFILE): one line on stderr for every part that took longer than x ms
    double slow_ms = getenv("RAILS_SUBSPACE_SLOW_MS") ? atof(getenv("RAILS_SUBSPACE_SLOW_MS")) : 0.0;
    struct Tick {
        SubspaceBasis *b;
        double *acc;
        const char *what;
        std::chrono::steady_clock::time_point t0;
        Tick(SubspaceBasis *bb, double *a, const char *w = "") : b(bb), acc(a), what(w)
        {
            if (b->profile_sync) rails_ctx_sync(b->ctx);
            t0 = std::chrono::steady_clock::now();
        }
        ~Tick()
        {
            if (b->profile_sync) rails_ctx_sync(b->ctx);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            *acc += dt;
            if (b->slow_ms > 0.0 && dt * 1e3 > b->slow_ms)
                std::cerr << "[subspace] " << what << ": " << dt * 1e3 << " ms (dim " << b->dim << ", absorbs so far " << b->n_absorb << ")" << std::endl;
        }
    };
    // The next 1-column random() of the solver (the Lanczos start vector of the coming trip, src/LyapunovSolver.hpp:374) is
    // drawn ahead of time and expressed in the basis together with the A*W block it follows: one more column in a block
    // projection instead of two passes over the basis of its own.  The RNG stream it consumes is the one that random() call
    // would have consumed (streams are handed out in call order and nothing else draws in between).
    std::shared_ptr<CoefStore> cached_random;
    bool cached_valid = false;
    bool prefetch_random = true;

    // ---- overlapped block orthogonalisation ------------------------------------------------------------------------------------
    // After the first projection round of a block the host knows the block's coordinates to ~1e-12 (first-round coefficients, Gram
    // matrix by Pythagoras): enough to go on with the projected solve and the residual Lanczos run, which only steer the iteration.  The
    // second round and the CholQR of the block run on the device meanwhile (the deferred entry points of rails_hip.h: no host decision
    // sits in between); when the basis is next needed -- the coming A*W -- the small matrices they produced are read back and every live
    // coefficient store is re-based from the predicted new basis columns to the real ones (a w x w and a dim x w map, exact).
    // RAILS_SUBSPACE_OVERLAP=0 switches it off (every block then waits for its own orthogonalisation, as in round 1).
    bool overlap = !(getenv("RAILS_SUBSPACE_OVERLAP") && atoi(getenv("RAILS_SUBSPACE_OVERLAP")) == 0);
    double overlap_min_survival = 1e-4; // below: rounding in the Pythagorean Gram matrix (eps / survival) is no longer negligible
    long n_overlapped = 0;
    struct Pending {
        bool active = false;
        int dim0 = 0, w = 0, w2 = 0;
        std::vector<double> Rfp;    // predicted R factor (w x w, upper): coordinates of the block along the new basis columns
        std::vector<double> g0diag; // squared lengths of the block's columns before any projection
    } pending;
    enum { SLOT_C2 = 0, SLOT_G = 1, SLOT_M1 = 2, SLOT_G2 = 3, SLOT_M2 = 4, N_SLOTS = 5 };

    SubspaceBasis(rails_ctx *c, int64_t ml, int64_t mg, int rows) : ctx(c), m_local(ml), m_global(mg), P(ml, std::max(rows, 16), c), row_cap(std::max(rows, 16))
    {
        P.set_global_rows(mg);
        P.resize(0);
    }

    // the ping-pong panel of compress(), allocated up front (same capacity as P) where a multi-gigabyte device allocation in the
    // middle of a solve would hurt: hipMalloc of 3 GB takes 1-70 ms on the shared boxes
    void preallocate_compress_panel()
    {
        const size_t cap = (size_t)P.capacity();
        hip_ok(rails_ctx_reserve_staging(ctx, cap * cap / 2 * sizeof(double)), "rails_ctx_reserve_staging"); // the rotation's Q, Gram results
        if (P2.N() >= 0 && P2.capacity() >= P.capacity()) return;
        P2 = HipMultiVectorWrapper(m_local, P.capacity(), ctx);
        P2.set_global_rows(m_global);
        // one rotation-shaped product now (zero coefficients: P2 is scratch until the first compress()): whatever the GEMM path loads
        // or selects on its first call, it does here and not at the first restart of a solve
        const int kw = std::min(256, (int)P.capacity());
        if (kw >= 64 && rails_ctx_library_gemm_ready(ctx)) { // (the hand-written kernels are loaded with the library)
            std::vector<double> Z((size_t)kw * kw, 0.0);
            const int keep = P.N();
            P.resize(kw);
            P2.resize(kw);
            hip_ok(rails_panel_fill(ctx, P.panel(), keep, kw - keep, 0.0), "rails_panel_fill");
            hip_ok(rails_panel_gemm_wide(ctx, 1.0, P.panel(), 0, kw, Z.data(), kw, kw, 0.0, P2.panel(), 0), "rails_panel_gemm_wide");
            P.resize(keep);
        }
        P2.resize(0);
    }

    std::shared_ptr<CoefStore> new_store(int ncap, bool in_basis, int rows = 0)
    {
        auto s = std::make_shared<CoefStore>(in_basis ? row_cap : rows, ncap, in_basis);
        if (in_basis) live.push_back(s);
        return s;
    }

    void ensure_rows(int need)
    {
        if (need <= row_cap) return;
        int ncap = std::max(need + 64, row_cap + row_cap / 2);
        for (auto &w : live)
            if (auto s = w.lock()) {
                std::vector<double> nc((size_t)ncap * s->ncap, 0.0);
                for (int j = 0; j < s->ncap; ++j) memcpy(nc.data() + (size_t)j * ncap, s->col(j), sizeof(double) * dim);
                s->c.swap(nc);
                s->ld = ncap;
            }
        row_cap = ncap;
    }

    // P(:, 0:dim) * C  (C host, dim x n, ldc) -> device multivector m x n.  The result aliases a scratch panel of the basis that the
    // next materialise() call overwrites (no device allocation per call); copy() it to keep it.
    HipMultiVectorWrapper scratch;
    HipMultiVectorWrapper materialise(const double *C, int ldc, int n)
    {
        if (scratch.N() < 0 || scratch.capacity() < std::max(n, 1)) {
            scratch = HipMultiVectorWrapper(m_local, std::max(n, 16), ctx);
            scratch.set_global_rows(m_global);
        }
        if (!resolve_pending()) return scratch;
        scratch.resize(n);
        HipMultiVectorWrapper out;
        out = scratch; // shares the panel
        n_materialise++;
        Tick tick(this, &t_materialise, "materialise");
        if (n <= 0) return out;
        if (dim == 0) {
            out = 0.0;
            return out;
        }
        for (int j0 = 0; j0 < n; j0 += 128) { // 128 output columns per launch: the faster tile shape of k_panel_gemm
            int nc = std::min(128, n - j0);
            if (!hip_ok(rails_panel_gemm(ctx, 1.0, P.panel(), 0, dim, C + (size_t)j0 * ldc, ldc, nc, 0.0, out.panel(), j0), "rails_panel_gemm")) failed = true;
        }
        return out;
    }

    // room for w more columns behind the basis; returns the first tail column
    int tail(int w)
    {
        if (dim + w > P.capacity()) {
            P.resize(dim); // reserve keeps the columns in use
            P.resize(dim + w + 64);
        }
        P.resize(dim);
        ensure_rows(dim + w);
        return dim;
    }

    // The w columns X sitting in P's tail [dim, dim+w) are expressed in the basis: the part of X outside span(P) becomes new
    // orthonormal basis columns, coef (row_cap x w, zero-initialised by the caller, ld = row_cap) receives the coordinates of X in
    // the extended basis.  Block CGS2 against P, CholQR2 inside the block; the representation is X = P_old (C1 + C2) + Q (R2 R1).
    // RAILS_SUBSPACE_VERIFY=1 (diagnostics, tests): every synchronous block is checked after the fact -- the largest column of
    // X - P * coef relative to the column of X, and the largest entry of P'P - I, are kept in verify_repr / verify_orth.
    bool verify = getenv("RAILS_SUBSPACE_VERIFY") != nullptr;
    double verify_repr = 0.0, verify_orth = 0.0;
    HipMultiVectorWrapper Xv;
    bool absorb_tail(int w, double *coef)
    {
        if (!verify || w <= 0) return absorb_tail_impl(w, coef);
        if (!resolve_pending()) return false;
        if (Xv.N() < 0 || Xv.capacity() < w) Xv = HipMultiVectorWrapper(m_local, std::max(w, 64), ctx);
        Xv.resize(w);
        if (!hip_ok(rails_panel_copy(ctx, P.panel(), dim, w, Xv.panel(), 0), "rails_panel_copy")) return fail();
        std::vector<double> n0(w * w), n1(w * w);
        if (!hip_ok(rails_gram(ctx, Xv.panel(), 0, w, Xv.panel(), 0, w, n0.data(), w), "rails_gram")) return fail();
        const int dim_before = dim;
        const bool ok = absorb_tail_impl(w, coef);
        if (!ok || pending.active) return ok;
        if (!hip_ok(rails_panel_gemm(ctx, -1.0, P.panel(), 0, dim, coef, row_cap, w, 1.0, Xv.panel(), 0), "rails_panel_gemm")) return fail();
        if (!hip_ok(rails_gram(ctx, Xv.panel(), 0, w, Xv.panel(), 0, w, n1.data(), w), "rails_gram")) return fail();
        double worst = 0.0;
        for (int j = 0; j < w; ++j)
            if (n0[j + (size_t)j * w] > 0.0) worst = std::max(worst, std::sqrt(n1[j + (size_t)j * w] / n0[j + (size_t)j * w]));
        std::vector<double> PP((size_t)dim * dim);
        if (!hip_ok(rails_gram(ctx, P.panel(), 0, dim, P.panel(), 0, dim, PP.data(), dim), "rails_gram")) return fail();
        double orth = 0.0, against_old = 0.0;
        int wi = -1, wj = -1;
        for (int j = 0; j < dim; ++j)
            for (int i = 0; i < dim; ++i) {
                const double e = std::fabs(PP[i + (size_t)j * dim] - (i == j ? 1.0 : 0.0));
                if (e > orth) { orth = e; wi = i; wj = j; }
                if (j >= dim_before && i < dim_before) against_old = std::max(against_old, e);
            }
        if (trace) std::cerr << "absorb verified: dim " << dim_before << " -> " << dim << " w " << w << ": |X - P c| / |X| <= " << worst << ", |P'P - I| = " << orth << " at (" << wi << ", " << wj << "), new against old columns " << against_old << std::endl;
        verify_repr = std::max(verify_repr, worst);
        verify_orth = std::max(verify_orth, orth);
        return true;
    }
    bool absorb_tail_impl(int w, double *coef)
    {
        if (!resolve_pending()) return false;
        n_absorb++;
        n_absorb_cols += w;
        if (w <= 0) return true;
        Tick tick(this, &t_absorb, "absorb");
        const int ld = row_cap;
        rails_panel *pp = P.panel();
        // one Gram call per round gives both the projections P'X (first dim rows) and the block's own Gram matrix X'X (last w rows):
        // X sits right behind P in the same panel
        const int dw = dim + w;
        std::vector<double> CG((size_t)dw * w), G0((size_t)w * w), G((size_t)w * w);
        bool have_G = false;
        int w2 = w;                // columns [0, w2) take part in the second round
        std::vector<double> C1t;   // first-round coefficients of the columns [w2, w) (for their Gram entries, see below)
        for (int round = 0; round < 2; ++round) {
            const int wr = round == 0 ? w : w2;
            if (!hip_ok(rails_gram(ctx, pp, 0, dw, pp, dim, wr, CG.data(), dw), "rails_gram")) return fail();
            if (round == 0) {
                for (int j = 0; j < w; ++j)
                    for (int i = 0; i < w; ++i) G0[i + (size_t)j * w] = CG[(dim + i) + (size_t)j * dw];
            }
            if (dim == 0) {
                G = G0;
                have_G = true;
                break;
            }
            std::vector<double> c2(wr, 0.0); // squared length of what this round finds along P, per column
            for (int j = 0; j < wr; ++j)
                for (int i = 0; i < dim; ++i) c2[j] += CG[i + (size_t)j * dw] * CG[i + (size_t)j * dw];
            for (int j = 0; j < wr; ++j)
                for (int i = 0; i < dim; ++i) coef[i + (size_t)j * ld] += CG[i + (size_t)j * dw];
            if (round == 0) {
                // "twice is enough" (Kahan / Parlett; the DGKS rule): one projection leaves a component (eps + delta) * ||x|| / ||x'||
                // along P, delta = ||P'P - I||.  Where at least half of a column's squared norm survives that factor is <= sqrt(2)
                // and a second projection has nothing to repair.  A looser rule is unstable over long runs: the defect of each new
                // basis column is the old delta times ||x|| / ||x'||, and chains of small survivals compound it (measured with 1 %: V'V - I
                // of 1e-14, 2e-12, 6e-7, 0.9 after 50, 100, 200, 400 trips of a stagnating solve).
                // The rule is applied per column; the second round runs on the leading columns up to the last one that needs it (the
                // prefetched random vector at the end of an A*W block never does: one MFMA tile of 16 columns instead of two).
                double worst = 1.0; // smallest fraction of a column's squared norm that survives the projection
                w2 = 0;
                for (int j = 0; j < w; ++j) {
                    const double g = G0[j + (size_t)j * w];
                    const double surv = g > 0.0 ? 1.0 - c2[j] / g : 0.0;
                    worst = std::min(worst, surv);
                    if (!(surv > reorth_survival)) w2 = j + 1;
                }
                if (trace) std::cerr << "absorb: dim " << dim << " w " << w << " first round: smallest survival " << worst << ", second round on " << w2 << " columns" << std::endl;
                // (the overlapped form queues the update of this round itself: fused with the second projection, one pass over P)
                if (overlap && w <= 48 && worst >= overlap_min_survival && start_overlapped(w, w2, coef, CG, G0)) return true;
                if (failed) return false; // (the chain could not be queued: part of it may have run, the block is not to be touched again)
                if (!hip_ok(rails_panel_gemm(ctx, -1.0, pp, 0, dim, CG.data(), dw, wr, 1.0, pp, dim), "rails_panel_gemm")) return fail();
                if (w2 == 0) break;
                n_second_round++;
                if (w2 < w) { // keep what the Gram entries of the untouched columns need
                    C1t.assign((size_t)dim * (w - w2), 0.0);
                    for (int j = w2; j < w; ++j) memcpy(C1t.data() + (size_t)(j - w2) * dim, CG.data() + (size_t)j * dw, sizeof(double) * dim);
                }
            } else {
                if (!hip_ok(rails_panel_gemm(ctx, -1.0, pp, 0, dim, CG.data(), dw, wr, 1.0, pp, dim), "rails_panel_gemm")) return fail();
                // The block's Gram matrix after the second round.  Columns i, j < w2: the one just measured minus the (tiny) second
                // correction.  One of them >= w2: as measured (the correction term is the product of two rounding-level quantities).
                // Both >= w2 (projected once, more than half survived): the first Gram matrix minus the first correction.
                for (int j = 0; j < w2; ++j)
                    for (int i = 0; i < w; ++i) {
                        double v = CG[(dim + i) + (size_t)j * dw];
                        if (i < w2) {
                            double s2 = 0.0;
                            for (int l = 0; l < dim; ++l) s2 += CG[l + (size_t)i * dw] * CG[l + (size_t)j * dw];
                            v -= s2;
                        }
                        G[i + (size_t)j * w] = v;
                        G[j + (size_t)i * w] = v;
                    }
                for (int j = w2; j < w; ++j)
                    for (int i = w2; i < w; ++i) {
                        double s1 = 0.0;
                        const double *ci = C1t.data() + (size_t)(i - w2) * dim, *cj = C1t.data() + (size_t)(j - w2) * dim;
                        for (int l = 0; l < dim; ++l) s1 += ci[l] * cj[l];
                        G[i + (size_t)j * w] = G0[i + (size_t)j * w] - s1;
                    }
                have_G = true;
            }
        }
        if (!have_G && !hip_ok(rails_gram(ctx, pp, dim, w, pp, dim, w, G.data(), w), "rails_gram")) return fail();
        // columns that (numerically) lie in span(P): nothing new to add
        std::vector<int> keep;
        for (int j = 0; j < w; ++j) {
            if (G[j + (size_t)j * w] > 1e-26 * G0[j + (size_t)j * w] && G[j + (size_t)j * w] > 0.0)
                keep.push_back(j);
            else {
                n_dropped++;
                if (trace) std::cerr << "absorb: column " << j << " dropped after the projections: " << G[j + (size_t)j * w] << " left of " << G0[j + (size_t)j * w] << std::endl;
            }
        }
        // A column of which less than 1e-4 of its length survived the projections may be nothing but their rounding error --
        // normalised, such a "direction" would not be orthogonal to P (error ~ eps / survival) and poison the basis; this happens
        // when span(P) is (nearly) the whole space or the block repeats old vectors.  Test: normalise the survivors and project
        // once more; a genuine direction keeps most of its unit length, rounding noise does not.
        std::vector<double> colscale(w, 1.0); // original residual column = colscale * column now in the panel
        bool delicate = false;
        for (int j : keep)
            if (G[j + (size_t)j * w] < 1e-8 * G0[j + (size_t)j * w]) delicate = true;
        if (delicate && dim > 0) {
            n_delicate++;
            for (int j : keep) {
                colscale[j] = std::sqrt(G[j + (size_t)j * w]);
                if (!hip_ok(rails_panel_scale(ctx, pp, dim + j, 1, 1.0 / colscale[j]), "rails_panel_scale")) return fail();
            }
            if (!hip_ok(rails_gram(ctx, pp, 0, dim, pp, dim, w, CG.data(), dim), "rails_gram")) return fail();
            if (!hip_ok(rails_panel_gemm(ctx, -1.0, pp, 0, dim, CG.data(), dim, w, 1.0, pp, dim), "rails_panel_gemm")) return fail();
            if (!hip_ok(rails_gram(ctx, pp, dim, w, pp, dim, w, G.data(), w), "rails_gram")) return fail();
            std::vector<int> survivors;
            for (int j : keep) {
                // (what the third projection took out belongs to the column whether or not the rest of it is kept)
                for (int i = 0; i < dim; ++i) coef[i + (size_t)j * ld] += colscale[j] * CG[i + (size_t)j * dim];
                if (G[j + (size_t)j * w] > 0.25)
                    survivors.push_back(j);
                else {
                    n_dropped++;
                    if (trace) std::cerr << "absorb: delicate column " << j << " dropped: " << G[j + (size_t)j * w] << " of its unit length left (it was " << colscale[j] << " long)" << std::endl;
                }
            }
            keep.swap(survivors);
        }
        const int r = (int)keep.size();
        if (r == 0) return true;
        // Cholesky of the diagonally scaled Gram matrix of the kept columns
        std::vector<double> d(r), S((size_t)r * r);
        for (int a = 0; a < r; ++a) d[a] = std::sqrt(G[keep[a] + (size_t)keep[a] * w]);
        for (int b = 0; b < r; ++b)
            for (int a = 0; a < r; ++a) S[a + (size_t)b * r] = G[keep[a] + (size_t)keep[b] * w] / (d[a] * d[b]);
        std::vector<double> R1 = S;
        int info = 0;
        rails_dpotrf('U', r, R1.data(), r, &info);
        bool ok = info == 0;
        for (int a = 0; a < r && ok; ++a)
            if (!(R1[a + (size_t)a * r] > 1e-6)) ok = false;
        if (!ok) return absorb_one_by_one(w, coef, keep, colscale);
        for (int b = 0; b < r; ++b)
            for (int a = b + 1; a < r; ++a) R1[a + (size_t)b * r] = 0.0;
        // Q1 = X[:, keep] D^-1 R1^-1, written over the whole tail block in place (dropped columns become zero, Q1 fills the
        // first r tail columns)
        std::vector<double> M1((size_t)w * w, 0.0), Rinv((size_t)r * r, 0.0);
        upper_inverse(R1, r, Rinv);
        for (int b = 0; b < r; ++b)
            for (int a = 0; a <= b; ++a) M1[keep[a] + (size_t)b * w] = Rinv[a + (size_t)b * r] / d[a];
        if (!hip_ok(rails_panel_gemm(ctx, 1.0, pp, dim, w, M1.data(), w, w, 0.0, pp, dim), "rails_panel_gemm")) return fail();
        // second pass: Q = Q1 R2^-1
        std::vector<double> G2((size_t)r * r), R2;
        if (!hip_ok(rails_gram(ctx, pp, dim, r, pp, dim, r, G2.data(), r), "rails_gram")) return fail();
        R2 = G2;
        rails_dpotrf('U', r, R2.data(), r, &info);
        if (info != 0) return fail("Cholesky of a nearly orthonormal block failed");
        for (int b = 0; b < r; ++b)
            for (int a = b + 1; a < r; ++a) R2[a + (size_t)b * r] = 0.0;
        std::vector<double> R2inv((size_t)r * r, 0.0);
        upper_inverse(R2, r, R2inv);
        if (!hip_ok(rails_panel_gemm(ctx, 1.0, pp, dim, r, R2inv.data(), r, r, 0.0, pp, dim), "rails_panel_gemm")) return fail();
        // X[:, keep[b]] = Q * (R2 * R1(:, b)) * d[b]
        std::vector<double> Rf((size_t)r * r, 0.0);
        for (int b = 0; b < r; ++b)
            for (int a = 0; a <= b; ++a) {
                double s = 0.0;
                for (int l = a; l <= b; ++l) s += R2[a + (size_t)l * r] * R1[l + (size_t)b * r];
                Rf[a + (size_t)b * r] = s * d[b] * colscale[keep[b]];
                coef[(dim + a) + (size_t)keep[b] * ld] = Rf[a + (size_t)b * r];
            }
        // An ill-conditioned block: Q = X R^-1 has magnified what rounding left of span(P) in X by up to 1 / min diag(R1).  Take
        // it out again (it is tiny: the block stays orthonormal to second order) and book it on the old coordinates.
        double rmin = 1.0;
        for (int a = 0; a < r; ++a) rmin = std::min(rmin, R1[a + (size_t)a * r]);
        if (rmin < 1e-2 && dim > 0) {
            n_reprojected++;
            std::vector<double> C3((size_t)dim * r);
            if (!hip_ok(rails_gram(ctx, pp, 0, dim, pp, dim, r, C3.data(), dim), "rails_gram")) return fail();
            if (!hip_ok(rails_panel_gemm(ctx, -1.0, pp, 0, dim, C3.data(), dim, r, 1.0, pp, dim), "rails_panel_gemm")) return fail();
            for (int b = 0; b < r; ++b)
                for (int l = 0; l <= b; ++l) {
                    const double f = Rf[l + (size_t)b * r];
                    for (int i = 0; i < dim; ++i) coef[i + (size_t)keep[b] * ld] += C3[i + (size_t)l * dim] * f;
                }
        }
        dim += r;
        P.resize(dim);
        return true;
    }

    // Re-base P on an orthonormal basis of the column space of all live coefficient matrices.  Exact (to rounding) for every
    // live multivector; directions no live object uses any more (stale Lanczos start vectors, pre-restart V and AV) go away.
    void compress()
    {
        if (!resolve_pending()) return;
        // (the rotation below is a plain wide GEMM: it goes through the platform's BLAS where the application has asked for that --
        // rails_ctx_enable_library_gemm, as bench.py does when it sets up.  Not by itself: loading the library and its kernels takes
        // 0.3 s in a warm process and several seconds in a cold one, against 2 ms saved per restart.)
        std::vector<std::shared_ptr<CoefStore>> stores;
        std::vector<std::weak_ptr<CoefStore>> still;
        int ncols = 0;
        for (auto &w : live)
            if (auto s = w.lock()) {
                stores.push_back(s);
                still.push_back(w);
                ncols += s->ncap;
            }
        live.swap(still);
        if (dim == 0 || ncols == 0) return;
        // only columns that hold something take part in the factorisation (capacity columns and discarded ones are zero)
        std::vector<double> Call((size_t)dim * ncols);
        int c0 = 0;
        for (auto &s : stores)
            for (int j = 0; j < s->ncap; ++j) {
                const double *cj = s->col(j);
                bool nz = false;
                for (int i = 0; i < dim && !nz; ++i) nz = cj[i] != 0.0;
                if (!nz) continue;
                memcpy(Call.data() + (size_t)c0 * dim, cj, sizeof(double) * dim);
                ++c0;
            }
        ncols = c0;
        if (trace) {
            std::cerr << "compress: dim " << dim << ", " << stores.size() << " live stores, columns in use:";
            for (auto &st : stores) {
                int used = 0;
                for (int j = 0; j < st->ncap; ++j) {
                    const double *cj = st->col(j);
                    for (int i = 0; i < dim; ++i)
                        if (cj[i] != 0.0) {
                            ++used;
                            break;
                        }
                }
                std::cerr << " " << used << "/" << st->ncap;
            }
            std::cerr << " -> " << ncols << " columns" << std::endl;
        }
        if (ncols == 0) return;
        std::vector<double> Q((size_t)dim * std::min(dim, ncols));
        int rank = 0, info = 0;
        {
            Tick tick(this, &t_qr, "compress: pivoted QR");
            rails_range_basis(dim, ncols, Call.data(), dim, 1e-14, Q.data(), dim, &rank, &info);
        }
        if (info != 0 || rank <= 0 || rank >= dim - 8) return; // nothing (worth it) to drop
        // device: P2 = P * Q
        if (P2.N() < 0 || P2.capacity() < rank + 64) {
            P2 = HipMultiVectorWrapper(m_local, rank + 128, ctx);
            P2.set_global_rows(m_global);
        }
        P2.resize(rank);
        {
            Tick rot(this, &t_rotate, "compress: rotation");
            if (!hip_ok(rails_panel_gemm_wide(ctx, 1.0, P.panel(), 0, dim, Q.data(), dim, rank, 0.0, P2.panel(), 0), "rails_panel_gemm_wide")) {
                failed = true;
                return;
            }
        }
        Tick tick(this, &t_recoef, "compress: coefficients");
        // host: C <- Q' C (columns up to the last non-zero one of every store)
        for (auto &s : stores) {
            int used = 0;
            for (int j = s->ncap - 1; j >= 0 && !used; --j) {
                const double *cj = s->col(j);
                for (int i = 0; i < dim; ++i)
                    if (cj[i] != 0.0) {
                        used = j + 1;
                        break;
                    }
            }
            if (used == 0) continue;
            std::vector<double> nc((size_t)rank * used);
            rails_dgemm('T', 'N', rank, used, dim, 1.0, Q.data(), dim, s->c.data(), s->ld, 0.0, nc.data(), rank);
            for (int j = 0; j < used; ++j) {
                memcpy(s->col(j), nc.data() + (size_t)j * rank, sizeof(double) * rank);
                std::fill(s->col(j) + rank, s->col(j) + dim, 0.0);
            }
        }
        std::swap(P, P2);
        dim = rank;
        n_compress++;
    }

    // The first round of a block has been applied (X <- X - P C1 is queued, coef holds C1): predict the block's coordinates along its
    // own new basis columns, queue the rest of the orthogonalisation on the device, and return without waiting.  false: conditions not
    // met (nothing was queued; the caller goes on in the ordinary way).
    bool start_overlapped(int w, int w2, double *coef, std::vector<double> const &CG, std::vector<double> const &G0)
    {
        const int ld = row_cap, dw = dim + w;
        // Gram matrix of the projected block by Pythagoras, its scaled Cholesky factor
        std::vector<double> Gp((size_t)w * w), d(w);
        for (int j = 0; j < w; ++j)
            for (int i = 0; i <= j; ++i) {
                double s = 0.0;
                for (int l = 0; l < dim; ++l) s += CG[l + (size_t)i * dw] * CG[l + (size_t)j * dw];
                Gp[i + (size_t)j * w] = Gp[j + (size_t)i * w] = G0[i + (size_t)j * w] - s;
            }
        for (int j = 0; j < w; ++j) {
            if (!(Gp[j + (size_t)j * w] > 1e-8 * G0[j + (size_t)j * w]) || !(Gp[j + (size_t)j * w] > 0.0)) return false;
            d[j] = std::sqrt(Gp[j + (size_t)j * w]);
        }
        std::vector<double> R1((size_t)w * w);
        for (int b = 0; b < w; ++b)
            for (int a = 0; a < w; ++a) R1[a + (size_t)b * w] = Gp[a + (size_t)b * w] / (d[a] * d[b]);
        int info = 0;
        rails_dpotrf('U', w, R1.data(), w, &info);
        if (info != 0) return false;
        for (int a = 0; a < w; ++a)
            if (!(R1[a + (size_t)a * w] > 1e-2)) return false; // an ill-conditioned block takes the careful way (re-projection)
        if (!hip_ok(rails_deferred_reserve(ctx, N_SLOTS, (int64_t)(P.capacity() + 64) * (w <= 32 ? 32 : 48)), "rails_deferred_reserve")) return false;
        // the device's part: the first update, the second round on the leading w2 columns, Gram matrix of the block, its Cholesky
        // factor inverted, Q1 = X M1, the same once more (CholQR2)
        rails_panel *pp = P.panel();
        bool ok = true;
        if (w2 > 0) {
            // first update and second projection in one pass over P (rails_update_gram_deferred), then the second update from the slot
            ok = ok && hip_ok(rails_update_gram_deferred(ctx, -1.0, pp, 0, dim, CG.data(), dw, w, pp, dim, w2, SLOT_C2), "rails_update_gram_deferred");
            ok = ok && hip_ok(rails_panel_gemm_deferred(ctx, -1.0, pp, 0, dim, SLOT_C2, dim, w2, 1.0, pp, dim), "rails_panel_gemm_deferred");
        } else
            ok = ok && hip_ok(rails_panel_gemm(ctx, -1.0, pp, 0, dim, CG.data(), dw, w, 1.0, pp, dim), "rails_panel_gemm");
        ok = ok && hip_ok(rails_gram_deferred(ctx, pp, dim, w, pp, dim, w, SLOT_G), "rails_gram_deferred");
        ok = ok && hip_ok(rails_chol_inverse_deferred(ctx, SLOT_G, w, SLOT_M1), "rails_chol_inverse_deferred");
        ok = ok && hip_ok(rails_panel_gemm_deferred(ctx, 1.0, pp, dim, w, SLOT_M1, w, w, 0.0, pp, dim), "rails_panel_gemm_deferred");
        ok = ok && hip_ok(rails_gram_deferred(ctx, pp, dim, w, pp, dim, w, SLOT_G2), "rails_gram_deferred");
        ok = ok && hip_ok(rails_chol_inverse_deferred(ctx, SLOT_G2, w, SLOT_M2), "rails_chol_inverse_deferred");
        ok = ok && hip_ok(rails_panel_gemm_deferred(ctx, 1.0, pp, dim, w, SLOT_M2, w, w, 0.0, pp, dim), "rails_panel_gemm_deferred");
        if (!ok) return fail("the overlapped block orthogonalisation could not be queued"); // false, with `failed` latched: the caller gives up
        // predicted coordinates along the new columns: X = Q Rfp, Rfp = R1 D
        pending.active = true;
        pending.dim0 = dim;
        pending.w = w;
        pending.w2 = w2;
        pending.Rfp.assign((size_t)w * w, 0.0);
        pending.g0diag.resize(w);
        for (int b = 0; b < w; ++b) {
            pending.g0diag[b] = G0[b + (size_t)b * w];
            for (int a = 0; a <= b; ++a) {
                pending.Rfp[a + (size_t)b * w] = R1[a + (size_t)b * w] * d[b];
                coef[(dim + a) + (size_t)b * ld] = pending.Rfp[a + (size_t)b * w];
            }
        }
        if (w2 > 0) n_second_round++;
        n_overlapped++;
        // test hook (tests/test_gpu_solver.py): the n-th overlapped block books a prediction that is off by 1 %, so that the read-back's
        // check has something to reject -- the failure has to be loud and end the run
        static const long spoil = getenv("RAILS_SUBSPACE_TEST_SPOIL_PREDICTION") ? atol(getenv("RAILS_SUBSPACE_TEST_SPOIL_PREDICTION")) : 0;
        if (spoil > 0 && n_overlapped == spoil) {
            pending.Rfp[0] *= 1.01;
            coef[dim] = pending.Rfp[0];
        }
        dim += w;
        P.resize(dim);
        return true;
    }

    // What is checked here can only fail if the prediction was grossly wrong: the block was taken on because (by Pythagoras, relative
    // error eps / survival <= 1e-12) every column keeps >= 1e-4 of its squared length and the scaled Cholesky factor's diagonal is above
    // 1e-2; the read-back asks for 1e-8 and 1e-6 and for the predicted factor to match the real one to 1e-4.  A block that misses those by
    // four to ten orders of magnitude means the device did not meet the block the host was promised (a faulted kernel, non-finite data):
    // by then the host has used the predicted coordinates in a projected solve and a Lanczos run, and the block itself has been
    // overwritten in place -- there is nothing sound to fall back to, so the failure is latched (`failed`), reported on stderr, and ends the
    // run at the next trip (Solver::set_failure_check); rails_solver_solve returns RAILS_EHIP.  Blocks that need the careful treatment
    // (nearly dependent columns, directions already in span(P)) never get here: they fail the conditions of start_overlapped and take the
    // synchronous path of absorb_tail.
    // Read back what the queued orthogonalisation produced and move every live coefficient store from the predicted new basis columns to
    // the real ones: with X = P (C1 + C2) + Q Rft the truth and (C1, Rfp) what was booked, a vector with booked coordinates
    // (a_old, a_new) is P (a_old + C2 Rfp^-1 a_new) + Q (Rft Rfp^-1 a_new).
    bool resolve_pending()
    {
        if (!pending.active) return !failed;
        pending.active = false;
        const int d0 = pending.dim0, w = pending.w, w2 = pending.w2;
        if (!hip_ok(rails_ctx_sync(ctx), "rails_ctx_sync")) return fail();
        std::vector<double> C2((size_t)d0 * w, 0.0), G((size_t)w * w), G2((size_t)w * w);
        bool ok = true;
        if (w2 > 0) ok = ok && hip_ok(rails_deferred_fetch(ctx, SLOT_C2, (int64_t)d0 * w2, C2.data()), "rails_deferred_fetch");
        ok = ok && hip_ok(rails_deferred_fetch(ctx, SLOT_G, (int64_t)w * w, G.data()), "rails_deferred_fetch");
        ok = ok && hip_ok(rails_deferred_fetch(ctx, SLOT_G2, (int64_t)w * w, G2.data()), "rails_deferred_fetch");
        if (!ok) return fail();
        // the factors the device applied, repeated on the host's copies of the two Gram matrices
        auto factor = [&](std::vector<double> const &Gm, std::vector<double> &R, std::vector<double> &dd) {
            dd.resize(w);
            R.assign((size_t)w * w, 0.0);
            for (int j = 0; j < w; ++j) {
                if (!(Gm[j + (size_t)j * w] > 0.0) || !std::isfinite(Gm[j + (size_t)j * w])) return false;
                dd[j] = std::sqrt(Gm[j + (size_t)j * w]);
            }
            for (int b = 0; b < w; ++b)
                for (int a = 0; a < w; ++a) R[a + (size_t)b * w] = Gm[a + (size_t)b * w] / (dd[a] * dd[b]);
            int inf = 0;
            rails_dpotrf('U', w, R.data(), w, &inf);
            if (inf != 0) return false;
            for (int b = 0; b < w; ++b)
                for (int a = b + 1; a < w; ++a) R[a + (size_t)b * w] = 0.0;
            return true;
        };
        std::vector<double> R1, d1, R2, d2;
        if (!factor(G, R1, d1) || !factor(G2, R2, d2)) return fail("the overlapped block orthogonalisation met a block that is not of full rank");
        for (int j = 0; j < w; ++j)
            if (!(G[j + (size_t)j * w] > 1e-8 * pending.g0diag[j]) || !(R1[j + (size_t)j * w] > 1e-6))
                return fail("the overlapped block orthogonalisation met a block it should have treated with care");
        // Rft = (R2 D2) (R1 D1), upper triangular
        std::vector<double> Rft((size_t)w * w, 0.0);
        for (int b = 0; b < w; ++b)
            for (int a = 0; a <= b; ++a) {
                double sum = 0.0;
                for (int l = a; l <= b; ++l) sum += R2[a + (size_t)l * w] * d2[l] * R1[l + (size_t)b * w];
                Rft[a + (size_t)b * w] = sum * d1[b];
            }
        // Tn = Rft Rfp^-1 (w x w, upper), To = [C2 0] Rfp^-1 (d0 x w)
        std::vector<double> Rpi((size_t)w * w, 0.0), Tn((size_t)w * w, 0.0), To((size_t)d0 * w, 0.0);
        upper_inverse(pending.Rfp, w, Rpi);
        double off = 0.0;
        for (int b = 0; b < w; ++b)
            for (int a = 0; a <= b; ++a) {
                double sum = 0.0;
                for (int l = a; l <= b; ++l) sum += Rft[a + (size_t)l * w] * Rpi[l + (size_t)b * w];
                Tn[a + (size_t)b * w] = sum;
                off = std::max(off, std::fabs(sum - (a == b ? 1.0 : 0.0)));
            }
        if (!(off < 1e-4)) return fail("the overlapped block orthogonalisation did not confirm its prediction");
        if (w2 > 0) rails_dgemm('N', 'N', d0, w, w2, 1.0, C2.data(), d0, Rpi.data(), w, 0.0, To.data(), d0);
        if (trace) std::cerr << "absorb (overlapped): dim " << d0 << " w " << w << ": prediction off by " << off << std::endl;
        std::vector<double> an(w), bn(w);
        for (auto &wk : live)
            if (auto st = wk.lock()) {
                for (int j = 0; j < st->ncap; ++j) {
                    double *cj = st->col(j);
                    bool nz = false;
                    for (int a = 0; a < w; ++a) {
                        an[a] = cj[d0 + a];
                        nz = nz || an[a] != 0.0;
                    }
                    if (!nz) continue;
                    if (w2 > 0)
                        for (int a = 0; a < w; ++a) {
                            if (an[a] == 0.0) continue;
                            const double *t = To.data() + (size_t)a * d0;
                            for (int i = 0; i < d0; ++i) cj[i] += t[i] * an[a];
                        }
                    for (int a = 0; a < w; ++a) {
                        double sum = 0.0;
                        for (int l = a; l < w; ++l) sum += Tn[a + (size_t)l * w] * an[l];
                        bn[a] = sum;
                    }
                    for (int a = 0; a < w; ++a) cj[d0 + a] = bn[a];
                }
            }
        if (verify && !failed) { // (RAILS_SUBSPACE_VERIFY: the basis with the block the device has just finished)
            std::vector<double> PP((size_t)dim * dim);
            if (!hip_ok(rails_gram(ctx, P.panel(), 0, dim, P.panel(), 0, dim, PP.data(), dim), "rails_gram")) return fail();
            double orth = 0.0;
            for (int j = 0; j < dim; ++j)
                for (int i = 0; i < dim; ++i) orth = std::max(orth, std::fabs(PP[i + (size_t)j * dim] - (i == j ? 1.0 : 0.0)));
            if (trace) std::cerr << "absorb (overlapped) verified: dim " << d0 << " -> " << dim << ": |P'P - I| = " << orth << std::endl;
            verify_orth = std::max(verify_orth, orth);
        }
        return !failed;
    }

private:
    bool fail(const char *what = nullptr)
    {
        if (what) std::cerr << "rails_amd: SubspaceBasis: " << what << std::endl;
        failed = true;
        return false;
    }

    static void upper_inverse(std::vector<double> const &R, int r, std::vector<double> &Rinv)
    {
        for (int j = 0; j < r; ++j) {
            Rinv[j + (size_t)j * r] = 1.0 / R[j + (size_t)j * r];
            for (int i = j - 1; i >= 0; --i) {
                double s = 0.0;
                for (int l = i + 1; l <= j; ++l) s += R[i + (size_t)l * r] * Rinv[l + (size_t)j * r];
                Rinv[i + (size_t)j * r] = -s / R[i + (size_t)i * r];
            }
        }
    }

    // degenerate block (its kept columns are dependent among themselves after the projection): take the columns one at a
    // time, each against the basis extended by its predecessors.  The tail block holds the twice-projected columns; coef holds
    // their coordinates in the old basis.
    bool absorb_one_by_one(int w, double *coef, std::vector<int> const &keep, std::vector<double> const &colscale)
    {
        n_single++;
        const int ld = row_cap;
        rails_panel *pp = P.panel();
        const int base = dim;
        // move the kept columns to the front of the tail one at a time: column j of the block sits at base + j
        int accepted = 0;
        for (int idx = 0; idx < (int)keep.size(); ++idx) {
            const int j = keep[idx];
            const int src = base + j, dst = base + accepted;
            if (src != dst && !hip_ok(rails_panel_copy(ctx, pp, src, 1, pp, dst), "rails_panel_copy")) return fail();
            double g0 = 0.0, g = 0.0;
            if (!hip_ok(rails_gram(ctx, pp, dst, 1, pp, dst, 1, &g0, 1), "rails_gram")) return fail();
            // Two rounds against the WHOLE basis so far, not only against the columns accepted from this block: a column that keeps a
            // fraction f of its length here inherits the accepted columns' defects against the old basis magnified by 1 / f, and in a
            // chain of nearly dependent columns (the first residual directions of a run all lie close to span(B)) that compounds
            // geometrically -- measured on BASELINE configs[1]: P'P - I of 2e-14, 2e-12, 1e-10, 4e-9, 1e-8 along one block of 25
            // columns, 3e-5 two trips later, and projected matrices V'AV off by the same.
            for (int round = 0; round < 2 && accepted > 0; ++round) {
                const int nb = base + accepted;
                std::vector<double> c(nb);
                if (!hip_ok(rails_gram(ctx, pp, 0, nb, pp, dst, 1, c.data(), nb), "rails_gram")) return fail();
                if (!hip_ok(rails_panel_gemm(ctx, -1.0, pp, 0, nb, c.data(), nb, 1, 1.0, pp, dst), "rails_panel_gemm")) return fail();
                for (int a = 0; a < nb; ++a) coef[a + (size_t)j * ld] += colscale[j] * c[a];
            }
            if (!hip_ok(rails_gram(ctx, pp, dst, 1, pp, dst, 1, &g, 1), "rails_gram")) return fail();
            if (!(g > 1e-24 * g0) || !(g > 0.0)) {
                n_dropped++;
                if (trace) std::cerr << "absorb one by one: column " << j << " dropped: " << g << " left of " << g0 << " (scale " << colscale[j] << ")" << std::endl;
                continue;
            }
            const double nrm = std::sqrt(g);
            if (!hip_ok(rails_panel_scale(ctx, pp, dst, 1, 1.0 / nrm), "rails_panel_scale")) return fail();
            double last = nrm;
            if (g < 1e-8 * g0 && base + accepted > 0) {
                // less than 1e-4 of the column was left: is it a direction or the rounding error of the projections?  Project the
                // normalised vector against the whole basis so far; a direction survives (and is now orthogonal to rounding)
                const int nb = base + accepted;
                std::vector<double> c(nb);
                double n2 = 0.0;
                if (!hip_ok(rails_gram(ctx, pp, 0, nb, pp, dst, 1, c.data(), nb), "rails_gram")) return fail();
                if (!hip_ok(rails_panel_gemm(ctx, -1.0, pp, 0, nb, c.data(), nb, 1, 1.0, pp, dst), "rails_panel_gemm")) return fail();
                if (!hip_ok(rails_gram(ctx, pp, dst, 1, pp, dst, 1, &n2, 1), "rails_gram")) return fail();
                // what the projection took out belongs to the column whether or not the rest of it is kept
                for (int a = 0; a < nb; ++a) coef[a + (size_t)j * ld] += colscale[j] * nrm * c[a];
                if (!(n2 > 0.25)) {
                    n_dropped++;
                    if (trace) std::cerr << "absorb one by one: column " << j << " dropped after the check against the whole basis: " << n2 << " of its unit length left (it was " << colscale[j] * nrm << " long)" << std::endl;
                    continue;
                }
                if (!hip_ok(rails_panel_scale(ctx, pp, dst, 1, 1.0 / std::sqrt(n2)), "rails_panel_scale")) return fail();
                last = nrm * std::sqrt(n2);
            }
            coef[(base + accepted) + (size_t)j * ld] = colscale[j] * last;
            if (trace) std::cerr << "absorb one by one: column " << j << " -> basis column " << base + accepted << ": " << g << " left of " << g0 << " (scale " << colscale[j] << ")" << std::endl;
            if (verify && trace && base + accepted > 0) {
                std::vector<double> c(base + accepted);
                rails_gram(ctx, pp, 0, base + accepted, pp, dst, 1, c.data(), base + accepted);
                double e = 0.0;
                int at = -1;
                for (int a = 0; a < base + accepted; ++a)
                    if (std::fabs(c[a]) > e) { e = std::fabs(c[a]); at = a; }
                std::cerr << "    against the basis so far: " << e << " at column " << at << std::endl;
            }
            accepted++;
        }
        dim = base + accepted;
        P.resize(dim);
        return true;
    }
};

class SubspaceOperator;

// The MultiVector role.  Value semantics as in the reference (src/StlWrapper.cpp:31-121): deep-copy construction, assignment to a
// non-view shares storage, assignment to a view copies in; views are column windows.
class SubspaceMultiVector
{
    friend class SubspaceOperator;
    std::shared_ptr<SubspaceBasis> basis_;
    std::shared_ptr<CoefStore> store_;
    int c0_ = 0, n_ = -1;
    int orthogonalized_ = 0;
    bool is_view_ = false, transpose_ = false;

    int rows() const { return !store_ ? 0 : (store_->in_basis ? basis_->dim : store_->ld); }
    double *cptr(int j = 0) const { return store_->col(c0_ + j); }
    int ld() const { return store_->ld; }
    bool in_basis() const { return !store_ || store_->in_basis; }

    void ensure_cols(int n)
    {
        if (!store_) return;
        if (c0_ + n <= store_->ncap) return;
        int ncap = c0_ + n;
        store_->c.resize((size_t)store_->ld * ncap, 0.0);
        store_->ncap = ncap;
    }

public:
    SubspaceMultiVector() {}

    // an in-basis multivector with n columns (capacity n), all zero
    SubspaceMultiVector(std::shared_ptr<SubspaceBasis> const &b, int n) : basis_(b), store_(b->new_store(std::max(n, 1), true)), n_(n) {}

    // a plain small replicated matrix (rows x n)
    static SubspaceMultiVector Plain(std::shared_ptr<SubspaceBasis> const &b, int rows, int n)
    {
        SubspaceMultiVector out;
        out.basis_ = b;
        out.store_ = b->new_store(std::max(n, 1), false, rows);
        out.n_ = n;
        return out;
    }

    SubspaceMultiVector(SubspaceMultiVector const &o) // deep copy (src/StlWrapper.cpp:31-44)
        : basis_(o.basis_), c0_(0), n_(o.n_), orthogonalized_(o.orthogonalized_), is_view_(false), transpose_(o.transpose_)
    {
        if (o.store_) {
            int cap = std::max(o.store_->ncap - o.c0_, 1);
            store_ = basis_->new_store(cap, o.store_->in_basis, o.store_->ld);
            for (int j = 0; j < std::max(o.n_, 0); ++j) memcpy(store_->col(j), o.cptr(j), sizeof(double) * o.rows());
        }
    }
    SubspaceMultiVector(SubspaceMultiVector &&o) = default;

    SubspaceMultiVector(SubspaceMultiVector const &o, int n) // same row space, n columns (src/StlWrapper.cpp:46-51)
        : basis_(o.basis_), n_(n)
    {
        if (o.store_) store_ = basis_->new_store(std::max(n, 1), o.store_->in_basis, o.store_->ld);
    }

    virtual ~SubspaceMultiVector() {}

    std::shared_ptr<SubspaceBasis> const &basis() const { return basis_; }
    const double *coefficients() const { return cptr(); }
    int coefficient_ld() const { return ld(); }
    int coefficient_rows() const { return rows(); }

    SubspaceMultiVector &operator=(SubspaceMultiVector const &o)
    {
        if (!is_view_) { // share
            basis_ = o.basis_;
            store_ = o.store_;
            c0_ = o.c0_;
            n_ = o.n_;
            orthogonalized_ = o.orthogonalized_;
            transpose_ = o.transpose_;
            return *this;
        }
        int cols = std::min(n_, o.n_);
        for (int j = 0; j < cols; ++j) memcpy(cptr(j), o.cptr(j), sizeof(double) * rows());
        return *this;
    }

    SubspaceMultiVector &operator=(double v)
    {
        orthogonalized_ = 0;
        if (!store_ || n_ <= 0) return *this;
        if (!in_basis() || v == 0.0) {
            for (int j = 0; j < n_; ++j) std::fill_n(cptr(j), in_basis() ? ld() : rows(), v);
            return *this;
        }
        // every entry of the m-dimensional columns equal to v: one constant vector, expressed in the basis
        SubspaceBasis &b = *basis_;
        int t0 = b.tail(1);
        b.P.resize(t0 + 1);
        if (!hip_ok(rails_panel_fill(b.ctx, b.P.panel(), t0, 1, v), "rails_panel_fill")) b.failed = true;
        b.P.resize(t0);
        std::fill_n(cptr(0), ld(), 0.0);
        b.absorb_tail(1, cptr(0));
        for (int j = 1; j < n_; ++j) memcpy(cptr(j), cptr(0), sizeof(double) * ld());
        return *this;
    }

    SubspaceMultiVector &operator*=(double s)
    {
        for (int j = 0; j < n_; ++j) {
            double *c = cptr(j);
            for (int i = 0, r = rows(); i < r; ++i) c[i] *= s;
        }
        orthogonalized_ = 0;
        return *this;
    }
    SubspaceMultiVector &operator/=(double s) { return *this *= 1.0 / s; }

    SubspaceMultiVector &axpy(double a, SubspaceMultiVector const &o)
    {
        int cols = std::min(n_, o.n_);
        for (int j = 0; j < cols; ++j) {
            double *c = cptr(j);
            const double *x = o.cptr(j);
            for (int i = 0, r = rows(); i < r; ++i) c[i] += a * x[i];
        }
        orthogonalized_ = 0;
        return *this;
    }
    SubspaceMultiVector &operator+=(SubspaceMultiVector const &o) { return axpy(1.0, o); }
    SubspaceMultiVector &operator-=(SubspaceMultiVector const &o) { return axpy(-1.0, o); }
    SubspaceMultiVector operator+(SubspaceMultiVector const &o) const
    {
        SubspaceMultiVector out(*this);
        out += o;
        return out;
    }

    int M() const { return transpose_ ? n_ : (in_basis() ? (basis_ ? (int)basis_->m_global : -1) : store_->ld); }
    int N() const { return transpose_ ? (in_basis() ? (basis_ ? (int)basis_->m_global : -1) : store_->ld) : n_; }

    void resize(int n) // capacity preserving (src/StlWrapper.cpp:219-263)
    {
        orthogonalized_ = std::min(orthogonalized_, n);
        ensure_cols(n);
        n_ = n;
    }

    // zero the allocated columns past the ones in use: they hold pre-restart vectors that nothing reads again, and would keep
    // their directions alive in SubspaceBasis::compress()
    void discard_unused_columns()
    {
        if (!store_ || is_view_) return;
        for (int j = c0_ + std::max(n_, 0); j < store_->ncap; ++j) std::fill_n(store_->col(j), store_->ld, 0.0);
    }

    SubspaceMultiVector view(int a = -1, int b = -1) const
    {
        SubspaceMultiVector out;
        out.basis_ = basis_;
        out.store_ = store_;
        out.transpose_ = transpose_;
        out.is_view_ = true;
        int num = 1;
        if (b > 0 && a >= 0)
            num = b - a + 1;
        else if (a < 0) {
            a = 0;
            num = n_;
        }
        out.c0_ = c0_ + a;
        out.n_ = num;
        return out;
    }

    SubspaceMultiVector copy() const { return SubspaceMultiVector(*this); }

    void push_back(SubspaceMultiVector const &o) // src/StlWrapper.cpp:367-374
    {
        if (!store_) { // empty default-constructed target takes the shape of the source
            basis_ = o.basis_;
            store_ = basis_->new_store(std::max(o.n_, 1), o.store_->in_basis, o.store_->ld);
            n_ = 0;
        }
        int n = n_, on = o.n_;
        resize(n + on);
        for (int j = 0; j < on; ++j) memcpy(cptr(n + j), o.cptr(j), sizeof(double) * rows());
    }

    // U(-1,1) entries in the m-dimensional space (src/StlWrapper.cpp:414-423): drawn on the device into the basis panel's tail
    // (one RNG stream per call, like the direct back end), then expressed in the basis
    void random()
    {
        if (!in_basis() || n_ <= 0) {
            if (!in_basis()) std::cerr << "rails_amd: random() on a plain replicated object is not supported" << std::endl;
            return;
        }
        SubspaceBasis &b = *basis_;
        orthogonalized_ = 0;
        if (b.cached_valid) {
            b.cached_valid = false;
            if (n_ == 1) { // drawn ahead with the last A*W block
                memcpy(cptr(0), b.cached_random->col(0), sizeof(double) * ld());
                std::fill_n(b.cached_random->col(0), b.cached_random->ld, 0.0); // nothing for compress() to keep alive
                return;
            }
            std::cerr << "rails_amd: a prefetched random vector is discarded (random() on " << n_ << " columns)" << std::endl;
        }
        for (int j0 = 0; j0 < n_; j0 += 64) {
            int w = std::min(64, n_ - j0);
            int t0 = b.tail(w);
            b.P.resize(t0 + w);
            if (!hip_ok(rails_panel_random(b.ctx, b.P.panel(), t0, w), "rails_panel_random")) b.failed = true;
            b.P.resize(t0);
            for (int j = 0; j < w; ++j) std::fill_n(cptr(j0 + j), ld(), 0.0);
            b.absorb_tail(w, cptr(j0));
        }
        orthogonalized_ = 0;
    }

    SubspaceMultiVector transpose() const
    {
        SubspaceMultiVector out = view();
        out.is_view_ = false;
        out.transpose_ = !transpose_;
        out.orthogonalized_ = orthogonalized_;
        return out;
    }

    // X' Y (src/StlWrapper.cpp:394-412): P is orthonormal, so the m-dimensional inner products are coefficient inner products
    HostDenseMatrix dot(SubspaceMultiVector const &o) const
    {
        HostDenseMatrix out(n_, o.n_);
        if (in_basis() != o.in_basis() || rows() != o.rows()) {
            std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << o.M() << "x" << o.N() << std::endl;
            return out;
        }
        if (n_ > 0 && o.n_ > 0 && rows() > 0) rails_dgemm('T', 'N', n_, o.n_, rows(), 1.0, cptr(), ld(), o.cptr(), o.ld(), 0.0, (double *)out, out.LDA());
        return out;
    }

    SubspaceMultiVector operator*(HostDenseMatrix const &C) const // src/StlWrapper.cpp:168-187
    {
        SubspaceMultiVector out(*this, C.N());
        if (C.M() != n_) {
            std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << C.M() << "x" << C.N() << std::endl;
            return out;
        }
        if (C.N() > 0 && n_ > 0 && rows() > 0)
            rails_dgemm('N', C.transposed() ? 'T' : 'N', rows(), C.N(), n_, 1.0, cptr(), ld(), (double *)C, C.raw_ld(), 0.0, out.cptr(), out.ld());
        return out;
    }

    // op(this) * other (src/MatrixOrMultiVectorWrapper.hpp:54,59): B'W -> plain p x w; B y with y plain -> in-basis
    SubspaceMultiVector operator*(SubspaceMultiVector const &o) const
    {
        if (transpose_) {
            SubspaceMultiVector out = Plain(basis_, n_, o.n_);
            if (rows() != o.rows() || in_basis() != o.in_basis()) {
                std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << o.M() << "x" << o.N() << std::endl;
                return out;
            }
            if (n_ > 0 && o.n_ > 0 && rows() > 0) rails_dgemm('T', 'N', n_, o.n_, rows(), 1.0, cptr(), ld(), o.cptr(), o.ld(), 0.0, out.cptr(), out.ld());
            return out;
        }
        SubspaceMultiVector out(*this, o.n_);
        if (o.in_basis() || o.rows() != n_) {
            std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << o.M() << "x" << o.N() << std::endl;
            return out;
        }
        if (o.n_ > 0 && n_ > 0 && rows() > 0) rails_dgemm('N', 'N', rows(), o.n_, n_, 1.0, cptr(), ld(), o.cptr(), o.ld(), 0.0, out.cptr(), out.ld());
        return out;
    }

    double norm() const // spectral 2-norm (src/StlWrapper.cpp:265-289)
    {
        if (n_ <= 0) return 0.0;
        HostDenseMatrix G = dot(*this);
        std::vector<double> w(n_);
        int info = 0;
        rails_dsyev('V', 'U', n_, (double *)G, G.LDA(), w.data(), &info);
        double mx = 0.0;
        for (int i = 0; i < n_; ++i) mx = std::max(mx, std::sqrt(std::abs(w[i])));
        return mx;
    }

    // The reference's recurrence (src/StlWrapper.cpp:305-321) on the coefficient columns.  One case needs care here that the reference does
    // not know: a column that lies in the span of its predecessors (an expansion vector of a residual that has nothing new to offer -- an
    // invariant subspace has been reached, a block repeats old directions).  In R^m the reference normalises what two projections leave
    // of it -- rounding noise, i.e. an arbitrary direction orthogonal to the others -- and carries on with an orthonormal V.  In
    // coordinates that noise has nowhere to live once V fills span(P) (more columns than dimensions cannot be orthonormal; found with
    // tests/test_gpu_solver.py::test_subspace_backend_rank_deficient_expansion_blocks: V'V != I, estimates diverging).  So a column of
    // which less than 1e-13 of its unit length survives -- nothing but rounding: a remainder of 1e-10 still is a direction good to 1e-6,
    // and slowly converging solves (the MOC problem) live on those -- is replaced by what the reference's noise is in effect: a fresh
    // random direction (drawn on the device, absorbed into the basis like every random vector), orthogonalised like any other column.
    void orthogonalize()
    {
        std::vector<double> h;
        int replaced = 0;
        for (int i = orthogonalized_; i < n_; ++i) {
            const int r = rows();
            double *v = cptr(i);
            auto nrm2 = [&]() {
                double s = 0.0;
                for (int l = 0; l < r; ++l) s += v[l] * v[l];
                return std::sqrt(s);
            };
            double nr = nrm2();
            if (nr > 0.0)
                for (int l = 0; l < r; ++l) v[l] /= nr;
            for (int pass = 0; pass < 2 && i > 0; ++pass) {
                h.assign(i, 0.0);
                rails_dgemm('T', 'N', i, 1, r, 1.0, cptr(), ld(), v, ld(), 0.0, h.data(), i);
                rails_dgemm('N', 'N', r, 1, i, -1.0, cptr(), ld(), h.data(), i, 1.0, v, ld());
            }
            nr = nrm2();
            if (!(nr > 1e-13) && in_basis() && replaced < 2 * n_ + 8) {
                ++replaced;
                basis_->n_replaced++;
                SubspaceMultiVector fresh(basis_, 1);
                fresh.random(); // (may grow the stores' row capacity: pointers are taken again below)
                if (basis_->failed) break;
                std::fill_n(cptr(i), ld(), 0.0);
                memcpy(cptr(i), fresh.cptr(0), sizeof(double) * rows());
                --i; // the same column again, now with something outside the span of its predecessors
                continue;
            }
            for (int l = 0; l < r; ++l) v[l] /= nr;
        }
        orthogonalized_ = n_;
    }

    // device image P * C (m_local x n): used by the operator, the final read-out and the tests
    HipMultiVectorWrapper materialise() const { return basis_->materialise(cptr(), ld(), n_); }
    void to_host(double *data, int64_t ldd) const
    {
        if (!in_basis()) {
            for (int j = 0; j < n_; ++j) memcpy(data + (size_t)j * ldd, cptr(j), sizeof(double) * rows());
            return;
        }
        materialise().to_host(data, ldd);
    }

    // set the columns from m_local x n host data (tests, I/O): upload, express in the basis
    void from_host(const double *data, int64_t ldd)
    {
        orthogonalized_ = 0;
        if (!store_ || n_ <= 0) return;
        if (!in_basis()) {
            for (int j = 0; j < n_; ++j) memcpy(cptr(j), data + (size_t)j * ldd, sizeof(double) * rows());
            return;
        }
        HipMultiVectorWrapper X(basis_->m_local, n_, basis_->ctx);
        X.from_host(data, ldd);
        SubspaceMultiVector tmp = Absorb(basis_, X);
        for (int j = 0; j < n_; ++j) memcpy(cptr(j), tmp.cptr(j), sizeof(double) * ld());
    }
    int64_t local_rows() const { return in_basis() ? basis_->m_local : rows(); }
    int orthogonalized() const { return orthogonalized_; }
    void set_orthogonalized(int n) { orthogonalized_ = n; }
    int capacity() const { return store_ ? store_->ncap - c0_ : 0; }
    bool replicated() const { return !in_basis(); }
    const double *host_data() const { return cptr(); }

    // express a device multivector (m_local x n) in the basis, extending it as needed
    static SubspaceMultiVector Absorb(std::shared_ptr<SubspaceBasis> const &b, HipMultiVectorWrapper const &X)
    {
        const int n = X.N();
        SubspaceMultiVector out(b, n);
        for (int j0 = 0; j0 < n; j0 += 64) {
            int w = std::min(64, n - j0);
            int t0 = b->tail(w);
            b->P.resize(t0 + w);
            if (!hip_ok(rails_panel_copy(b->ctx, X.panel(), X.offset() + j0, w, b->P.panel(), t0), "rails_panel_copy")) b->failed = true;
            b->P.resize(t0);
            b->absorb_tail(w, out.cptr(j0));
        }
        return out;
    }
};

inline SubspaceMultiVector operator*(double d, SubspaceMultiVector const &o)
{
    SubspaceMultiVector out(o);
    out *= d;
    return out;
}

// The Matrix role: A * W = materialise W, CSR SpMM straight into the basis panel's tail, absorb
class SubspaceOperator
{
    HipOperatorWrapper A_;
    std::shared_ptr<SubspaceBasis> basis_;

public:
    SubspaceOperator() {}
    SubspaceOperator(HipOperatorWrapper const &A, std::shared_ptr<SubspaceBasis> const &b) : A_(A), basis_(b) {}
    virtual ~SubspaceOperator() {}

    int M() const { return A_.M(); }
    int N() const { return A_.N(); }
    SubspaceOperator transpose() const { return SubspaceOperator(A_.transpose(), basis_); }
    double norm() const { return A_.norm(); }

    SubspaceMultiVector operator*(SubspaceMultiVector const &X) const
    {
        SubspaceBasis &b = *basis_;
        const int n = X.N();
        SubspaceMultiVector out(basis_, n);
        if (n <= 0) return out;
        HipMultiVectorWrapper Xd = X.materialise();
        for (int j0 = 0; j0 < n; j0 += 64) {
            int w = std::min(64, n - j0);
            const bool pre = b.prefetch_random && !b.cached_valid && j0 + w == n && w < 64;
            int t0 = b.tail(w + (pre ? 1 : 0));
            b.P.resize(t0 + w + (pre ? 1 : 0));
            HipMultiVectorWrapper Xw = (w == 1) ? Xd.view(j0) : Xd.view(j0, j0 + w - 1);
            if (!A_.apply_into(Xw, b.P, t0)) b.failed = true;
            if (pre && !hip_ok(rails_panel_random(b.ctx, b.P.panel(), t0 + w, 1), "rails_panel_random")) b.failed = true;
            b.P.resize(t0);
            if (!pre) {
                b.absorb_tail(w, out.cptr(j0));
                continue;
            }
            // the block [A*W | q] is absorbed as one; its last column's coordinates are kept for the coming random()
            std::vector<double> coef((size_t)b.row_cap * (w + 1), 0.0);
            b.absorb_tail(w + 1, coef.data());
            for (int j = 0; j < w; ++j) memcpy(out.cptr(j0 + j), coef.data() + (size_t)j * b.row_cap, sizeof(double) * b.row_cap);
            if (!b.cached_random || b.cached_random->ld != b.row_cap) b.cached_random = b.new_store(1, true);
            memcpy(b.cached_random->col(0), coef.data() + (size_t)w * b.row_cap, sizeof(double) * b.row_cap);
            b.cached_valid = true;
            b.n_prefetched++;
        }
        return out;
    }
};

} // namespace rails

#endif
