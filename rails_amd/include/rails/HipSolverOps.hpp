// SolverOps specialisation for the HIP backend: the places where the solver's member-by-member sequences are
// replaced by fused device work (see rails/LyapunovSolver.hpp).
//
// Residual Lanczos comes in two forms:
//  * fused   (default): rails_resid_lanczos, ONE pass over [AV V B] per Lanczos step (lanczos.hip);
//  * projected (opt-in, Solver::set_projected_lanczos): the same Lanczos recurrence carried in coefficient space.
//    With W~ = [AV - V*VAV, B - V*(V'B)] (the parts of AV and B orthogonal to V) the Galerkin condition
//    VAV*T + T*VAV' + VBV = 0 makes the V-V block of the residual vanish, so
//        R = [V Qw] K [V Qw]',   K = [[0, N], [N', D]],   W~ = Qw*Rw (Cholesky of the Gram matrix of W~),
//        N = [T | V'B] * Rw',    D = Rw * diag(0_k, I_p) * Rw'.
//    Lanczos is invariant under an orthonormal change of basis, so running the reference's recurrence on
//    diag(K, 0) from the coordinates of the same random start vector q0 in the basis [V, Qw, q^] reproduces the same
//    tridiagonal matrix up to rounding -- with ONE pass over the panels (for [AV V B]'q0) and an incremental
//    w x k Gram block per trip instead of L+1 passes.  The Ritz vectors are mapped back with three panel GEMMs.
//    Only for M = I; the generalized form and any trip whose Gram matrix is not numerically positive definite use
//    the fused kernel.
#ifndef RAILS_HIPSOLVEROPS_HPP
#define RAILS_HIPSOLVEROPS_HPP

#include <cstdlib>

#include "rails/HipWrappers.hpp"
#include "rails/LyapunovSolver.hpp"

namespace rails
{

template <>
struct SolverOps<HipOperatorWrapper, HipMultiVectorWrapper, HostDenseMatrix> {
    typedef Solver<HipOperatorWrapper, HipMultiVectorWrapper, HostDenseMatrix> SolverT;

    // Gram blocks of the projected form, maintained incrementally over the trips of one solve
    struct State {
        std::vector<double> SAA, SAB, SBB; // AV'AV (ld = cap), AV'B (ld = cap), B'B (p x p)
        int cap = 0, kc = 0, p = -1;
        long projected_trips = 0, fused_trips = 0;
        bool below_noise_floor = false; // the projected form has reached the rounding level of its Gram differences
    };

    struct Lanczos {
        HostDenseMatrix eigenvalues;
        HostDenseMatrix v;                  // Ritz vectors in the Lanczos basis (steps x steps)
        HipMultiVectorWrapper eigenvectors; // only filled by the generic fallback
        rails_ctx *ctx = nullptr;
        int steps = 0;
        int mode = 0; // 0 generic members, 1 fused kernel, 2 projected
        // projected form: what is needed to map small-space Ritz vectors back
        int k = 0, p = 0, r = 0;                        // r = numerical rank of W~
        std::vector<double> Y, Rw, cV, cW, d, VAV, Bv; // Y (k+r+1) x steps; Rw r x r upper; d column scales; VAV k x k; Bv k x p
        std::vector<int> piv;                          // pivoted columns of W~ spanning Qw
        double cq = 0.0, nq = 1.0;
        const HipMultiVectorWrapper *AVp = nullptr, *Bp = nullptr;

        // V <- [V, selected Ritz vectors]
        void append_to(HipMultiVectorWrapper &V, std::vector<int> const &indices, int count) const
        {
            if (mode == 0) {
                for (int i = 0; i < count; i++) V.push_back(eigenvectors.view(indices[i]));
                return;
            }
            if (count <= 0) return;
            int n0 = V.N();
            if (mode == 1) { // Q * v(:, indices) formed straight in V's tail (src/LyapunovSolver.hpp:443,338-339)
                V.resize(n0 + count);
                std::vector<double> S((size_t)steps * count);
                for (int j = 0; j < count; ++j)
                    for (int i = 0; i < steps; ++i) S[i + (size_t)j * steps] = v(i, indices[j]);
                hip_ok(rails_lanczos_vectors(ctx, S.data(), steps, count, V.panel(), V.offset() + n0), "rails_lanczos_vectors");
                return;
            }
            // projected: E = AV*Ua + B*Ub + V*(Zv - cV*g - VAV*Ua - Bv*Ub) + q0*(g/|q0|),  U = P D^-1 R11^-1 (Zw - cW*g)
            const int kp = k + p, n = k + r, w = count;
            std::vector<double> Z((size_t)(n + 1) * w, 0.0);
            for (int j = 0; j < w; ++j)
                for (int i = 0; i <= n; ++i) {
                    double s = 0.0;
                    for (int l = 0; l < steps; ++l) s += Y[i + (size_t)l * (n + 1)] * v(l, indices[j]);
                    Z[i + (size_t)j * (n + 1)] = s;
                }
            std::vector<double> U((size_t)kp * w, 0.0), g(w), CV((size_t)k * w), x(r);
            for (int j = 0; j < w; ++j) {
                g[j] = (cq > 1e-12) ? Z[n + (size_t)j * (n + 1)] / cq : 0.0;
                for (int i = r - 1; i >= 0; --i) { // back substitution with R11
                    double s = Z[k + i + (size_t)j * (n + 1)] - cW[i] * g[j];
                    for (int l = i + 1; l < r; ++l) s -= Rw[i + (size_t)l * r] * x[l];
                    x[i] = s / Rw[i + (size_t)i * r];
                }
                for (int i = 0; i < r; ++i) U[piv[i] + (size_t)j * kp] = x[i] / d[piv[i]];
                for (int i = 0; i < k; ++i) {
                    double s = Z[i + (size_t)j * (n + 1)] - cV[i] * g[j];
                    for (int l = 0; l < k; ++l) s -= VAV[i + (size_t)l * k] * U[l + (size_t)j * kp];
                    for (int l = 0; l < p; ++l) s -= Bv[i + (size_t)l * k] * U[k + l + (size_t)j * kp];
                    CV[i + (size_t)j * k] = s;
                }
            }
            V.resize(n0 + w);
            std::vector<double> gamma(w);
            for (int j = 0; j < w; ++j) gamma[j] = g[j] / nq;
            hip_ok(rails_lanczos_vectors(ctx, gamma.data(), 1, w, V.panel(), V.offset() + n0), "rails_lanczos_vectors"); // E = q0*gamma
            hip_ok(rails_panel_gemm(ctx, 1.0, AVp->panel(), AVp->offset(), k, U.data(), kp, w, 1.0, V.panel(), V.offset() + n0), "rails_panel_gemm");
            hip_ok(rails_panel_gemm(ctx, 1.0, V.panel(), V.offset(), k, CV.data(), k, w, 1.0, V.panel(), V.offset() + n0), "rails_panel_gemm");
            if (p > 0)
                hip_ok(rails_panel_gemm(ctx, 1.0, Bp->panel(), Bp->offset(), p, U.data() + k, kp, w, 1.0, V.panel(), V.offset() + n0), "rails_panel_gemm");
        }
    };

    // A*W lands directly in AV's tail; the returned multivector aliases those columns
    static HipMultiVectorWrapper apply_append(HipOperatorWrapper const &A, HipMultiVectorWrapper const &W, HipMultiVectorWrapper &AV)
    {
        int n = AV.N(), wn = W.N();
        AV.resize(n + wn);
        A.apply_into(W, AV, n);
        return AV.view(n, n + wn - 1);
    }

    static void on_restart(State &st, HostDenseMatrix const &X, HipMultiVectorWrapper &, HipMultiVectorWrapper &)
    {
        if (st.kc <= 0 || X.M() != st.kc) { // nothing cached (or inconsistent): recompute from scratch next time
            st.kc = 0;
            return;
        }
        const int k = st.kc, r = X.N(), cap = st.cap;
        HostDenseMatrix S(k, k), SB(k, std::max(st.p, 1));
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < k; ++i) S(i, j) = st.SAA[i + (size_t)j * cap];
        for (int j = 0; j < st.p; ++j)
            for (int i = 0; i < k; ++i) SB(i, j) = st.SAB[i + (size_t)j * cap];
        HostDenseMatrix S2 = X.transpose() * (S * X);
        for (int j = 0; j < r; ++j)
            for (int i = 0; i < r; ++i) st.SAA[i + (size_t)j * cap] = S2(i, j);
        if (st.p > 0) {
            HostDenseMatrix SB2 = X.transpose() * SB;
            for (int j = 0; j < st.p; ++j)
                for (int i = 0; i < r; ++i) st.SAB[i + (size_t)j * cap] = SB2(i, j);
        }
        st.kc = r;
    }

    // pivots of the scaled Gram matrix below this are rounding noise of the differences it is built from
    static double projected_rank_tolerance()
    {
        static const double tol = [] {
            const char *e = getenv("RAILS_PROJECTED_RANK_TOL");
            return e ? atof(e) : 1e-13;
        }();
        return tol;
    }

    static double projected_noise_margin()
    {
        static const double f = [] {
            const char *e = getenv("RAILS_PROJECTED_NOISE_MARGIN");
            return e ? atof(e) : 30.0;
        }();
        return f;
    }

    static int lanczos_fused(SolverT &solver, HipMultiVectorWrapper const &AV, HipMultiVectorWrapper const &MV, HostDenseMatrix const &T,
                             int max_iter, Lanczos &out)
    {
        HipMultiVectorWrapper const &B = solver.B().panel();
        HostDenseMatrix H(max_iter + 1, max_iter + 1);
        HostDenseMatrix Tc = T.copy(); // contiguous copy, also drops a transpose flag
        int steps = 0;
        if (!hip_ok(rails_resid_lanczos(out.ctx, AV.panel(), AV.offset(), MV.panel(), MV.offset(), AV.N(), (double *)Tc, Tc.LDA(), B.panel(),
                                        B.offset(), B.N(), max_iter, (double *)H, H.LDA(), &steps),
                    "rails_resid_lanczos"))
            return -1;
        H.resize(steps, steps);
        out.v = HostDenseMatrix(steps, steps);
        out.eigenvalues = HostDenseMatrix(max_iter, 1);
        H.eigs(out.v, out.eigenvalues); // :441
        out.steps = steps;
        out.mode = 1;
        return 0;
    }

    // returns 0 on success, 1 when the projected form is not applicable this trip (caller falls back)
    static int lanczos_projected(SolverT &solver, State &st, HipMultiVectorWrapper const &AV, HipMultiVectorWrapper const &V,
                                 HostDenseMatrix const &T, HostDenseMatrix const &VAVm, HipMultiVectorWrapper const &BV, int L, Lanczos &out)
    {
        HipMultiVectorWrapper const &B = solver.B().panel();
        const int k = AV.N(), p = B.N(), kp = k + p;
        if (k <= 0 || V.N() != k || VAVm.M() != k || BV.N() != k || !BV.replicated()) return 1;
        // ---- Gram blocks AV'AV, AV'B (incremental), B'B (once) -----------------------------------------------
        if (st.p != p || st.cap < AV.capacity()) {
            int cap = std::max(AV.capacity(), k);
            std::vector<double> nAA((size_t)cap * cap, 0.0), nAB((size_t)cap * std::max(p, 1), 0.0);
            if (st.p == p && st.kc > 0)
                for (int j = 0; j < st.kc; ++j) {
                    for (int i = 0; i < st.kc; ++i) nAA[i + (size_t)j * cap] = st.SAA[i + (size_t)j * st.cap];
                }
            if (st.p == p && st.kc > 0)
                for (int j = 0; j < p; ++j)
                    for (int i = 0; i < st.kc; ++i) nAB[i + (size_t)j * cap] = st.SAB[i + (size_t)j * st.cap];
            if (st.p != p) {
                st.kc = 0;
                st.SBB.assign((size_t)std::max(p, 1) * std::max(p, 1), 0.0);
                if (p > 0 && !hip_ok(rails_gram(out.ctx, B.panel(), B.offset(), p, B.panel(), B.offset(), p, st.SBB.data(), p), "rails_gram")) return -1;
            }
            st.SAA.swap(nAA);
            st.SAB.swap(nAB);
            st.cap = cap;
            st.p = p;
        }
        if (st.kc > k) st.kc = 0;
        if (st.kc < k) {
            const int c0 = st.kc, wn = k - c0, cap = st.cap;
            std::vector<double> G((size_t)wn * k);
            if (!hip_ok(rails_gram(out.ctx, AV.panel(), AV.offset() + c0, wn, AV.panel(), AV.offset(), k, G.data(), wn), "rails_gram")) return -1;
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < wn; ++i) {
                    st.SAA[(c0 + i) + (size_t)j * cap] = G[i + (size_t)j * wn];
                    st.SAA[j + (size_t)(c0 + i) * cap] = G[i + (size_t)j * wn];
                }
            if (p > 0) {
                std::vector<double> GB((size_t)wn * p);
                if (!hip_ok(rails_gram(out.ctx, AV.panel(), AV.offset() + c0, wn, B.panel(), B.offset(), p, GB.data(), wn), "rails_gram")) return -1;
                for (int j = 0; j < p; ++j)
                    for (int i = 0; i < wn; ++i) st.SAB[(c0 + i) + (size_t)j * cap] = GB[i + (size_t)j * wn];
            }
            st.kc = k;
        }
        // ---- small matrices -----------------------------------------------------------------------------------
        out.k = k;
        out.p = p;
        out.VAV.assign((size_t)k * k, 0.0);
        out.Bv.assign((size_t)k * std::max(p, 1), 0.0);
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < k; ++i) out.VAV[i + (size_t)j * k] = VAVm(i, j);
        {
            const double *bv = BV.host_data(); // p x k column-major: (B'V)(i, j)
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < p; ++i) out.Bv[j + (size_t)i * k] = bv[i + (size_t)j * p];
        }
        // Sw = W~'W~,  W~ = [AV - V VAV, B - V Bv], columns scaled by the norms d of the un-projected columns
        std::vector<double> Sw((size_t)kp * kp, 0.0), d(kp);
        {
            std::vector<double> VtV((size_t)k * k), VtB((size_t)k * std::max(p, 1)), BtB((size_t)std::max(p, 1) * std::max(p, 1));
            rails_dgemm('T', 'N', k, k, k, 1.0, out.VAV.data(), k, out.VAV.data(), k, 0.0, VtV.data(), k);
            if (p > 0) {
                rails_dgemm('T', 'N', k, p, k, 1.0, out.VAV.data(), k, out.Bv.data(), k, 0.0, VtB.data(), k);
                rails_dgemm('T', 'N', p, p, k, 1.0, out.Bv.data(), k, out.Bv.data(), k, 0.0, BtB.data(), p);
            }
            const int cap = st.cap;
            for (int i = 0; i < k; ++i) d[i] = std::sqrt(st.SAA[i + (size_t)i * cap]);
            for (int i = 0; i < p; ++i) d[k + i] = std::sqrt(st.SBB[i + (size_t)i * p]);
            for (int i = 0; i < kp; ++i)
                if (!(d[i] > 0.0)) d[i] = 1.0;
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) Sw[i + (size_t)j * kp] = (st.SAA[i + (size_t)j * cap] - VtV[i + (size_t)j * k]) / (d[i] * d[j]);
            for (int j = 0; j < p; ++j)
                for (int i = 0; i < k; ++i) {
                    double sv = (st.SAB[i + (size_t)j * cap] - VtB[i + (size_t)j * k]) / (d[i] * d[k + j]);
                    Sw[i + (size_t)(k + j) * kp] = sv;
                    Sw[(k + j) + (size_t)i * kp] = sv;
                }
            for (int j = 0; j < p; ++j)
                for (int i = 0; i < p; ++i)
                    Sw[(k + i) + (size_t)(k + j) * kp] = (st.SBB[i + (size_t)j * p] - BtB[i + (size_t)j * p]) / (d[k + i] * d[k + j]);
        }
        // Cholesky with complete pivoting, P'Sw P = Rw'Rw, stopped where the remaining columns of W~ are below the rounding
        // level of the Gram differences above (B lies in span(V) until the first restart, so Sw is rank deficient by
        // construction): W~ D^-1 P = Qw Rw(0:r, :)
        std::vector<int> piv(kp);
        int r = 0, info = 0;
        rails_dpstrf('U', kp, Sw.data(), kp, piv.data(), &r, projected_rank_tolerance(), &info);
        if (info < 0 || r <= 0) return 1;
        out.r = r;
        out.piv = piv;
        out.d = d;
        out.Rw.assign((size_t)r * r, 0.0); // R11
        for (int j = 0; j < r; ++j)
            for (int i = 0; i <= j; ++i) out.Rw[i + (size_t)j * r] = Sw[i + (size_t)j * kp];
        // G (r x kp): W~ = Qw G
        std::vector<double> G((size_t)r * kp, 0.0);
        for (int j = 0; j < kp; ++j) {
            const int cj = piv[j];
            for (int i = 0; i < std::min(r, j + 1); ++i) G[i + (size_t)cj * r] = Sw[i + (size_t)j * kp] * d[cj];
        }
        // K = [[0, N], [N', D]],  N = [T | Bv] G' (k x r),  D = G(:, k:kp) G(:, k:kp)' (r x r)
        const int n = k + r;
        std::vector<double> TB((size_t)k * kp), N((size_t)k * r), K((size_t)n * n, 0.0);
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < k; ++i) TB[i + (size_t)j * k] = T(i, j);
        for (int j = 0; j < p; ++j)
            for (int i = 0; i < k; ++i) TB[i + (size_t)(k + j) * k] = out.Bv[i + (size_t)j * k];
        rails_dgemm('N', 'T', k, r, kp, 1.0, TB.data(), k, G.data(), r, 0.0, N.data(), k);
        // what the columns of W~ dropped (or mis-resolved) at the rank tolerance can contribute to N: the level below which
        // eigenvalue estimates of this form are rounding noise
        double noise = 0.0;
        for (int j = 0; j < kp; ++j) {
            double cs = 0.0;
            for (int i = 0; i < k; ++i) cs += TB[i + (size_t)j * k] * TB[i + (size_t)j * k];
            noise += cs * d[j] * d[j];
        }
        noise = std::sqrt(noise * projected_rank_tolerance());
        for (int j = 0; j < r; ++j)
            for (int i = 0; i < k; ++i) {
                K[i + (size_t)(k + j) * n] = N[i + (size_t)j * k];
                K[(k + j) + (size_t)i * n] = N[i + (size_t)j * k];
            }
        if (p > 0) {
            std::vector<double> D((size_t)r * r);
            rails_dgemm('N', 'T', r, r, p, 1.0, G.data() + (size_t)k * r, r, G.data() + (size_t)k * r, r, 0.0, D.data(), r);
            for (int j = 0; j < r; ++j)
                for (int i = 0; i < r; ++i) K[(k + i) + (size_t)(k + j) * n] = D[i + (size_t)j * r];
        }
        // ---- start vector: ONE pass over the panels -----------------------------------------------------------
        std::vector<double> sums((size_t)2 * k + p + 1);
        if (!hip_ok(rails_lanczos_start(out.ctx, AV.panel(), AV.offset(), V.panel(), V.offset(), k, B.panel(), B.offset(), p, sums.data()),
                    "rails_lanczos_start"))
            return -1;
        const double *yA = sums.data(), *yV = sums.data() + k, *yB = sums.data() + 2 * k;
        out.nq = std::sqrt(sums[2 * k + p]);
        if (!(out.nq > 0.0)) return -1;
        std::vector<double> c((size_t)n + 1, 0.0), wv(kp);
        for (int i = 0; i < k; ++i) {
            double s = yA[i];
            for (int l = 0; l < k; ++l) s -= out.VAV[l + (size_t)i * k] * yV[l]; // VAV' yV
            wv[i] = s / out.nq;
        }
        for (int i = 0; i < p; ++i) {
            double s = yB[i];
            for (int l = 0; l < k; ++l) s -= out.Bv[l + (size_t)i * k] * yV[l]; // Bv' yV
            wv[k + i] = s / out.nq;
        }
        // cW = Qw'q0/|q0| = R11^-T (D^-1 w)(piv(0:r))   (forward substitution)
        for (int i = 0; i < r; ++i) {
            double s = wv[piv[i]] / d[piv[i]];
            for (int l = 0; l < i; ++l) s -= out.Rw[l + (size_t)i * r] * c[k + l];
            c[k + i] = s / out.Rw[i + (size_t)i * r];
        }
        out.cV.assign(k, 0.0);
        out.cW.assign(c.begin() + k, c.begin() + k + r);
        double rest = 1.0;
        for (int i = 0; i < k; ++i) {
            out.cV[i] = c[i] = yV[i] / out.nq;
            rest -= c[i] * c[i];
        }
        for (int i = 0; i < r; ++i) rest -= c[k + i] * c[k + i];
        out.cq = rest > 0.0 ? std::sqrt(rest) : 0.0;
        c[n] = out.cq;
        // ---- the reference's Lanczos recurrence (src/LyapunovSolver.hpp:380-434) on diag(K, 0) ---------------
        const int n1 = n + 1;
        out.Y.assign((size_t)n1 * (L + 1), 0.0);
        for (int i = 0; i < n1; ++i) out.Y[i] = c[i];
        HostDenseMatrix H(L + 1, L + 1);
        H = 0.0;
        double alpha = 0.0, beta = 0.0;
        int iter = 0;
        for (int it = 0; it < L; ++it) {
            double *y = &out.Y[(size_t)iter * n1], *z = &out.Y[(size_t)(iter + 1) * n1];
            rails_dgemm('N', 'N', n, 1, n, 1.0, K.data(), n, y, n1, 0.0, z, n1);
            z[n] = 0.0;
            alpha = 0.0;
            for (int i = 0; i < n1; ++i) alpha += z[i] * y[i];
            H(iter, iter) = alpha;
            for (int i = 0; i < n1; ++i) z[i] -= alpha * y[i];
            if (iter > 0) {
                const double *ym = &out.Y[(size_t)(iter - 1) * n1];
                for (int i = 0; i < n1; ++i) z[i] -= beta * ym[i];
            }
            beta = 0.0;
            for (int i = 0; i < n1; ++i) beta += z[i] * z[i];
            beta = std::sqrt(beta);
            if (beta < 1e-14) {
                iter++;
                break;
            }
            H(iter + 1, iter) = beta;
            H(iter, iter + 1) = beta;
            const double ib = 1.0 / beta;
            for (int i = 0; i < n1; ++i) z[i] *= ib;
            iter++;
        }
        H.resize(iter, iter);
        out.v = HostDenseMatrix(iter, iter);
        out.eigenvalues = HostDenseMatrix(L, 1);
        H.eigs(out.v, out.eigenvalues);
        {
            double res = 0.0;
            for (int i = 0; i < iter; ++i) res = std::max(res, std::abs(out.eigenvalues(i)));
            if (!(res > projected_noise_margin() * noise)) {
                st.below_noise_floor = true;
                return 1;
            }
        }
        out.steps = iter;
        out.mode = 2;
        out.AVp = &AV;
        out.Bp = &B;
        return 0;
    }

    static int lanczos(SolverT &solver, State &st, HipMultiVectorWrapper const &AV, HipMultiVectorWrapper const &MV, HostDenseMatrix const &T,
                       HostDenseMatrix const &VAV, HipMultiVectorWrapper const &BV, int max_iter, Lanczos &out)
    {
        out.ctx = AV.context();
        bool can_fuse = !solver.B().given_as_operator() && AV.N() <= 512 && solver.B().panel().N() <= 128 && ((AV.offset() | MV.offset()) & 1) == 0 &&
                        (solver.B().panel().offset() & 1) == 0;
        if (!can_fuse) {
            HostDenseMatrix H(max_iter + 1, max_iter + 1);
            out.eigenvalues = HostDenseMatrix(max_iter, 1);
            out.mode = 0;
            return solver.resid_lanczos(AV, MV, T, H, out.eigenvectors, out.eigenvalues, max_iter);
        }
        if (solver.projected_lanczos() && !solver.mass_matrix_in_use() && !st.below_noise_floor) {
            uint64_t seed = 0, stream = 0;
            rails_ctx_rng_state(out.ctx, &seed, &stream);
            int rc = lanczos_projected(solver, st, AV, MV, T, VAV, BV, max_iter, out);
            if (rc == 0) {
                st.projected_trips++;
                return 0;
            }
            if (rc < 0) return rc;
            rails_ctx_set_seed(out.ctx, seed, stream); // the fused form repeats the draw of the start vector
        }
        st.fused_trips++;
        return lanczos_fused(solver, AV, MV, T, max_iter, out);
    }

    // V <- V * X in place (row-local panel GEMM), then shrink to X.N() columns
    static void multiply_inplace(HipMultiVectorWrapper &V, HostDenseMatrix const &X)
    {
        if (V.replicated() || X.N() > 256 || X.N() <= 0 || X.M() != V.N()) {
            V.view(0, X.N() - 1) = V * X;
            V.resize(X.N());
            return;
        }
        hip_ok(rails_panel_gemm(V.context(), 1.0, V.panel(), V.offset(), V.N(), (double *)X, X.LDA(), X.N(), 0.0, V.panel(), V.offset()),
               "rails_panel_gemm");
        V.resize(X.N());
    }
};

typedef Solver<HipOperatorWrapper, HipMultiVectorWrapper, HostDenseMatrix> HipSolver;

} // namespace rails

#endif
