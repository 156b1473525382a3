// SolverOps specialisation for the HIP backend: the three places where the solver's member-by-member
// sequences are replaced by fused device work (see rails/LyapunovSolver.hpp).
#ifndef RAILS_HIPSOLVEROPS_HPP
#define RAILS_HIPSOLVEROPS_HPP

#include "rails/HipWrappers.hpp"
#include "rails/LyapunovSolver.hpp"

namespace rails
{

template <>
struct SolverOps<HipOperatorWrapper, HipMultiVectorWrapper, HostDenseMatrix> {
    typedef Solver<HipOperatorWrapper, HipMultiVectorWrapper, HostDenseMatrix> SolverT;

    struct Lanczos {
        HostDenseMatrix eigenvalues;
        HostDenseMatrix v;                   // Ritz vectors in the Lanczos basis (steps x steps)
        HipMultiVectorWrapper eigenvectors;  // only filled by the generic fallback
        rails_ctx *ctx = nullptr;
        int steps = 0;
        bool fused = false;

        // V <- [V, Q * v(:, indices)]: the selected Ritz vectors are formed straight in V's tail
        // (replaces `eigenvectors = Q * v` and the push_back loop, src/LyapunovSolver.hpp:443,338-339)
        void append_to(HipMultiVectorWrapper &V, std::vector<int> const &indices, int count) const
        {
            if (!fused) {
                for (int i = 0; i < count; i++) V.push_back(eigenvectors.view(indices[i]));
                return;
            }
            if (count <= 0) return;
            int n = V.N();
            V.resize(n + count);
            std::vector<double> S((size_t)steps * count);
            for (int j = 0; j < count; ++j)
                for (int i = 0; i < steps; ++i) S[i + (size_t)j * steps] = v(i, indices[j]);
            hip_ok(rails_lanczos_vectors(ctx, S.data(), steps, count, V.panel(), V.offset() + n), "rails_lanczos_vectors");
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

    static int lanczos(SolverT &solver, HipMultiVectorWrapper const &AV, HipMultiVectorWrapper const &MV, HostDenseMatrix const &T, int max_iter,
                       Lanczos &out)
    {
        out.ctx = AV.context();
        bool can_fuse = !solver.B().is_matrix() && AV.N() <= 512 && solver.B().vector().N() <= 128 && ((AV.offset() | MV.offset()) & 1) == 0 &&
                        (solver.B().vector().offset() & 1) == 0;
        if (!can_fuse) {
            HostDenseMatrix H(max_iter + 1, max_iter + 1);
            out.eigenvalues = HostDenseMatrix(max_iter, 1);
            out.fused = false;
            return solver.resid_lanczos(AV, MV, T, H, out.eigenvectors, out.eigenvalues, max_iter);
        }
        HipMultiVectorWrapper const &B = solver.B().vector();
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
        out.fused = true;
        return 0;
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
