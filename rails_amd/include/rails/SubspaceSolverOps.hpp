// SolverOps specialisation for the coordinate-space back end (rails/SubspaceWrappers.hpp): the solver's member-by-member
// sequences are already cheap there (host coefficient operations); the only customisation is what happens at a restart.
#ifndef RAILS_SUBSPACESOLVEROPS_HPP
#define RAILS_SUBSPACESOLVEROPS_HPP

#include "rails/LyapunovSolver.hpp"
#include "rails/SubspaceWrappers.hpp"

namespace rails
{

// the restart products are coefficient GEMMs; afterwards the basis is re-based on what is still alive
template <>
struct SolverOps<SubspaceOperator, SubspaceMultiVector, HostDenseMatrix> {
    typedef Solver<SubspaceOperator, SubspaceMultiVector, HostDenseMatrix> SolverT;
    struct State {
    };
    struct Lanczos {
        HostDenseMatrix eigenvalues;
        SubspaceMultiVector eigenvectors;
        void append_to(SubspaceMultiVector &V, std::vector<int> const &indices, int count) const
        {
            for (int i = 0; i < count; i++) V.push_back(eigenvectors.view(indices[i]));
        }
    };
    static SubspaceMultiVector apply_append(SubspaceOperator const &A, SubspaceMultiVector const &W, SubspaceMultiVector &AV)
    {
        SubspaceMultiVector AW = A * W;
        AV.push_back(AW);
        return AW;
    }
    static int lanczos(SolverT &solver, State &, SubspaceMultiVector const &AV, SubspaceMultiVector const &MV, HostDenseMatrix const &T,
                       HostDenseMatrix const &, SubspaceMultiVector const &, int max_iter, Lanczos &out)
    {
        HostDenseMatrix H(max_iter + 1, max_iter + 1);
        out.eigenvalues = HostDenseMatrix(max_iter, 1);
        return solver.resid_lanczos(AV, MV, T, H, out.eigenvectors, out.eigenvalues, max_iter);
    }
    static void multiply_inplace(SubspaceMultiVector &V, HostDenseMatrix const &X)
    {
        V.view(0, X.N() - 1) = V * X;
        V.resize(X.N());
        V.discard_unused_columns();
    }
    static void on_restart(State &, HostDenseMatrix const &, SubspaceMultiVector &V, SubspaceMultiVector &)
    {
        if (V.basis()) V.basis()->compress();
    }
};

typedef Solver<SubspaceOperator, SubspaceMultiVector, HostDenseMatrix> SubspaceSolver;

} // namespace rails

#endif
