// Compile-only check (g++ -fsyntax-only, nothing is linked or run): the reference's own header-only
// RAILS::Solver (src/LyapunovSolver.hpp, included from where it lies) instantiates with the HIP backend's
// wrapper classes as its three template parameters -- i.e. the wrappers satisfy the duck-typed plug-in contract
// (src/LyapunovSolverDecl.hpp:9-51) and drop in next to StlWrapper / Epetra_*Wrapper.
#include <cstring>
#include <iostream> // src/LyapunovSolver.hpp uses std::cout without including <iostream>
#include <map>
#include <string>

#include "src/LyapunovSolver.hpp"

#include "rails/HipWrappers.hpp"
#include "rails/SubspaceWrappers.hpp"

struct ParameterList { // the mock of test/LyapunovSolver_test.cpp:160-179
    std::map<std::string, double> p;
    template <typename T>
    T get(std::string const &name, T def)
    {
        auto it = p.find(name);
        return it == p.end() ? def : (T)it->second;
    }
};

typedef RAILS::Solver<rails::HipOperatorWrapper, rails::HipMultiVectorWrapper, rails::HostDenseMatrix> RefSolverOnHip;
template class RAILS::Solver<rails::HipOperatorWrapper, rails::HipMultiVectorWrapper, rails::HostDenseMatrix>;

int use(rails::HipOperatorWrapper const &A, rails::HipMultiVectorWrapper const &B)
{
    RefSolverOnHip with_multivector_B(A, B, A); // B as a MultiVector
    RefSolverOnHip with_operator_B(A, A, A);    // B as a Matrix (src/MatrixOrMultiVectorWrapper.hpp:17)
    ParameterList params;
    with_multivector_B.set_parameters(params);
    rails::HipMultiVectorWrapper V;
    rails::HostDenseMatrix T;
    return with_multivector_B.solve(V, T) + with_operator_B.solve(V, T);
}

// the coordinate-space back end: the same contract, a different representation (rails/SubspaceWrappers.hpp)
typedef RAILS::Solver<rails::SubspaceOperator, rails::SubspaceMultiVector, rails::HostDenseMatrix> RefSolverOnSubspace;
template class RAILS::Solver<rails::SubspaceOperator, rails::SubspaceMultiVector, rails::HostDenseMatrix>;

int use_subspace(rails::SubspaceOperator const &A, rails::SubspaceMultiVector const &B)
{
    RefSolverOnSubspace solver(A, B, A);
    ParameterList params;
    solver.set_parameters(params);
    rails::SubspaceMultiVector V;
    rails::HostDenseMatrix T;
    return solver.solve(V, T);
}
