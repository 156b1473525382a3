// rails::Solver<Matrix, MultiVector, DenseMatrix> -- the RAILS outer loop for the duck-typed backend contract of the
// reference (src/LyapunovSolverDecl.hpp:9-51).  The public surface is the reference's -- constructor (A, B, M),
// set_parameters, solve(V, T), dense_solve, resid_lanczos, compute_restart_vectors, the same parameter names and defaults
// (src/LyapunovSolver.hpp:27-36,76-87) and return codes (0 converged / -1 stopped without converging / 1 trips used up,
// :239-240,345; set_parameters 0 / 1, :89-97) -- and every DECISION is the reference's, cited where it is taken:
// convergence (:223), when to stop (:224-241), when to shrink the space (:245-247), how many vectors to add (:306-307),
// which (:336-339).  The organisation is this library's own: a solve is a Run object that owns the search space (V, A V, M V,
// B'V), the reduced matrices (V'AV, V'BB'V, V'MV) and the settings, and advances by trips of four steps --
//
//     border    A (and M) applied to the columns added last; the reduced matrices get their new border
//     reduce    the projected Lyapunov equation, on the host (rails_sb03md)
//     estimate  largest Ritz values / vectors of the residual operator (Lanczos)
//     adapt     stop, shrink the space to the dominant part of the solution, or add the leading residual directions
//
// Three customisation points (struct SolverOps, specialised in rails/HipSolverOps.hpp and rails/SubspaceSolverOps.hpp) let a
// backend replace member-by-member sequences by fused work: apply_append (A W straight into AV's tail), lanczos (the
// residual estimate) and multiply_inplace (V <- V X when the space shrinks).
// Unlike the reference's C++ (which stores M and never reads it, src/LyapunovSolver.hpp:26) a mass matrix can be switched
// on with use_mass_matrix(true); the generalized iteration follows matlab/RAILSsolver.m:368-395,499-504.
#ifndef RAILS_LYAPUNOVSOLVER_HPP
#define RAILS_LYAPUNOVSOLVER_HPP

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <numeric>
#include <string>
#include <utility>
#include <vector>

#include "rails_hip.h"

namespace rails
{

// The right-hand side factor B, given either as an operator (Matrix) or as a tall panel (MultiVector) -- the two forms the
// reference accepts through its B adaptor (src/MatrixOrMultiVectorWrapper.hpp:7-98).  Both members exist (the contract asks
// for default-constructible types, :11-12); `kind_` says which one is in use.
template <class Matrix, class MultiVector>
class BOperand
{
public:
    enum Kind { Operator, Panel };

    BOperand(Matrix const &op) : kind_(Operator), op_(op) {}
    BOperand(MultiVector const &panel) : kind_(Panel), panel_(panel) {}

    bool given_as_operator() const { return kind_ == Operator; }
    MultiVector const &panel() const { return panel_; }
    // ||B||_2, the scale of the stopping test (src/LyapunovSolver.hpp:134)
    double norm2() const { return kind_ == Operator ? op_.norm() : panel_.norm(); }
    // B X and B'X
    MultiVector times(MultiVector const &X) const { return kind_ == Operator ? op_ * X : panel_ * X; }
    MultiVector transposed_times(MultiVector const &X) const { return kind_ == Operator ? op_.transpose() * X : panel_.transpose() * X; }

private:
    Kind kind_;
    Matrix op_;
    MultiVector panel_;
};

// One class plays both roles (the Stl-style backends, src/StlWrapper.hpp): there is nothing to choose.
template <class Both>
class BOperand<Both, Both>
{
public:
    BOperand(Both const &b) : b_(b) {}
    bool given_as_operator() const { return false; }
    Both const &panel() const { return b_; }
    double norm2() const { return b_.norm(); }
    Both times(Both const &X) const { return b_ * X; }
    Both transposed_times(Both const &X) const { return b_.transpose() * X; }

private:
    Both b_;
};

// Positions of the `count` entries of largest modulus, in the order the reference's selection produces them
// (src/StlTools.hpp:12-30: std::sort by decreasing |value|; Ritz values come in +- pairs, so ties are common and the order
// among them is whatever that sort yields -- an index sort with the same comparison makes the same moves).
template <class DenseMatrix>
int find_largest_eigenvalues(DenseMatrix const &values, std::vector<int> &positions, int count)
{
    std::vector<int> order(values.M());
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&values](int a, int b) { return std::abs(values(a)) > std::abs(values(b)); });
    positions.insert(positions.end(), order.begin(), order.begin() + count);
    return 0;
}

// A parameter under any of the spellings the reference accepts (src/LyapunovSolver.hpp:40-70): as written, upper case,
// lower case, and capitalised words; later spellings override earlier ones.
template <class ParameterList, class Value>
Value lookup_parameter(ParameterList &params, std::string const &name, Value fallback)
{
    std::string upper(name), lower(name), title(name);
    bool word_start = true;
    for (size_t i = 0; i < name.size(); ++i) {
        const unsigned char ch = (unsigned char)name[i];
        upper[i] = (char)std::toupper(ch);
        lower[i] = (char)std::tolower(ch);
        title[i] = word_start ? (char)std::toupper(ch) : name[i];
        word_start = !std::isalpha(ch);
    }
    Value value = fallback;
    const std::string *spellings[4] = {&name, &upper, &lower, &title};
    for (const std::string *spelling : spellings) value = params.get(*spelling, value);
    return value;
}

// Wall-clock time per part of a trip, under the section names of the reference's profiler (src/Timer.hpp:101-106), so that
// reports stay comparable.  Host time between synchronisation points; the device side is profiled with rocprofv3.
class ScopedTimer
{
    double *slot_;
    std::chrono::steady_clock::time_point start_;

public:
    ScopedTimer(std::map<std::string, double> *sections, const char *name) : slot_(&(*sections)[name]), start_(std::chrono::steady_clock::now()) {}
    ~ScopedTimer() { *slot_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - start_).count(); }
};

template <class Matrix, class MultiVector, class DenseMatrix>
class Solver;

// ---- customisation points; the generic versions go through the backend contract member by member -----------------------
template <class Matrix, class MultiVector, class DenseMatrix>
struct SolverOps {
    typedef Solver<Matrix, MultiVector, DenseMatrix> SolverT;

    // result of one residual estimate: Ritz values and a way to append selected Ritz vectors to V
    struct Lanczos {
        DenseMatrix eigenvalues;
        MultiVector eigenvectors;
        void append_to(MultiVector &V, std::vector<int> const &indices, int count) const
        {
            for (auto it = indices.begin(); it != indices.begin() + count; ++it) V.push_back(eigenvectors.view(*it)); // src/LyapunovSolver.hpp:338-339
        }
    };

    // AV <- [AV, A*W]; returns A*W                                        (src/LyapunovSolver.hpp:146,203)
    static MultiVector apply_append(Matrix const &A, MultiVector const &W, MultiVector &AV)
    {
        MultiVector AW = A * W;
        AV.push_back(AW);
        return AW;
    }

    // per-solve scratch of the backend (nothing for the generic form)
    struct State {
    };

    // VAV and BV (the reduced operator and B'V) are passed along for backends that can use them
    static int lanczos(SolverT &solver, State &, MultiVector const &AV, MultiVector const &MV, DenseMatrix const &T, DenseMatrix const &,
                       MultiVector const &, int max_iter, Lanczos &out)
    {
        DenseMatrix H(max_iter + 1, max_iter + 1);
        out.eigenvalues = DenseMatrix(max_iter, 1);
        return solver.resid_lanczos(AV, MV, T, H, out.eigenvectors, out.eigenvalues, max_iter);
    }

    // after V <- V X and AV <- AV X (:265-291)
    static void on_restart(State &, DenseMatrix const &, MultiVector &, MultiVector &) {}

    // V <- V * X (first X.N() columns), as `V.view(0, X.N()-1) = V * X; V.resize(X.N())`   (:265-266)
    static void multiply_inplace(MultiVector &V, DenseMatrix const &X)
    {
        const int kept = X.N();
        MultiVector product = V * X;
        V.view(0, kept - 1) = product;
        V.resize(kept);
    }
};

template <class Matrix, class MultiVector, class DenseMatrix>
class Solver
{
public:
    typedef SolverOps<Matrix, MultiVector, DenseMatrix> Ops;
    typedef BOperand<Matrix, MultiVector> BType;

    // What a solve is asked to do; names and defaults of the reference's parameters (src/LyapunovSolver.hpp:27-36).
    struct Settings {
        int trip_limit = 1000;            // "Maximum iterations"
        double tolerance = 1e-3;          // "Tolerance": on the residual estimate relative to ||B||^2
        int expand_by = 3;                // "Expand size": residual directions added per trip
        int lanczos_steps = 10;           // "Lanczos iterations"
        int shrink_at = -1;               // "Restart size": space dimension that triggers a shrink (<= 0: never by size)
        int shrink_to = -1;               // "Reduced size": dimension kept (<= 0: whatever passes keep_above)
        int shrink_every = 20;            // "Restart iterations": trips between shrinks (<= 0: never by count)
        double keep_above = 1e-3 * 1e-3;  // "Restart tolerance": |eigenvalue of T| a kept direction must exceed (absolute, :471)
        bool shrink_on_convergence = true; // "Minimize solution space"
        bool warm_start = false;          // "Restart from solution": V holds an orthonormal basis to continue from
    };

    template <class RightHandSide>
    Solver(Matrix const &A, RightHandSide const &B, Matrix const &M) : op_A_(A), rhs_(BType(B)), op_M_(M) {}

    virtual ~Solver() {}

    template <class ParameterList>
    int set_parameters(ParameterList &params)
    {
        Settings &s = settings_;
        s.trip_limit = lookup_parameter(params, "Maximum iterations", s.trip_limit);
        s.tolerance = lookup_parameter(params, "Tolerance", s.tolerance);
        s.expand_by = lookup_parameter(params, "Expand size", s.expand_by);
        s.lanczos_steps = lookup_parameter(params, "Lanczos iterations", s.lanczos_steps);
        s.shrink_at = lookup_parameter(params, "Restart size", s.shrink_at);
        s.shrink_to = lookup_parameter(params, "Reduced size", s.shrink_to);
        s.shrink_every = lookup_parameter(params, "Restart iterations", s.shrink_every);
        s.keep_above = lookup_parameter(params, "Restart tolerance", s.tolerance * 1e-3); // default follows the tolerance (:84)
        s.shrink_on_convergence = lookup_parameter(params, "Minimize solution space", s.shrink_on_convergence);
        s.warm_start = lookup_parameter(params, "Restart from solution", s.warm_start);
        if (s.lanczos_steps <= s.expand_by) { // the estimate must offer more directions than a trip adds (:89-95)
            std::cerr << "rails::Solver: 'Lanczos iterations' (" << s.lanczos_steps << ") has to exceed 'Expand size' (" << s.expand_by << ")" << std::endl;
            return 1;
        }
        return 0;
    }

    // extensions (not in the reference): generalized M, quiet mode, bounded runs, instrumentation
    void use_mass_matrix(bool on) { mass_ = on; }
    // `opts.ortho = 'M'` of matlab/RAILSsolver.m:38-41,538-618: V is kept M-orthonormal (V'MV = I), so the projected equation is the
    // standard one (`lyap(VAV, VBV)`, :384) instead of the generalized one; needs use_mass_matrix(true) and a symmetric definite M
    void use_mass_orthogonalisation(bool on) { ortho_m_ = on; }
    bool mass_orthogonalisation() const { return mass_ && ortho_m_; }
    void set_verbose(bool on) { verbose_ = on; }
    void set_max_trips(int n) { trip_budget_ = n; }
    // residual Lanczos carried in the (2k+p+1)-dimensional coefficient space where the backend supports it
    void set_projected_lanczos(bool on) { projected_lanczos_ = on; }
    bool projected_lanczos() const { return projected_lanczos_; }
    bool mass_matrix_in_use() const { return mass_; }
    void set_trip_callback(std::function<void(int)> cb) { on_trip_ = cb; }
    // asked once per trip: true ends the run with the code of "stopped without converging" (a back end whose device work has failed)
    void set_failure_check(std::function<bool()> f) { broken_ = f; }
    int trips() const { return trips_; }
    std::vector<double> const &residual_history() const { return estimates_; }
    std::map<std::string, double> const &profile() const { return sections_; }
    void reset_profile() { sections_.clear(); }
    BType const &B() const { return rhs_; }
    Settings const &settings() const { return settings_; }
    int lanczos_iterations() const { return settings_.lanczos_steps; }

    // Low-rank solution X = V T V' of A X + X A' + B B' = 0 (A X M' + M X A' + B B' = 0 with a mass matrix)   (src/LyapunovSolver.hpp:100-346)
    int solve(MultiVector &V, DenseMatrix &T)
    {
        Run run(*this, V, T);
        return run.go();
    }

    // A X + X A' + B = 0 for small dense A, B                                (src/LyapunovSolver.hpp:348-365)
    int dense_solve(DenseMatrix const &Ar, DenseMatrix const &Br, DenseMatrix &Xr)
    {
        // rails_sb03md (SLICOT's interface, src/SlicotWrapper.hpp:14-16) overwrites both arguments: work on copies.  With TRANS = 'T' it
        // returns the solution of A X + X A' = scale * C, so the sign flips; info == n + 1 only warns of a perturbed spectrum (:361).
        DenseMatrix schur_work = Ar.copy();
        Xr = Br.copy();
        const int order = Ar.M(), lda = schur_work.LDA(), ldx = Xr.LDA();
        double scale = 1.0;
        int status = 0;
        rails_sb03md('C', 'X', 'N', 'T', order, schur_work, lda, Xr, ldx, &scale, &status);
        const bool failed = status != 0 && status != order + 1;
        Xr *= -1.0;
        if (failed) std::cerr << "rails::Solver: the projected Lyapunov solve failed (sb03md info " << status << ")" << std::endl;
        return status;
    }

    // A X M' + M X A' + B = 0 for small dense A, B, M: `lyap(VAV, VBV, [], VMV)` of matlab/RAILSsolver.m:382 (SLICOT sg03ad through
    // matlab/mex/lyap.c:125-133), here by congruence with the Cholesky factor M = L L':  (L^-1 A L^-T) Y + Y (..)' + L^-1 B L^-T = 0,
    // X = L^-T Y L^-1.  M symmetric definite; a negative definite one (the reference's MOC data, matlab/DataErik/Bp1.co) is the
    // positive case with the signs of A and M flipped, which leaves the equation unchanged.
    int generalized_dense_solve(DenseMatrix const &A, DenseMatrix const &B, DenseMatrix const &Mm, DenseMatrix &X)
    {
        const int k = A.M();
        DenseMatrix L(k, k), Ahat(k, k), Bhat(k, k);
        int info = 0;
        for (double sign : {1.0, -1.0}) {
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) {
                    L(i, j) = sign * 0.5 * (Mm(i, j) + Mm(j, i));
                    Ahat(i, j) = sign * A(i, j);
                    Bhat(i, j) = B(i, j);
                }
            rails_dpotrf('L', k, L, L.LDA(), &info);
            if (info == 0) break;
        }
        if (info) {
            std::cerr << "rails::Solver: the projected mass matrix is not definite (dpotrf info " << info << ")" << std::endl;
            return info;
        }
        for (DenseMatrix *S : {&Ahat, &Bhat}) { // S <- L^-1 S L^-T
            rails_dtrsm('L', 'L', 'N', 'N', k, k, 1.0, L, L.LDA(), *S, S->LDA());
            rails_dtrsm('R', 'L', 'T', 'N', k, k, 1.0, L, L.LDA(), *S, S->LDA());
        }
        DenseMatrix Y;
        const int ret = dense_solve(Ahat, Bhat, Y);
        rails_dtrsm('L', 'L', 'T', 'N', k, k, 1.0, L, L.LDA(), Y, Y.LDA());
        rails_dtrsm('R', 'L', 'N', 'N', k, k, 1.0, L, L.LDA(), Y, Y.LDA());
        X = Y;
        return ret;
    }

    // Ritz pairs of the residual operator R = AV T V' + V T AV' + B B' (never formed) from max_iter steps of Lanczos started at a
    // random unit vector                                                   (src/LyapunovSolver.hpp:367-447).
    // This member goes through the backend contract only; the HIP backends' solve() takes the fused forms (SolverOps::lanczos) but
    // this one stays available and equivalent.  The order of the additions into the new vector is the reference's (:389-402).
    int resid_lanczos(MultiVector const &image, MultiVector const &basis, DenseMatrix const &reduced, DenseMatrix &tridiagonal,
                      MultiVector &ritz_vectors, DenseMatrix &ritz_values, int steps_wanted)
    {
        MultiVector const &AV = image, &V = basis;
        DenseMatrix const &T = reduced;
        DenseMatrix &H = tridiagonal;
        const int max_iter = steps_wanted;
        MultiVector Q(basis, steps_wanted + 1); // the Lanczos vectors, one column per step (+ the one being built)
        Q.resize(1);
        Q.random();
        Q.view(0) /= Q.norm();
        H = 0.0;

        int steps = 0;
        double off_diagonal = 0.0;
        while (steps < max_iter) {
            const int j = steps++;
            Q.resize(j + 2);
            // column j + 1 <- R q_j, term by term (views are written through; a view passed by value would be a deep copy)
            Q.view(j + 1) = rhs_.times(rhs_.transposed_times(Q.view(j)));
            Q.view(j + 1) += AV * DenseMatrix(T * V.dot(Q.view(j)));
            Q.view(j + 1) += V * DenseMatrix(T * AV.dot(Q.view(j)));
            const double diagonal = Q.view(j + 1).dot(Q.view(j))(0, 0);
            H(j, j) = diagonal;
            Q.view(j + 1) -= diagonal * Q.view(j);
            if (j > 0) Q.view(j + 1) -= off_diagonal * Q.view(j - 1);
            off_diagonal = Q.view(j + 1).norm();
            if (off_diagonal < 1e-14) break; // invariant subspace found: the step counts, the recurrence ends (:419-426)
            H(j + 1, j) = off_diagonal;
            H(j, j + 1) = off_diagonal;
            Q.view(j + 1) /= off_diagonal;
        }
        H.resize(steps, steps);
        Q.resize(steps);
        DenseMatrix ritz(steps, steps);
        H.eigs(ritz, ritz_values);
        ritz_vectors = Q * ritz;
        return 0;
    }

    // The directions a shrink keeps: eigenvectors of T for the `num` eigenvalues of largest modulus, those above `tol` (absolute) only
    // (src/LyapunovSolver.hpp:449-482); num <= 0 asks for all.
    int compute_restart_vectors(DenseMatrix &kept_vectors, DenseMatrix const &reduced, int count, double threshold)
    {
        DenseMatrix &X = kept_vectors;
        DenseMatrix const &T = reduced;
        const int num = count;
        const double tol = threshold;
        const int k = T.N();
        DenseMatrix vectors = T.copy(), values(k, 1);
        int info = 0;
        rails_dsyev('V', 'U', k, vectors, vectors.LDA(), values, &info);
        const int wanted = num > 0 ? num : k;
        std::vector<int> leading;
        find_largest_eigenvalues(values, leading, wanted);
        // a column per candidate, filled where the eigenvalue passes; the count of passes is the width that remains (:468-478)
        X = DenseMatrix(k, wanted);
        int kept = 0;
        for (int c = 0; c < wanted; ++c) {
            if (!(std::abs(values(leading[c], 0)) > tol)) continue;
            for (int r = 0; r < k; ++r) X(r, c) = vectors(r, leading[c]);
            ++kept;
        }
        X.resize(k, kept);
        return 0;
    }

private:
    // One solve.  Everything the reference keeps in locals of solve() lives here, named for what it is.
    class Run
    {
        Solver &s_;
        Settings const &opt_;
        MultiVector &V_;  // orthonormal basis of the search space (the caller's, in place)
        DenseMatrix &T_;  // reduced solution
        MultiVector fresh_; // the columns of V the operators have not been applied to yet
        MultiVector AV_, MV_, BtV_; // A V, M V (generalized form), B'V
        DenseMatrix a_, b_, m_;     // V'AV, V'B B'V, V'MV
        typename Ops::State backend_state_;
        int n_;         // problem dimension
        int capacity_;  // columns the panels and reduced matrices have room for
        int trip_ = 0;
        int last_shrink_ = 0;
        bool converged_once_ = false;
        double scale_ = 1.0; // ||B||_2^2

    public:
        Run(Solver &s, MultiVector &V, DenseMatrix &T) : s_(s), opt_(s.settings_), V_(V), T_(T), n_(V.M()) {}

        int go()
        {
            open();
            // RAILS_SOLVER_TRIP_TRACE=x: after every trip that took more than x ms, its time per section on stderr
            static const double trace_ms = getenv("RAILS_SOLVER_TRIP_TRACE") ? atof(getenv("RAILS_SOLVER_TRIP_TRACE")) : 0.0;
            std::map<std::string, double> before;
            auto trip_start = std::chrono::steady_clock::now();
            for (trip_ = 0; trip_ < opt_.trip_limit; ++trip_) {
                if (trace_ms > 0.0) {
                    const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - trip_start).count();
                    if (trip_ > 0 && ms > trace_ms) {
                        std::cerr << "[rails trip " << trip_ << "] " << ms << " ms:";
                        for (auto const &kv : s_.sections_) std::cerr << " " << kv.first << " " << 1e3 * (kv.second - before[kv.first]);
                        std::cerr << std::endl;
                    }
                    before = s_.sections_;
                    trip_start = std::chrono::steady_clock::now();
                }
                s_.notify();
                const int width_before = V_.N();
                if (fresh_.N()) border();
                reduce();
                typename Ops::Lanczos ritz;
                const double estimate = estimate_residual(ritz);
                const bool converged = std::abs(estimate) < opt_.tolerance * scale_; // :223

                // Stop?  Convergence ends the run unless the space is to be minimised first: then the first convergence only
                // triggers a shrink and the second one ends it.  Running out of trips or of dimensions ends it either way (:224-241).
                if (converged || trip_ + 1 >= opt_.trip_limit || V_.N() >= n_) {
                    if (converged && opt_.shrink_on_convergence && !converged_once_)
                        converged_once_ = true;
                    else
                        return finish(converged, estimate);
                }
                if (s_.broken_ && s_.broken_()) { // extension: the back end has latched a failure -- no point in running on (the caller reports it)
                    s_.notify();
                    return -1;
                }
                if (s_.trip_budget_ > 0 && s_.trips_ >= s_.trip_budget_) { // extension: bounded run
                    s_.notify();
                    return 2;
                }
                if (shrink_due(converged)) {
                    shrink(converged);
                    continue;
                }
                expand(ritz, width_before);
            }
            s_.notify();
            return 1;
        }

    private:
        // Room for min(shrink_at or 100, n) columns (:106); a cold start draws one random unit vector (:110-114), a warm start keeps
        // the caller's columns (:116-121).
        void open()
        {
            s_.trips_ = 0;
            s_.estimates_.clear();
            capacity_ = std::max(V_.N(), std::min(opt_.shrink_at > 0 ? opt_.shrink_at : 100, n_));
            if (!opt_.warm_start) {
                V_.resize(capacity_);
                V_.resize(1);
                V_.random();
                if (s_.mass_orthogonalisation())
                    m_orthogonalize(0);
                else
                    V_.orthogonalize();
            } else if (V_.N() != capacity_)
                reserve_columns(V_, capacity_);
            fresh_ = MultiVector(V_); // a deep copy: every column is still to be multiplied (:123)
            AV_ = MultiVector(V_, capacity_);
            AV_.resize(0);
            a_ = DenseMatrix(capacity_, capacity_);
            b_ = DenseMatrix(capacity_, capacity_);
            if (s_.mass_) {
                MV_ = MultiVector(V_, capacity_);
                MV_.resize(0);
                m_ = DenseMatrix(capacity_, capacity_);
            }
            const double nb = s_.rhs_.norm2();
            scale_ = nb * nb;
        }

        // capacity change that keeps the columns in use (resize up, then back: the contract's resize preserves data, src/StlWrapper.cpp:225-263)
        static void reserve_columns(MultiVector &X, int columns)
        {
            const int in_use = X.N();
            X.resize(columns);
            X.resize(in_use);
        }
        static void reserve_order(DenseMatrix &S, int order)
        {
            const int in_use = S.M();
            S.resize(order, order);
            S.resize(in_use, in_use);
        }

        // block (r0.., c0..) of S <- G
        static void put(DenseMatrix &S, int r0, int c0, DenseMatrix const &G)
        {
            for (int j = 0; j < G.N(); ++j)
                for (int i = 0; i < G.M(); ++i) S(r0 + i, c0 + j) = G(i, j);
        }

        // A, B' (and M) applied to the fresh columns W; with k = columns multiplied before, the reduced matrices grow from k to k + w:
        // new rows W'[AV] and new columns V'[A W] of V'AV (:171-192), the symmetric border of V'BB'V (:174-200), likewise V'MV
        // (matlab/RAILSsolver.m:375-381).
        void border()
        {
            const int k = AV_.N(), w = fresh_.N();
            MultiVector AW, BtW, MW;
            {
                ScopedTimer t(&s_.sections_, "Apply A");
                AW = Ops::apply_append(s_.op_A_, fresh_, AV_);
            }
            {
                ScopedTimer t(&s_.sections_, "Apply B");
                BtW = s_.rhs_.transposed_times(fresh_);
            }
            if (s_.mass_) MW = Ops::apply_append(s_.op_M_, fresh_, MV_);
            if (trip_ == 0) { // B'V has the shape of B', not of V: made from the first B'W (:154-158)
                BtV_ = MultiVector(BtW, capacity_);
                BtV_.resize(0);
            }
            ScopedTimer t(&s_.sections_, "Compute VAV");
            a_.resize(k + w, k + w);
            b_.resize(k + w, k + w);
            if (s_.mass_) m_.resize(k + w, k + w);
            if (k > 0) {
                put(a_, k, 0, fresh_.dot(AV_.view(0, k - 1)));
                const DenseMatrix cross = BtW.dot(BtV_);
                put(b_, k, 0, cross);
                for (int j = 0; j < cross.N(); ++j)
                    for (int i = 0; i < cross.M(); ++i) b_(j, k + i) = cross(i, j);
                if (s_.mass_ && !s_.ortho_m_) put(m_, k, 0, fresh_.dot(MV_.view(0, k - 1)));
            }
            put(a_, 0, k, V_.dot(AW));
            if (s_.mass_ && !s_.ortho_m_) put(m_, 0, k, V_.dot(MW));
            put(b_, k, k, BtW.dot(BtW));
            BtV_.push_back(BtW);
        }

        void reduce()
        {
            ScopedTimer t(&s_.sections_, "dense_solve");
            if (const char *dump = getenv("RAILS_DEBUG_DUMP_PROJECTED")) { // diagnostics: the projected matrices of this trip, as text
                if (FILE *f = fopen(dump, "a")) {
                    const int k = a_.M();
                    fprintf(f, "%d\n", k);
                    for (int j = 0; j < k; ++j)
                        for (int i = 0; i < k; ++i) fprintf(f, "%.17g %.17g\n", (double)a_(i, j), (double)b_(i, j));
                    fclose(f);
                }
            }
            if (s_.mass_ && !s_.ortho_m_)
                s_.generalized_dense_solve(a_, b_, m_, T_);
            else
                s_.dense_solve(a_, b_, T_); // :209; with V'MV = I the generalized equation projects to the standard one (RAILSsolver.m:384)
        }

        // largest modulus among the Ritz values of the residual operator (:211-221)
        double estimate_residual(typename Ops::Lanczos &ritz)
        {
            {
                ScopedTimer t(&s_.sections_, "Residual Lanczos");
                Ops::lanczos(s_, backend_state_, AV_, s_.mass_ ? MV_ : V_, T_, a_, BtV_, opt_.lanczos_steps, ritz);
            }
            const double estimate = ritz.eigenvalues.norm_inf();
            s_.estimates_.push_back(estimate);
            s_.trips_++;
            if (s_.verbose_)
                std::cout << "trip " << trip_ + 1 << ": space " << V_.N() << ", residual estimate " << estimate << " (" << std::abs(estimate) / scale_
                          << " of ||B||^2)" << std::endl;
            return estimate;
        }

        int finish(bool converged, double estimate)
        {
            if (s_.verbose_)
                std::cout << "rails::Solver: " << (converged ? "converged" : "stopped without converging") << " after " << trip_ + 1
                          << " trips, residual estimate " << estimate / scale_ << " of ||B||^2, " << V_.N() << " basis vectors" << std::endl;
            s_.notify();
            return converged ? 0 : -1;
        }

        // by size, by trips since the last shrink, or on (first) convergence (:245-247)
        bool shrink_due(bool converged) const
        {
            return (opt_.shrink_at > 0 && V_.N() >= opt_.shrink_at) || (opt_.shrink_every > 0 && trip_ - last_shrink_ >= opt_.shrink_every) || converged;
        }

        // V <- V X with X the dominant eigenvectors of T; everything expressed in V follows (:263-298, matlab/RAILSsolver.m:499-504).
        // The next trip multiplies nothing (no fresh columns, :284) and V is not re-orthogonalised (X has orthonormal columns, :270).
        void shrink(bool converged)
        {
            if (s_.verbose_)
                std::cout << "rails::Solver: shrinking the space ("
                          << (converged ? "converged" : (opt_.shrink_at > 0 && V_.N() >= opt_.shrink_at ? "size limit" : "trip count")) << "), asking for "
                          << (opt_.shrink_to > 0 ? opt_.shrink_to : V_.N()) << " of " << V_.N() << " vectors" << std::endl;
            ScopedTimer t(&s_.sections_, "Restart");
            DenseMatrix X;
            s_.compute_restart_vectors(X, T_, std::min(opt_.shrink_to, V_.N()), opt_.keep_above);
            const int kept = X.N();
            auto congruence = [&](DenseMatrix &S) { // S <- X' S X
                DenseMatrix reducedS = X.transpose() * (S * X);
                S.resize(kept, kept);
                S.view() = reducedS;
            };
            Ops::multiply_inplace(V_, X);
            fresh_.resize(0);
            congruence(a_);
            Ops::multiply_inplace(AV_, X);
            Ops::on_restart(backend_state_, X, V_, AV_);
            congruence(b_);
            BtV_.view(0, kept - 1) = BtV_ * X;
            BtV_.resize(kept);
            if (s_.mass_) {
                congruence(m_);
                Ops::multiply_inplace(MV_, X);
            }
            if (s_.verbose_) std::cout << "rails::Solver: " << kept << " vectors kept" << std::endl;
            last_shrink_ = trip_;
        }

        // The Ritz vectors of the `add` largest |Ritz values| join V: add = min(expand_by, Ritz values available, room up to shrink_at
        // or n) (:306-307); capacity grows by 100 columns at a time (:311-332); only the new columns are orthogonalised (:340) and
        // they are the fresh ones of the next trip (:342).
        void expand(typename Ops::Lanczos const &ritz, int width_before)
        {
            const int room = (opt_.shrink_at > 0 ? opt_.shrink_at : n_) - V_.N();
            const int add = std::min(std::min(opt_.expand_by, ritz.eigenvalues.M()), room);
            if (V_.N() + add > capacity_) {
                capacity_ += 100;
                reserve_columns(V_, capacity_);
                reserve_columns(AV_, capacity_);
                reserve_order(a_, capacity_);
                reserve_columns(BtV_, capacity_);
                reserve_order(b_, capacity_);
                if (s_.mass_) {
                    reserve_columns(MV_, capacity_);
                    reserve_order(m_, capacity_);
                }
            }
            std::vector<int> leading;
            find_largest_eigenvalues(ritz.eigenvalues, leading, add);
            {
                ScopedTimer t(&s_.sections_, "Expand");
                ritz.append_to(V_, leading, add);
            }
            int kept = add;
            {
                ScopedTimer t(&s_.sections_, "Orthogonalize");
                if (s_.mass_orthogonalisation())
                    kept = m_orthogonalize(width_before);
                else
                    V_.orthogonalize();
            }
            fresh_ = V_.view(width_before, width_before + kept - 1);
        }

        // Modified Gram-Schmidt in the M inner product on the columns from `first` on (matlab/RAILSsolver.m:583-597, Morth with M): every
        // new column is normalised, M-projected against all columns before it (twice: the reference's single sweep leaves a component
        // of relative size eps * cond, and V'MV = I is what makes the projected equation the standard one), M-normalised, and DROPPED
        // when less than 1e-8 of it is left (:592-594) -- V then grows by fewer columns than asked for.  Returns the columns kept.
        int m_orthogonalize(int first)
        {
            int n = V_.N(), j = first;
            while (j < n) {
                MultiVector v = V_.view(j);
                v /= v.norm();
                if (j > 0) {
                    MultiVector const before = V_.view(0, j - 1);
                    for (int pass = 0; pass < 2; ++pass) {
                        MultiVector const Mv = s_.op_M_ * v;
                        v -= before * before.dot(Mv);
                    }
                }
                MultiVector const Mv = s_.op_M_ * v;
                DenseMatrix const vMv = v.dot(Mv);
                const double energy = std::abs(vMv(0, 0)); // (a definite M of either sign: the MOC data's mass entries are negative)
                const double nrm = std::sqrt(energy);
                if (!(nrm >= 1e-8)) { // nothing new in this column: the later ones move up
                    for (int l = j + 1; l < n; ++l) V_.view(l - 1) = V_.view(l);
                    V_.resize(--n);
                    continue;
                }
                v /= nrm;
                ++j;
            }
            return n - first;
        }
    };
    friend class Run;

    void notify()
    {
        if (on_trip_) on_trip_(trips_);
    }

protected:
    Matrix op_A_;
    BType rhs_;
    Matrix op_M_;
    Settings settings_;

    bool mass_ = false;
    bool ortho_m_ = false;
    bool verbose_ = true;
    int trip_budget_ = 0;
    int trips_ = 0;
    std::vector<double> estimates_;
    std::function<void(int)> on_trip_;
    std::function<bool()> broken_;
    std::map<std::string, double> sections_;
    bool projected_lanczos_ = false;
};

} // namespace rails

#endif
