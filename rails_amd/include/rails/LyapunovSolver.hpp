// rails::Solver<Matrix, MultiVector, DenseMatrix> -- RAILS outer loop with the template surface of the
// reference's RAILS::Solver (src/LyapunovSolverDecl.hpp:9-51): ctor (A, B, M), set_parameters,
// solve(V, T), dense_solve, resid_lanczos, compute_restart_vectors, same parameter names and defaults
// (src/LyapunovSolver.hpp:27-36,76-87), same return codes (0 converged / -1 not converged / 1 loop
// exhausted, :239-240,345; set_parameters 0 / 1, :89-97).
//
// It is written against the duck-typed backend contract only, so any conforming backend works.  For
// the HIP backend three customisation points (struct SolverOps below, specialised in
// rails/HipSolverOps.hpp) replace member-by-member sequences by fused device work:
//   * apply_append    : A*W written straight into AV's tail (no temporary, no push_back copy);
//   * lanczos         : fused one-pass-per-step residual Lanczos, alpha/beta on the device;
//   * multiply_inplace: V <- V*X at restart without the temporary of `V.view(..) = V * X`.
// Unlike the reference's C++ (which stores M and never reads it, src/LyapunovSolver.hpp:26), a
// non-identity mass matrix can be switched on with use_mass_matrix(true): the generalized iteration
// then follows the MATLAB implementation (matlab/RAILSsolver.m:368-395,499-504).
#ifndef RAILS_LYAPUNOVSOLVER_HPP
#define RAILS_LYAPUNOVSOLVER_HPP

#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <map>
#include <iostream>
#include <locale>
#include <string>
#include <utility>
#include <vector>

#include "rails_hip.h"

namespace rails
{

// ---- B as operator or multivector (src/MatrixOrMultiVectorWrapper.hpp:7-98) --------------------
template <class Matrix, class MultiVector>
class MatrixOrMultiVectorWrapper
{
    bool is_matrix_;
    Matrix matrix_;
    MultiVector vector_;
    bool transpose_;

public:
    MatrixOrMultiVectorWrapper() = delete;
    MatrixOrMultiVectorWrapper(Matrix const &other) : is_matrix_(true), matrix_(other), transpose_(false) {}
    MatrixOrMultiVectorWrapper(MultiVector const &other) : is_matrix_(false), vector_(other), transpose_(false) {}
    virtual ~MatrixOrMultiVectorWrapper() {}

    bool is_matrix() const { return is_matrix_; }
    MultiVector const &vector() const { return vector_; }

    double norm() const { return is_matrix_ ? matrix_.norm() : vector_.norm(); }

    MatrixOrMultiVectorWrapper transpose() const
    {
        MatrixOrMultiVectorWrapper tmp(*this);
        tmp.transpose_ = !tmp.transpose_;
        return tmp;
    }

    MultiVector operator*(MultiVector const &other) const
    {
        if (transpose_) return is_matrix_ ? matrix_.transpose() * other : vector_.transpose() * other;
        return is_matrix_ ? matrix_ * other : vector_ * other;
    }
};

template <class Type>
class MatrixOrMultiVectorWrapper<Type, Type>
{
    Type type_;
    bool transpose_;

public:
    MatrixOrMultiVectorWrapper() = delete;
    template <class MatrixOrMultiVector>
    MatrixOrMultiVectorWrapper(MatrixOrMultiVector const &other) : type_(other), transpose_(false)
    {
    }
    virtual ~MatrixOrMultiVectorWrapper() {}
    bool is_matrix() const { return false; }
    Type const &vector() const { return type_; }
    double norm() const { return type_.norm(); }
    MatrixOrMultiVectorWrapper transpose() const
    {
        MatrixOrMultiVectorWrapper tmp(*this);
        tmp.transpose_ = !tmp.transpose_;
        return tmp;
    }
    Type operator*(Type const &other) const { return transpose_ ? type_.transpose() * other : type_ * other; }
};

// ---- eigenvalue selection (src/StlTools.hpp:12-30): indices of the N largest |values| ----------
template <class DenseMatrix>
int find_largest_eigenvalues(DenseMatrix const &eigenvalues, std::vector<int> &indices, int N)
{
    std::vector<std::pair<int, double>> index_to_value;
    for (int i = 0; i < eigenvalues.M(); i++) index_to_value.push_back(std::pair<int, double>(i, eigenvalues(i)));
    std::sort(index_to_value.begin(), index_to_value.end(),
              [](std::pair<int, double> const &a, std::pair<int, double> const &b) { return std::abs(a.second) > std::abs(b.second); });
    for (int i = 0; i < N; i++) indices.push_back(index_to_value[i].first);
    return 0;
}

// ---- parameter lookup with the reference's case variants (src/LyapunovSolver.hpp:40-70) ---------
template <class ParameterList, class Type>
Type get_parameter(ParameterList &params, std::string const &name, Type def)
{
    Type ret = params.get(name, def);
    std::locale loc;
    std::string str = name;
    for (std::string::iterator it = str.begin(); it != str.end(); ++it) *it = std::toupper(*it, loc);
    ret = params.get(str, ret);
    str = name;
    for (std::string::iterator it = str.begin(); it != str.end(); ++it) *it = std::tolower(*it, loc);
    ret = params.get(str, ret);
    str = name;
    if (str.length() > 0) str[0] = std::toupper(str[0]);
    for (std::string::iterator it = str.begin() + 1; it < str.end(); ++it)
        if (!isalpha(*(it - 1)) && islower(*it)) *it = std::toupper(*it, loc);
    ret = params.get(str, ret);
    return ret;
}

// Wall-clock accumulators under the reference's profile section names (src/Timer.hpp:101-106 RAILS_START_TIMER /
// RAILS_END_TIMER; "Apply A", "Apply B", "Compute VAV", "dense_solve", "Residual Lanczos", ...).  Host time between
// synchronisation points; the device side is profiled with rocprofv3.
class ScopedTimer
{
    std::map<std::string, double> *acc_;
    std::string name_;
    std::chrono::steady_clock::time_point t0_;

public:
    ScopedTimer(std::map<std::string, double> *acc, const char *name) : acc_(acc), name_(name), t0_(std::chrono::steady_clock::now()) {}
    ~ScopedTimer() { (*acc_)[name_] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
};

template <class Matrix, class MultiVector, class DenseMatrix>
class Solver;

// ---- customisation points; the generic versions restate the reference member by member ---------
template <class Matrix, class MultiVector, class DenseMatrix>
struct SolverOps {
    typedef Solver<Matrix, MultiVector, DenseMatrix> SolverT;

    // result of one residual-Lanczos run: Ritz values and a way to append selected Ritz vectors to V
    struct Lanczos {
        DenseMatrix eigenvalues;
        MultiVector eigenvectors;
        void append_to(MultiVector &V, std::vector<int> const &indices, int count) const
        {
            for (int i = 0; i < count; i++) V.push_back(eigenvectors.view(indices[i])); // src/LyapunovSolver.hpp:338-339
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

    // VAV and BV (the projected operator and B'V) are passed along for backends that can use them; the generic form
    // is the reference's member sequence and ignores them
    static int lanczos(SolverT &solver, State &, MultiVector const &AV, MultiVector const &MV, DenseMatrix const &T, DenseMatrix const &,
                       MultiVector const &, int max_iter, Lanczos &out)
    {
        DenseMatrix H(max_iter + 1, max_iter + 1);
        out.eigenvalues = DenseMatrix(max_iter, 1);
        return solver.resid_lanczos(AV, MV, T, H, out.eigenvectors, out.eigenvalues, max_iter);
    }

    // after V <- V X and AV <- AV X of a restart (:265-291)
    static void on_restart(State &, DenseMatrix const &, MultiVector &, MultiVector &) {}

    // V <- V * X (first X.N() columns), as `V.view(0, X.N()-1) = V * X; V.resize(X.N())`   (:265-266)
    static void multiply_inplace(MultiVector &V, DenseMatrix const &X)
    {
        V.view(0, X.N() - 1) = V * X;
        V.resize(X.N());
    }
};

template <class Matrix, class MultiVector, class DenseMatrix>
class Solver
{
public:
    typedef SolverOps<Matrix, MultiVector, DenseMatrix> Ops;

    template <class MatrixOrMultiVector>
    Solver(Matrix const &A, MatrixOrMultiVector const &B, Matrix const &M)
        : A_(A), B_(B), M_(M), max_iter_(1000), tol_(1e-3), expand_size_(3), lanczos_iterations_(10), restart_size_(-1), reduced_size_(-1),
          restart_iterations_(20), restart_tolerance_(tol_ * 1e-3), minimize_solution_space_(true), restart_from_solution_(false),
          use_mass_matrix_(false), verbose_(true), max_trips_(0), trips_(0), projected_lanczos_(false)
    {
    }

    virtual ~Solver() {}

    template <class ParameterList>
    int set_parameters(ParameterList &params)
    {
        max_iter_ = get_parameter(params, "Maximum iterations", max_iter_);
        tol_ = get_parameter(params, "Tolerance", tol_);
        expand_size_ = get_parameter(params, "Expand size", expand_size_);
        lanczos_iterations_ = get_parameter(params, "Lanczos iterations", lanczos_iterations_);
        restart_size_ = get_parameter(params, "Restart size", restart_size_);
        reduced_size_ = get_parameter(params, "Reduced size", reduced_size_);
        restart_iterations_ = get_parameter(params, "Restart iterations", restart_iterations_);
        restart_tolerance_ = get_parameter(params, "Restart tolerance", tol_ * 1e-3);
        minimize_solution_space_ = get_parameter(params, "Minimize solution space", minimize_solution_space_);
        restart_from_solution_ = get_parameter(params, "Restart from solution", restart_from_solution_);
        if (lanczos_iterations_ <= expand_size_) {
            std::cerr << "Amount of Lanczos iterations is smaller than "
                      << "the amount of vectors that are used to expand "
                      << "the space in every iteration" << std::endl;
            return 1;
        }
        return 0;
    }

    // extensions (not in the reference): generalized M, quiet mode, bounded runs, instrumentation
    void use_mass_matrix(bool on) { use_mass_matrix_ = on; }
    void set_verbose(bool on) { verbose_ = on; }
    void set_max_trips(int n) { max_trips_ = n; }
    // extension: residual Lanczos carried in the (2k+p+1)-dimensional coefficient space where the backend supports it
    void set_projected_lanczos(bool on) { projected_lanczos_ = on; }
    bool projected_lanczos() const { return projected_lanczos_; }
    bool mass_matrix_in_use() const { return use_mass_matrix_; }
    void set_trip_callback(std::function<void(int)> cb) { on_trip_ = cb; }
    int trips() const { return trips_; }
    std::vector<double> const &residual_history() const { return res_hist_; }
    std::map<std::string, double> const &profile() const { return profile_; }
    void reset_profile() { profile_.clear(); }
    MatrixOrMultiVectorWrapper<Matrix, MultiVector> const &B() const { return B_; }

    // Solve A*V*T*V' + V*T*V'*A' + B*B' = 0                                (src/LyapunovSolver.hpp:100-346)
    int solve(MultiVector &V, DenseMatrix &T)
    {
        int n = V.M();
        int max_size = std::max(V.N(), std::min(restart_size_ > 0 ? restart_size_ : 100, n));
        trips_ = 0;
        res_hist_.clear();
        ops_state_ = typename Ops::State();

        if (!restart_from_solution_) {
            V.resize(max_size);
            V.resize(1);
            V.random();
            V.orthogonalize();
        } else if (max_size != V.N()) {
            int previous_size = V.N();
            V.resize(max_size);
            V.resize(previous_size);
        }

        MultiVector W = V; // deep copy (:123)

        MultiVector AV(V, max_size);
        DenseMatrix VAV(max_size, max_size);
        AV.resize(0);

        MultiVector MV; // generalized form only
        DenseMatrix VMV;
        if (use_mass_matrix_) {
            MV = MultiVector(V, max_size);
            MV.resize(0);
            VMV = DenseMatrix(max_size, max_size);
        }

        MultiVector BV;
        DenseMatrix VBV(max_size, max_size);

        bool converged_previously = false;
        int previous_restart = 0;
        double r0 = B_.norm();

        for (int iter = 0; iter < max_iter_; iter++) {
            if (on_trip_) on_trip_(trips_);
            int N_V = V.N();
            if (W.N()) {
                int N_AV = AV.N();
                int wn = W.N();
                MultiVector AW, BW;
                {
                    ScopedTimer t(&profile_, "Apply A");
                    AW = Ops::apply_append(A_, W, AV); // :146 and :203
                }
                {
                    ScopedTimer t(&profile_, "Apply B");
                    BW = B_.transpose() * W; // :150
                }
                MultiVector MW;
                if (use_mass_matrix_) MW = Ops::apply_append(M_, W, MV); // RAILSsolver.m:368-373

                if (!iter) { // BV looks like B', not V (:154-158)
                    BV = MultiVector(BW, max_size);
                    BV.resize(0);
                }

                ScopedTimer tvav(&profile_, "Compute VAV");
                int s = N_AV + wn;
                VAV.resize(s, s); // keeps what was there (:165-166)
                VBV.resize(s, s);
                if (use_mass_matrix_) VMV.resize(s, s);

                if (N_AV > 0) { // :171-184
                    DenseMatrix WAV = W.dot(AV.view(0, N_AV - 1));
                    DenseMatrix WBV = BW.dot(BV);
                    for (int i = 0; i < WAV.M(); i++)
                        for (int j = 0; j < WAV.N(); j++) {
                            VAV(i + N_AV, j) = WAV(i, j);
                            VBV(i + N_AV, j) = WBV(i, j);
                            VBV(j, i + N_AV) = WBV(i, j);
                        }
                    if (use_mass_matrix_) {
                        DenseMatrix WMV = W.dot(MV.view(0, N_AV - 1));
                        for (int i = 0; i < WMV.M(); i++)
                            for (int j = 0; j < WMV.N(); j++) VMV(i + N_AV, j) = WMV(i, j);
                    }
                }

                DenseMatrix VAW = V.dot(AW); // :187-192
                for (int i = 0; i < VAW.M(); i++)
                    for (int j = 0; j < VAW.N(); j++) VAV(i, j + N_AV) = VAW(i, j);
                if (use_mass_matrix_) {
                    DenseMatrix VMW = V.dot(MW);
                    for (int i = 0; i < VMW.M(); i++)
                        for (int j = 0; j < VMW.N(); j++) VMV(i, j + N_AV) = VMW(i, j);
                }

                DenseMatrix WBW = BW.dot(BW); // :195-200
                for (int i = 0; i < WBW.M(); i++)
                    for (int j = 0; j < WBW.N(); j++) VBV(i + N_AV, j + N_AV) = WBW(i, j);

                BV.push_back(BW); // :204 (AV was extended by apply_append)
            }

            {
                ScopedTimer t(&profile_, "dense_solve");
                if (use_mass_matrix_)
                    generalized_dense_solve(VAV, VBV, VMV, T);
                else
                    dense_solve(VAV, VBV, T); // :209
            }

            typename Ops::Lanczos lz; // :211-215
            {
                ScopedTimer t(&profile_, "Residual Lanczos");
                Ops::lanczos(*this, ops_state_, AV, use_mass_matrix_ ? MV : V, T, VAV, BV, lanczos_iterations_, lz);
            }

            double res = lz.eigenvalues.norm_inf(); // :217
            res_hist_.push_back(res);
            trips_++;

            if (verbose_)
                std::cout << "Iteration " << iter + 1 << ". Estimate Lanczos, absolute: " << res << ", relative: " << std::abs(res) / r0 / r0
                          << std::endl;

            bool converged = std::abs(res) < tol_ * r0 * r0; // :223
            if (converged || iter + 1 >= max_iter_ || V.N() >= n) {
                if (converged && minimize_solution_space_ && !converged_previously)
                    converged_previously = true;
                else {
                    if (verbose_)
                        std::cout << "The Lyapunov solver " << (converged ? "converged" : "did not converge") << " in " << iter + 1
                                  << " iterations with a final relative residual of " << res / r0 / r0 << ". The size of the space used "
                                  << "for the solution is " << V.N() << std::endl;
                    if (on_trip_) on_trip_(trips_);
                    return converged ? 0 : -1;
                }
            }
            if (max_trips_ > 0 && trips_ >= max_trips_) { // extension: bounded run
                if (on_trip_) on_trip_(trips_);
                return 2;
            }

            // restart with reduced_size_ vectors (:245-304)
            if ((restart_size_ > 0 && V.N() >= restart_size_) || (restart_iterations_ > 0 && iter - previous_restart >= restart_iterations_) ||
                converged) {
                if (verbose_) {
                    if (converged)
                        std::cout << "Method converged. Minimizing the solution space size";
                    else if (restart_size_ > 0)
                        std::cout << "Reached the maximum space size of " << restart_size_;
                    else if (restart_iterations_ > 0)
                        std::cout << restart_iterations_ << " iterations have passed";
                    else
                        std::cout << "No clue what happened";
                    std::cout << ". Trying to restart with " << (reduced_size_ > 0 ? reduced_size_ : V.N()) << " vectors" << std::endl;
                }

                ScopedTimer trs(&profile_, "Restart");
                DenseMatrix X;
                compute_restart_vectors(X, T, std::min(reduced_size_, V.N()), restart_tolerance_);

                Ops::multiply_inplace(V, X); // :265-266
                if (verbose_) std::cout << "Restarted with " << V.N() << " vectors" << std::endl;

                W.resize(0); // :284

                DenseMatrix tmp = X.transpose() * (VAV * X); // :286-288
                VAV.resize(X.N(), X.N());
                VAV.view() = tmp;

                Ops::multiply_inplace(AV, X); // :290-291
                Ops::on_restart(ops_state_, X, V, AV);

                tmp = X.transpose() * (VBV * X); // :293-295
                VBV.resize(X.N(), X.N());
                VBV.view() = tmp;

                BV.view(0, X.N() - 1) = BV * X; // :297-298
                BV.resize(X.N());

                if (use_mass_matrix_) { // RAILSsolver.m:499-504
                    tmp = X.transpose() * (VMV * X);
                    VMV.resize(X.N(), X.N());
                    VMV.view() = tmp;
                    Ops::multiply_inplace(MV, X);
                }

                previous_restart = iter;
                continue;
            }

            int expand_vectors =
                std::min(std::min(expand_size_, lz.eigenvalues.M()), (restart_size_ > 0 ? restart_size_ : n) - V.N()); // :306-307

            if (V.N() + expand_vectors > max_size) { // grow by 100 columns at a time (:311-332)
                max_size += 100;
                int previous_size = V.N();
                V.resize(max_size);
                V.resize(previous_size);
                AV.resize(max_size);
                AV.resize(previous_size);
                VAV.resize(max_size, max_size);
                VAV.resize(previous_size, previous_size);
                BV.resize(max_size);
                BV.resize(previous_size);
                VBV.resize(max_size, max_size);
                VBV.resize(previous_size, previous_size);
                if (use_mass_matrix_) {
                    MV.resize(max_size);
                    MV.resize(previous_size);
                    VMV.resize(max_size, max_size);
                    VMV.resize(previous_size, previous_size);
                }
            }

            std::vector<int> indices; // :335-340
            find_largest_eigenvalues(lz.eigenvalues, indices, expand_vectors);
            {
                ScopedTimer t(&profile_, "Expand");
                lz.append_to(V, indices, expand_vectors);
            }
            {
                ScopedTimer t(&profile_, "Orthogonalize");
                V.orthogonalize();
            }

            W = V.view(N_V, N_V + expand_vectors - 1); // :342
        }
        if (on_trip_) on_trip_(trips_);
        return 1;
    }

    // Solve A*X + X*A' + B = 0                                             (src/LyapunovSolver.hpp:348-365)
    int dense_solve(DenseMatrix const &A, DenseMatrix const &B, DenseMatrix &X)
    {
        X = B.copy();
        DenseMatrix A_copy = A.copy();
        double scale = 1.0;
        int info = 0;
        int n = A.M();
        rails_sb03md('C', 'X', 'N', 'T', n, A_copy, A_copy.LDA(), X, X.LDA(), &scale, &info);
        X *= -1.0;
        if (info != 0 && info != n + 1) std::cerr << "Error: sb03md returned info = " << info << std::endl;
        return info;
    }

    // T = lyap(VAV, VBV, [], VMV):  VAV T VMV' + VMV T VAV' + VBV = 0      (matlab/RAILSsolver.m:382,
    // matlab/mex/lyap.c:125-133 call SLICOT sg03ad).  Reduced to the standard equation with the Cholesky
    // factor VMV = L L':  (L^-1 VAV L^-T) Tt + Tt (..)' + L^-1 VBV L^-T = 0,  T = L^-T Tt L^-1.  M symmetric definite (either sign).
    int generalized_dense_solve(DenseMatrix const &A, DenseMatrix const &B, DenseMatrix const &Mm, DenseMatrix &X)
    {
        int k = A.M();
        DenseMatrix L(k, k), Ai(k, k), Bi(k, k);
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < k; ++i) {
                L(i, j) = 0.5 * (Mm(i, j) + Mm(j, i));
                Ai(i, j) = A(i, j);
                Bi(i, j) = B(i, j);
            }
        int info = 0;
        rails_dpotrf('L', k, L, L.LDA(), &info);
        if (info) {
            // The equation does not change under (A, M) -> (-A, -M): a negative definite mass matrix (the reference's MOC data set,
            // matlab/DataErik/Bp1.co, is one) is the definite case with both signs flipped.  SLICOT's sg03ad takes any nonsingular
            // M; an indefinite one is not supported here.
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) {
                    L(i, j) = -0.5 * (Mm(i, j) + Mm(j, i));
                    Ai(i, j) = -A(i, j);
                }
            rails_dpotrf('L', k, L, L.LDA(), &info);
        }
        if (info) {
            std::cerr << "Error: projected mass matrix is neither positive nor negative definite (dpotrf info = " << info << ")" << std::endl;
            return info;
        }
        // (L^-1 A L^-T, L^-1 B L^-T), solve, T = L^-T Tt L^-1: six triangular solves with k right-hand sides each
        rails_dtrsm('L', 'L', 'N', 'N', k, k, 1.0, L, L.LDA(), Ai, Ai.LDA());
        rails_dtrsm('R', 'L', 'T', 'N', k, k, 1.0, L, L.LDA(), Ai, Ai.LDA());
        rails_dtrsm('L', 'L', 'N', 'N', k, k, 1.0, L, L.LDA(), Bi, Bi.LDA());
        rails_dtrsm('R', 'L', 'T', 'N', k, k, 1.0, L, L.LDA(), Bi, Bi.LDA());
        DenseMatrix Tt;
        int ret = dense_solve(Ai, Bi, Tt);
        rails_dtrsm('L', 'L', 'T', 'N', k, k, 1.0, L, L.LDA(), Tt, Tt.LDA());
        rails_dtrsm('R', 'L', 'N', 'N', k, k, 1.0, L, L.LDA(), Tt, Tt.LDA());
        X = Tt;
        return ret;
    }

    // Eigenpairs of R = AV*T*V' + V*T*AV' + B*B' by Lanczos, member by member  (src/LyapunovSolver.hpp:367-447).
    // This generic form is what any conforming backend gets; the HIP backend's solve() uses the fused
    // kernel instead (SolverOps::lanczos) but this member stays available and equivalent.
    int resid_lanczos(MultiVector const &AV, MultiVector const &V, DenseMatrix const &T, DenseMatrix &H, MultiVector &eigenvectors,
                      DenseMatrix &eigenvalues, int max_iter)
    {
        MultiVector Q(V, max_iter + 1);
        Q.resize(1);
        Q.random();
        Q.view(0) /= Q.norm();

        H = 0.0;

        double alpha = 0.0;
        double beta = 0.0;

        int iter = 0;
        for (int i = 0; i < max_iter; i++) {
            Q.resize(iter + 2);

            MultiVector Y = B_.transpose() * Q.view(iter);
            Q.view(iter + 1) = B_ * Y;

            DenseMatrix Z = V.dot(Q.view(iter));
            Z = T * Z;
            Q.view(iter + 1) += AV * Z;

            Z = AV.dot(Q.view(iter));
            Z = T * Z;
            Q.view(iter + 1) += V * Z;

            alpha = Q.view(iter + 1).dot(Q.view(iter))(0, 0);
            H(iter, iter) = alpha;

            Q.view(iter + 1) -= alpha * Q.view(iter);
            if (iter > 0) Q.view(iter + 1) -= beta * Q.view(iter - 1);

            beta = Q.view(iter + 1).norm();
            if (beta < 1e-14) {
                iter++;
                break;
            }

            H(iter + 1, iter) = beta;
            H(iter, iter + 1) = beta;

            Q.view(iter + 1) /= beta;

            iter++;
        }

        H.resize(iter, iter);
        Q.resize(iter);

        DenseMatrix v(iter, iter);
        H.eigs(v, eigenvalues);

        eigenvectors = Q * v;
        return 0;
    }

    // Restart vectors from the eigen-decomposition of T                    (src/LyapunovSolver.hpp:449-482)
    int compute_restart_vectors(DenseMatrix &X, DenseMatrix const &T, int num, double tol)
    {
        int info;
        DenseMatrix eigenvectors = T.copy();
        DenseMatrix eigenvalues(T.N(), 1);
        rails_dsyev('V', 'U', eigenvectors.N(), eigenvectors, eigenvectors.LDA(), eigenvalues, &info);

        num = (num > 0 ? num : T.N());
        X = DenseMatrix(T.N(), num);

        std::vector<int> indices;
        find_largest_eigenvalues(eigenvalues, indices, num);

        int idx = 0;
        for (int i = 0; i < num; ++i) {
            if (std::abs(eigenvalues(indices[i], 0)) > tol) {
                for (int j = 0; j < T.N(); ++j) X(j, i) = eigenvectors(j, indices[i]);
                idx++;
            }
        }
        X.resize(T.N(), idx);
        return 0;
    }

    int lanczos_iterations() const { return lanczos_iterations_; }

protected:
    Matrix A_;
    MatrixOrMultiVectorWrapper<Matrix, MultiVector> B_;
    Matrix M_;

    int max_iter_;
    double tol_;
    int expand_size_;
    int lanczos_iterations_;
    int restart_size_;
    int reduced_size_;
    int restart_iterations_;
    double restart_tolerance_;
    bool minimize_solution_space_;
    bool restart_from_solution_;

    bool use_mass_matrix_;
    bool verbose_;
    int max_trips_;
    int trips_;
    std::vector<double> res_hist_;
    std::function<void(int)> on_trip_;
    std::map<std::string, double> profile_;
    bool projected_lanczos_;
    typename Ops::State ops_state_;
};

} // namespace rails

#endif
