// HostDenseMatrix -- the DenseMatrix template parameter of Solver<Matrix, MultiVector, DenseMatrix>
// for the HIP back end.  The small projected matrices (VAV, VBV, T, H, X) stay on the host (north
// star): column-major storage with capacity-preserving resize and `operator double*`, i.e. the
// DenseMatrix part of the reference's StlWrapper contract (src/StlWrapper.hpp:32-90; the uses in
// src/LyapunovSolver.hpp:126,165,180,286-288,357,441,458 and src/StlTools.hpp:22).
//
// Value semantics follow the reference: copy construction is a deep copy (src/StlWrapper.cpp:31-44),
// assignment to a non-view SHARES storage, assignment to a view copies INTO the viewed storage
// (:65-121), copy() is a deep copy.
#ifndef RAILS_HOSTDENSEMATRIX_HPP
#define RAILS_HOSTDENSEMATRIX_HPP

#include <algorithm>
#include <cmath>
#include <cstring>
#include <iostream>
#include <memory>
#include <vector>

#include "rails_hip.h"

namespace rails
{

class HostDenseMatrix
{
    std::shared_ptr<std::vector<double>> buf_;
    int m_, n_;         // actual size
    int m_max_, n_max_; // capacity
    bool is_view_;
    bool transpose_;

    double *data() const { return buf_ ? buf_->data() : nullptr; }

public:
    HostDenseMatrix() : m_(-1), n_(-1), m_max_(-1), n_max_(-1), is_view_(false), transpose_(false) {}

    HostDenseMatrix(int m, int n) : m_(m), n_(n), m_max_(m), n_max_(n), is_view_(false), transpose_(false)
    {
        buf_ = std::make_shared<std::vector<double>>((size_t)std::max(m, 0) * std::max(n, 0), 0.0);
    }

    HostDenseMatrix(HostDenseMatrix const &o)
        : m_(o.m_), n_(o.n_), m_max_(o.m_max_), n_max_(o.n_max_), is_view_(false), transpose_(o.transpose_)
    {
        if (o.buf_) buf_ = std::make_shared<std::vector<double>>(*o.buf_);
    }

    HostDenseMatrix(HostDenseMatrix &&o) = default;

    virtual ~HostDenseMatrix() {}

    HostDenseMatrix &operator=(HostDenseMatrix const &o)
    {
        if (!is_view_) {
            buf_ = o.buf_;
            m_ = o.m_;
            n_ = o.n_;
            m_max_ = o.m_max_;
            n_max_ = o.n_max_;
            transpose_ = o.transpose_;
            return *this;
        }
        // view target: copy into the viewed storage
        int cols = std::min(n_, o.n_), rows = std::min(m_, o.m_);
        for (int j = 0; j < cols; ++j) memcpy(data() + (size_t)j * m_max_, o.data() + (size_t)j * o.m_max_, sizeof(double) * rows);
        return *this;
    }

    HostDenseMatrix &operator=(double v)
    {
        for (int j = 0; j < n_; ++j) std::fill_n(data() + (size_t)j * m_max_, m_, v);
        return *this;
    }

    HostDenseMatrix &operator*=(double s)
    {
        for (int j = 0; j < n_; ++j)
            for (int i = 0; i < m_; ++i) data()[i + (size_t)j * m_max_] *= s;
        return *this;
    }

    operator double *() const { return data(); }

    double &operator()(int i, int j = 0) { return data()[i + (size_t)j * m_max_]; }
    double const &operator()(int i, int j = 0) const { return data()[i + (size_t)j * m_max_]; }

    int M() const { return transpose_ ? n_ : m_; }
    int N() const { return transpose_ ? m_ : n_; }
    int LDA() const { return transpose_ ? n_max_ : m_max_; }
    bool transposed() const { return transpose_; } // storage is that of the un-transposed matrix, leading dimension raw_ld()
    int raw_ld() const { return m_max_; }

    // resize keeps contents and, within capacity, the leading dimension (src/StlWrapper.cpp:225-263)
    void resize(int m, int n)
    {
        if (buf_ && m <= m_max_ && n <= n_max_) {
            m_ = m;
            n_ = n;
            return;
        }
        HostDenseMatrix out(m, n);
        if (buf_ && m_max_ > 0) {
            int cols = std::min(n_, n), rows = std::min(m_, m);
            for (int j = 0; j < cols; ++j) memcpy(out.data() + (size_t)j * out.m_max_, data() + (size_t)j * m_max_, sizeof(double) * rows);
        }
        bool was_view = is_view_;
        is_view_ = false;
        *this = out;
        is_view_ = was_view;
    }

    HostDenseMatrix view()
    {
        HostDenseMatrix out;
        out.buf_ = buf_;
        out.m_ = m_;
        out.n_ = n_;
        out.m_max_ = m_max_;
        out.n_max_ = n_max_;
        out.transpose_ = transpose_;
        out.is_view_ = true;
        return out;
    }

    HostDenseMatrix copy() const { return HostDenseMatrix(*this); }

    HostDenseMatrix transpose() const
    {
        HostDenseMatrix out;
        out.buf_ = buf_;
        out.m_ = m_;
        out.n_ = n_;
        out.m_max_ = m_max_;
        out.n_max_ = n_max_;
        out.transpose_ = !transpose_;
        return out;
    }

    // op(this) * op(other), small and on the host
    HostDenseMatrix operator*(HostDenseMatrix const &o) const
    {
        HostDenseMatrix out(M(), o.N());
        if (o.M() != N()) {
            std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << o.M() << "x" << o.N() << std::endl;
            return out;
        }
        if (M() > 0 && o.N() > 0 && N() > 0)
            rails_dgemm(transpose_ ? 'T' : 'N', o.transpose_ ? 'T' : 'N', M(), o.N(), N(), 1.0, data(), m_max_, o.data(), o.m_max_, 0.0, out.data(),
                        out.m_max_);
        return out;
    }

    double norm_inf() const // max absolute row sum (src/StlWrapper.cpp:291-303)
    {
        double out = 0.0;
        for (int i = 0; i < m_; ++i) {
            double row_sum = 0.0;
            for (int j = 0; j < n_; ++j) row_sum += std::abs((*this)(i, j));
            out = std::max(out, row_sum);
        }
        return out;
    }

    // all eigenpairs of a symmetric matrix, ascending (src/StlWrapper.cpp:433-479, num/tol selection included)
    int eigs(HostDenseMatrix &v, HostDenseMatrix &d, int num = -1, double tol = 1e-16) const
    {
        v = copy();
        int m = v.M();
        if (num < 1) num = m;
        d.resize(m, 1);
        int info = 0;
        rails_dsyev('V', 'U', m, v.data(), v.LDA(), d.data(), &info);
        if (num != m || tol > 1e-14) {
            std::vector<std::pair<int, double>> iv;
            for (int i = 0; i < m; ++i) iv.push_back(std::make_pair(i, d(i, 0)));
            std::sort(iv.begin(), iv.end(),
                      [](std::pair<int, double> const &a, std::pair<int, double> const &b) { return std::abs(a.second) > std::abs(b.second); });
            HostDenseMatrix tmpv(m, num), tmpd(num, 1);
            int idx = 0;
            for (int i = 0; i < num; ++i)
                if (std::abs(iv[i].second) > tol) {
                    for (int r = 0; r < m; ++r) tmpv(r, idx) = v(r, iv[i].first);
                    tmpd(idx, 0) = iv[i].second;
                    idx++;
                }
            tmpv.resize(m, idx);
            tmpd.resize(idx, 1);
            v = tmpv;
            d = tmpd;
        }
        if (info) std::cerr << "Eigenvalues info = " << info << std::endl;
        return info;
    }
};

} // namespace rails

#endif
