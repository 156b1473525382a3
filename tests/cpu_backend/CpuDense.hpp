// CpuDense -- TEST SCAFFOLDING: a small dense host class that plays all three template roles
// (Matrix, MultiVector, DenseMatrix) of rails::Solver, so the solver template's host logic can be exercised
// without a GPU (tests/test_host_logic.py).  It implements the backend contract listed in SURVEY.md 8(b)
// with plain loops; random() uses the counter-based generator shared with the HIP library and the oracle.
#ifndef CPUDENSE_HPP
#define CPUDENSE_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <vector>

#include "rails_hip.h"

namespace cpu
{

inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline double counter_uniform(uint64_t seed, uint64_t stream, uint64_t row, uint64_t col)
{
    uint64_t h = splitmix64(seed ^ splitmix64(stream * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
    h = splitmix64(h ^ splitmix64(row * 0x9E3779B97F4A7C15ull + col * 0xC2B2AE3D27D4EB4Full + 1));
    return 2.0 * ((double)(h >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
}
struct Rng {
    uint64_t seed = 1, stream = 0;
};
inline Rng &rng()
{
    static Rng r;
    return r;
}

class CpuDense
{
    std::shared_ptr<std::vector<double>> buf_;
    size_t off_; // element offset of the window into buf_ (views)
    int m_, n_, m_max_, n_max_;
    int orth_;
    bool view_, trans_;

    double *p() const { return buf_ ? buf_->data() + off_ : nullptr; }

public:
    CpuDense() : off_(0), m_(-1), n_(-1), m_max_(-1), n_max_(-1), orth_(0), view_(false), trans_(false) {}
    CpuDense(int m, int n) : off_(0), m_(m), n_(n), m_max_(m), n_max_(n), orth_(0), view_(false), trans_(false)
    {
        buf_ = std::make_shared<std::vector<double>>((size_t)std::max(m, 0) * std::max(n, 0), 0.0);
    }
    CpuDense(CpuDense const &o) : off_(0), m_(o.m_), n_(o.n_), m_max_(o.m_max_), n_max_(o.n_), orth_(o.orth_), view_(false), trans_(o.trans_)
    {
        if (o.buf_) {
            n_max_ = std::max(o.n_, 0);
            buf_ = std::make_shared<std::vector<double>>((size_t)std::max(m_max_, 0) * std::max(n_max_, 1), 0.0);
            for (int j = 0; j < n_; ++j) memcpy(p() + (size_t)j * m_max_, o.p() + (size_t)j * o.m_max_, sizeof(double) * m_);
        }
    }
    CpuDense(CpuDense &&) = default;
    CpuDense(CpuDense const &o, int n) : CpuDense(o.M(), n) {}
    virtual ~CpuDense() {}

    CpuDense &operator=(CpuDense const &o)
    {
        if (!view_) {
            buf_ = o.buf_;
            off_ = o.off_;
            m_ = o.m_;
            n_ = o.n_;
            m_max_ = o.m_max_;
            n_max_ = o.n_max_;
            orth_ = o.orth_;
            trans_ = o.trans_;
            return *this;
        }
        int cols = std::min(n_, o.n_), rows = std::min(m_, o.m_);
        for (int j = 0; j < cols; ++j) memcpy(p() + (size_t)j * m_max_, o.p() + (size_t)j * o.m_max_, sizeof(double) * rows);
        return *this;
    }
    CpuDense &operator=(double v)
    {
        for (int j = 0; j < n_; ++j) std::fill_n(p() + (size_t)j * m_max_, m_, v);
        orth_ = 0;
        return *this;
    }
    CpuDense &operator*=(double s)
    {
        for (int j = 0; j < n_; ++j)
            for (int i = 0; i < m_; ++i) p()[i + (size_t)j * m_max_] *= s;
        orth_ = 0;
        return *this;
    }
    CpuDense &operator/=(double s) { return *this *= 1.0 / s; }
    CpuDense &axpy(double a, CpuDense const &o)
    {
        for (int j = 0; j < n_; ++j)
            for (int i = 0; i < m_; ++i) p()[i + (size_t)j * m_max_] += a * o.p()[i + (size_t)j * o.m_max_];
        orth_ = 0;
        return *this;
    }
    CpuDense &operator+=(CpuDense const &o) { return axpy(1.0, o); }
    CpuDense &operator-=(CpuDense const &o) { return axpy(-1.0, o); }
    CpuDense operator+(CpuDense const &o) const
    {
        CpuDense out(*this);
        out += o;
        return out;
    }

    operator double *() const { return p(); }
    double &operator()(int i, int j = 0) { return p()[i + (size_t)j * m_max_]; }
    double const &operator()(int i, int j = 0) const { return p()[i + (size_t)j * m_max_]; }
    int M() const { return trans_ ? n_ : m_; }
    int N() const { return trans_ ? m_ : n_; }
    int LDA() const { return trans_ ? n_max_ : m_max_; }

    void resize(int n) { resize(m_, n); }
    void resize(int m, int n)
    {
        orth_ = std::min(orth_, n);
        if (buf_ && m <= m_max_ && n <= n_max_) {
            m_ = m;
            n_ = n;
            return;
        }
        CpuDense out(m, n);
        if (buf_ && m_max_ > 0)
            for (int j = 0; j < std::min(n_, n); ++j) memcpy(out.p() + (size_t)j * out.m_max_, p() + (size_t)j * m_max_, sizeof(double) * std::min(m_, m));
        int orth = orth_;
        bool v = view_;
        view_ = false;
        *this = out;
        view_ = v;
        orth_ = orth;
    }

    CpuDense view(int a = -1, int b = -1) const
    {
        CpuDense out;
        out.buf_ = buf_;
        out.m_ = m_;
        out.m_max_ = m_max_;
        out.trans_ = trans_;
        out.view_ = true;
        int num = 1;
        if (b > 0 && a >= 0)
            num = b - a + 1;
        else if (a < 0) {
            a = 0;
            num = n_;
        }
        out.off_ = off_ + (size_t)a * m_max_;
        out.n_ = num;
        out.n_max_ = num;
        return out;
    }
    CpuDense copy() const { return CpuDense(*this); }
    CpuDense transpose() const
    {
        CpuDense out = view();
        out.view_ = false;
        out.n_max_ = n_max_;
        out.trans_ = !trans_;
        return out;
    }
    void push_back(CpuDense const &o)
    {
        int n = n_ < 0 ? 0 : n_;
        if (m_ < 0) {
            *this = CpuDense(o.m_, o.n_);
            n_ = 0;
            n = 0;
        }
        resize(n + o.n_);
        for (int j = 0; j < o.n_; ++j) memcpy(p() + (size_t)(n + j) * m_max_, o.p() + (size_t)j * o.m_max_, sizeof(double) * m_);
    }
    void random()
    {
        uint64_t s = rng().stream++;
        for (int i = 0; i < m_; ++i)
            for (int j = 0; j < n_; ++j) (*this)(i, j) = counter_uniform(rng().seed, s, i, j);
        orth_ = 0;
    }

    // op(this) * op(other)
    CpuDense operator*(CpuDense const &o) const
    {
        CpuDense out(M(), o.N());
        if (o.M() != N()) {
            std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << o.M() << "x" << o.N() << std::endl;
            return out;
        }
        for (int j = 0; j < o.N(); ++j)
            for (int i = 0; i < M(); ++i) {
                double s = 0.0;
                for (int l = 0; l < N(); ++l) {
                    double a = trans_ ? p()[l + (size_t)i * m_max_] : p()[i + (size_t)l * m_max_];
                    double b = o.trans_ ? o.p()[j + (size_t)l * o.m_max_] : o.p()[l + (size_t)j * o.m_max_];
                    s += a * b;
                }
                out(i, j) = s;
            }
        return out;
    }
    CpuDense dot(CpuDense const &o) const
    {
        CpuDense out(n_, o.n_);
        for (int j = 0; j < o.n_; ++j)
            for (int i = 0; i < n_; ++i) {
                double s = 0.0;
                for (int r = 0; r < m_; ++r) s += p()[r + (size_t)i * m_max_] * o.p()[r + (size_t)j * o.m_max_];
                out(i, j) = s;
            }
        return out;
    }
    int eigs(CpuDense &v, CpuDense &d, int num = -1, double tol = 1e-16) const
    {
        v = copy();
        int m = v.M();
        d.resize(m, 1);
        int info = 0;
        rails_dsyev('V', 'U', m, (double *)v, v.LDA(), (double *)d, &info);
        (void)num;
        (void)tol;
        return info;
    }
    double norm() const
    {
        if (n_ <= 0) return 0.0;
        CpuDense G = dot(*this), v, d(n_, 1);
        G.eigs(v, d);
        double mx = 0.0;
        for (int i = 0; i < n_; ++i) mx = std::max(mx, std::sqrt(std::abs(d(i, 0))));
        return mx;
    }
    double norm_inf() const
    {
        double out = 0.0;
        for (int i = 0; i < m_; ++i) {
            double s = 0.0;
            for (int j = 0; j < n_; ++j) s += std::abs((*this)(i, j));
            out = std::max(out, s);
        }
        return out;
    }
    void orthogonalize()
    {
        for (int i = orth_; i < N(); i++) {
            CpuDense v = view(i);
            v /= v.norm();
            if (i) {
                CpuDense V = view(0, i - 1);
                for (int k = 0; k < 2; k++) v -= V * V.dot(v);
            }
            v /= v.norm();
        }
        orth_ = N();
    }
};

inline CpuDense operator*(double d, CpuDense const &o)
{
    CpuDense out(o);
    out *= d;
    return out;
}

} // namespace cpu
#endif
