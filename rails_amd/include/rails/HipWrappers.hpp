// HipMultiVectorWrapper / HipOperatorWrapper -- the MultiVector and Matrix template parameters of
// Solver<Matrix, MultiVector, DenseMatrix> for MI355X, header-only over the C ABI of librails_hip.so
// (include/rails_hip.h).  They drop in next to the reference's StlWrapper and Epetra_*Wrapper
// (src/LyapunovSolverDecl.hpp:9-51): every member the solver uses is provided with the reference's
// meaning (list in SURVEY.md section 8(b); semantics from src/StlWrapper.cpp).
//
//  * A HipMultiVectorWrapper is a column window [c0, c0+n) of a shared, ref-counted device panel
//    (row-major, rows partitioned over the GPUs).  `=` on a non-view shares the panel, on a view
//    copies into it; copy construction and copy() are deep copies; view(a,b) aliases columns
//    (src/StlWrapper.cpp:31-44,65-121,323-340).
//  * Results of `B.transpose() * W` are small p x w objects that are REPLICATED on every rank
//    (src/LyapunovSolver.hpp:150,156: BV "looks like B', not V").  They are held on the host inside
//    the same class (replicated_ == true), exactly as Epetra keeps them on a LocalMap
//    (src/Epetra_MultiVectorWrapper.cpp:226-228).
//  * norm() is the spectral 2-norm of the Stl backend (src/StlWrapper.cpp:265-289), not Epetra's
//    Frobenius norm, so convergence tests stop where the Stl path stops.
//  * Errors: no exceptions on the solver path; a failing device call prints rails_last_error() to
//    std::cerr and leaves the result unspecified, like the reference's shape errors
//    (src/StlWrapper.cpp:173-179).
#ifndef RAILS_HIPWRAPPERS_HPP
#define RAILS_HIPWRAPPERS_HPP

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <vector>

#include "rails/HostDenseMatrix.hpp"
#include "rails_hip.h"

namespace rails
{

inline rails_ctx *&default_context()
{
    static rails_ctx *ctx = nullptr;
    return ctx;
}
inline void set_default_context(rails_ctx *ctx) { default_context() = ctx; }

// The wrapper classes have the reference's error behaviour -- a message on stderr, no exception, a result that may be garbage
// (src/StlWrapper.cpp:173-179) -- which a caller of solve() cannot see.  Every failed library call therefore also latches its
// code here (first failure wins, per thread): rails_solver_solve() and C++ users check it after a solve (and may check it any
// time), and turn a run on zero-filled or partial panels into an error return instead of a normal-looking result.
inline int &sticky_error()
{
    static thread_local int code = RAILS_OK;
    return code;
}
inline void clear_sticky_error() { sticky_error() = RAILS_OK; }

inline bool hip_ok(int rc, const char *what)
{
    if (rc != RAILS_OK) {
        std::cerr << "rails_amd: " << what << " failed (" << rc << "): " << rails_last_error() << std::endl;
        if (sticky_error() == RAILS_OK) sticky_error() = rc;
        return false;
    }
    return true;
}

struct PanelHandle {
    rails_ctx *ctx;
    rails_panel *p;
    bool own;
    PanelHandle(rails_ctx *c, int64_t m, int cap) : ctx(c), p(nullptr), own(true) { hip_ok(rails_panel_create(c, m, cap, &p), "rails_panel_create"); }
    PanelHandle(rails_ctx *c, rails_panel *borrowed) : ctx(c), p(borrowed), own(false) {}
    ~PanelHandle()
    {
        if (p && own) rails_panel_destroy(p);
    }
    PanelHandle(PanelHandle const &) = delete;
    PanelHandle &operator=(PanelHandle const &) = delete;
};

class HipOperatorWrapper;

class HipMultiVectorWrapper
{
    friend class HipOperatorWrapper;

    rails_ctx *ctx_;
    std::shared_ptr<PanelHandle> panel_; // distributed form
    int64_t m_;                          // local rows (distributed) or rows (replicated)
    int c0_, n_;
    int orthogonalized_;
    bool is_view_;
    bool transpose_;
    // replicated small form: column-major m_ x cap on the host
    bool replicated_;
    std::shared_ptr<std::vector<double>> host_;
    int hcap_;
    int64_t m_global_ = -1; // global row count of a row-partitioned multivector (-1: same as m_)

    double *hdata() const { return host_->data() + (size_t)c0_ * m_; }

public:
    HipMultiVectorWrapper()
        : ctx_(default_context()), m_(-1), c0_(0), n_(-1), orthogonalized_(0), is_view_(false), transpose_(false), replicated_(false),
          hcap_(0)
    {
    }

    // m local rows, n columns (capacity n), on ctx (default context if null)
    HipMultiVectorWrapper(int64_t m, int n, rails_ctx *ctx = nullptr)
        : ctx_(ctx ? ctx : default_context()), m_(m), c0_(0), n_(n), orthogonalized_(0), is_view_(false), transpose_(false),
          replicated_(false), hcap_(0)
    {
        panel_ = std::make_shared<PanelHandle>(ctx_, m, std::max(n, 1));
    }

    // replicated p x n host object
    static HipMultiVectorWrapper Replicated(int p, int n, rails_ctx *ctx)
    {
        HipMultiVectorWrapper out;
        out.ctx_ = ctx;
        out.m_ = p;
        out.n_ = n;
        out.replicated_ = true;
        out.hcap_ = std::max(n, 1);
        out.host_ = std::make_shared<std::vector<double>>((size_t)p * out.hcap_, 0.0);
        return out;
    }

    // a view of columns [c0, c0 + n) of a panel somebody else owns (operator callbacks get their arguments this way)
    static HipMultiVectorWrapper Window(rails_ctx *ctx, rails_panel *panel, int c0, int n)
    {
        HipMultiVectorWrapper out;
        out.ctx_ = ctx;
        out.panel_ = std::make_shared<PanelHandle>(ctx, panel);
        out.m_ = rails_panel_rows(panel);
        out.c0_ = c0;
        out.n_ = n;
        out.is_view_ = true;
        return out;
    }

    // deep copy (src/StlWrapper.cpp:31-44)
    HipMultiVectorWrapper(HipMultiVectorWrapper const &o)
        : ctx_(o.ctx_), m_(o.m_), c0_(0), n_(o.n_), orthogonalized_(o.orthogonalized_), is_view_(false), transpose_(o.transpose_),
          replicated_(o.replicated_), hcap_(0), m_global_(o.m_global_)
    {
        if (o.replicated_) {
            hcap_ = std::max(o.n_, 1);
            host_ = std::make_shared<std::vector<double>>((size_t)m_ * hcap_, 0.0);
            if (o.host_ && o.n_ > 0) memcpy(host_->data(), o.hdata(), sizeof(double) * (size_t)m_ * o.n_);
        } else if (o.panel_) {
            panel_ = std::make_shared<PanelHandle>(ctx_, m_, std::max(o.capacity() - o.c0_, 1));
            if (o.n_ > 0) hip_ok(rails_panel_copy(ctx_, o.panel_->p, o.c0_, o.n_, panel_->p, 0), "rails_panel_copy");
        }
    }

    HipMultiVectorWrapper(HipMultiVectorWrapper &&o) = default;

    // same row map, n columns, capacity n (src/StlWrapper.cpp:46-51)
    HipMultiVectorWrapper(HipMultiVectorWrapper const &o, int n)
        : ctx_(o.ctx_), m_(o.m_), c0_(0), n_(n), orthogonalized_(0), is_view_(false), transpose_(false), replicated_(o.replicated_), hcap_(0),
          m_global_(o.m_global_)
    {
        if (replicated_) {
            hcap_ = std::max(n, 1);
            host_ = std::make_shared<std::vector<double>>((size_t)m_ * hcap_, 0.0);
        } else
            panel_ = std::make_shared<PanelHandle>(ctx_, m_, std::max(n, 1));
    }

    virtual ~HipMultiVectorWrapper() {}

    // ---- accessors used by the fused solver paths --------------------------------------------
    rails_ctx *context() const { return ctx_; }
    rails_panel *panel() const { return panel_ ? panel_->p : nullptr; }
    int offset() const { return c0_; }
    bool replicated() const { return replicated_; }
    int capacity() const { return replicated_ ? hcap_ : (panel_ ? rails_panel_capacity(panel_->p) : 0); }
    int orthogonalized() const { return orthogonalized_; }
    void set_orthogonalized(int n) { orthogonalized_ = n; }
    double *host_data() const { return hdata(); }

    // ---- assignment ----------------------------------------------------------------------------
    HipMultiVectorWrapper &operator=(HipMultiVectorWrapper const &o)
    {
        if (!is_view_) { // share
            ctx_ = o.ctx_;
            panel_ = o.panel_;
            host_ = o.host_;
            hcap_ = o.hcap_;
            replicated_ = o.replicated_;
            m_ = o.m_;
            m_global_ = o.m_global_;
            c0_ = o.c0_;
            n_ = o.n_;
            orthogonalized_ = o.orthogonalized_;
            transpose_ = o.transpose_;
            return *this;
        }
        int cols = std::min(n_, o.n_);
        if (cols <= 0) return *this;
        if (replicated_)
            memcpy(hdata(), o.hdata(), sizeof(double) * (size_t)m_ * cols);
        else
            hip_ok(rails_panel_copy(ctx_, o.panel_->p, o.c0_, cols, panel_->p, c0_), "rails_panel_copy");
        return *this;
    }

    HipMultiVectorWrapper &operator=(double v)
    {
        if (replicated_)
            std::fill_n(hdata(), (size_t)m_ * n_, v);
        else if (n_ > 0)
            hip_ok(rails_panel_fill(ctx_, panel_->p, c0_, n_, v), "rails_panel_fill");
        orthogonalized_ = 0;
        return *this;
    }

    HipMultiVectorWrapper &operator*=(double s)
    {
        if (replicated_)
            for (size_t i = 0; i < (size_t)m_ * n_; ++i) hdata()[i] *= s;
        else if (n_ > 0)
            hip_ok(rails_panel_scale(ctx_, panel_->p, c0_, n_, s), "rails_panel_scale");
        orthogonalized_ = 0;
        return *this;
    }
    HipMultiVectorWrapper &operator/=(double s) { return *this *= 1.0 / s; } // src/StlWrapper.cpp:139-143

    HipMultiVectorWrapper &operator+=(HipMultiVectorWrapper const &o) { return axpy(1.0, o); }
    HipMultiVectorWrapper &operator-=(HipMultiVectorWrapper const &o) { return axpy(-1.0, o); }

    HipMultiVectorWrapper &axpy(double a, HipMultiVectorWrapper const &o)
    {
        int cols = std::min(n_, o.n_);
        if (replicated_)
            for (size_t i = 0; i < (size_t)m_ * cols; ++i) hdata()[i] += a * o.hdata()[i];
        else if (cols > 0)
            hip_ok(rails_panel_axpy(ctx_, a, o.panel_->p, o.c0_, cols, panel_->p, c0_), "rails_panel_axpy");
        orthogonalized_ = 0;
        return *this;
    }

    HipMultiVectorWrapper operator+(HipMultiVectorWrapper const &o) const
    {
        HipMultiVectorWrapper out(*this);
        out += o;
        return out;
    }

    // ---- shape ---------------------------------------------------------------------------------
    int M() const { return (int)(transpose_ ? n_ : global_rows()); }
    int N() const { return (int)(transpose_ ? global_rows() : n_); }
    int64_t local_rows() const { return m_; }
    int64_t global_rows() const { return m_global_ >= 0 && !replicated_ ? m_global_ : m_; }
    void set_global_rows(int64_t mg) { m_global_ = mg; }

    void resize(int n) // capacity preserving (src/StlWrapper.cpp:219-263)
    {
        orthogonalized_ = std::min(orthogonalized_, n);
        if (replicated_) {
            if (c0_ + n > hcap_) {
                int ncap = c0_ + n;
                auto nb = std::make_shared<std::vector<double>>((size_t)m_ * ncap, 0.0);
                if (host_) memcpy(nb->data(), host_->data(), sizeof(double) * (size_t)m_ * std::min(hcap_, ncap));
                host_ = nb;
                hcap_ = ncap;
            }
        } else if (panel_) {
            if (c0_ + n > rails_panel_capacity(panel_->p) && !hip_ok(rails_panel_reserve(ctx_, panel_->p, c0_ + n), "rails_panel_reserve"))
                n = std::max(0, rails_panel_capacity(panel_->p) - c0_); // the failure is latched (sticky_error); stay inside the panel
        } else if (m_ >= 0) {
            panel_ = std::make_shared<PanelHandle>(ctx_, m_, std::max(n, 1));
        }
        n_ = n;
    }

    // view(i): column i; view(a,b): columns a..b inclusive; view(): all (src/StlWrapper.cpp:323-359)
    HipMultiVectorWrapper view(int a = -1, int b = -1) const
    {
        HipMultiVectorWrapper out;
        out.ctx_ = ctx_;
        out.panel_ = panel_;
        out.host_ = host_;
        out.hcap_ = hcap_;
        out.replicated_ = replicated_;
        out.m_ = m_;
        out.m_global_ = m_global_;
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
        out.orthogonalized_ = 0;
        return out;
    }

    HipMultiVectorWrapper copy() const { return HipMultiVectorWrapper(*this); }

    void push_back(HipMultiVectorWrapper const &o) // src/StlWrapper.cpp:367-374
    {
        int n = n_, on = o.n_;
        if (m_ < 0) { // empty default-constructed target takes the shape of the source
            m_ = o.m_;
            m_global_ = o.m_global_;
            replicated_ = o.replicated_;
            ctx_ = o.ctx_;
            n = 0;
            n_ = 0;
        }
        resize(n + on);
        if (on <= 0) return;
        if (replicated_)
            memcpy(hdata() + (size_t)n * m_, o.hdata(), sizeof(double) * (size_t)m_ * on);
        else
            hip_ok(rails_panel_copy(ctx_, o.panel_->p, o.c0_, on, panel_->p, c0_ + n), "rails_panel_copy");
    }

    void random() // src/StlWrapper.cpp:414-423 (counter-based generator, see rails_ctx_set_seed)
    {
        if (replicated_) {
            std::cerr << "rails_amd: random() on a replicated object is not supported" << std::endl;
            return;
        }
        if (n_ > 0) hip_ok(rails_panel_random(ctx_, panel_->p, c0_, n_), "rails_panel_random");
        orthogonalized_ = 0;
    }

    HipMultiVectorWrapper transpose() const // flag flip on a shallow alias (the reference deep-copies: SURVEY F8)
    {
        HipMultiVectorWrapper out = view();
        out.is_view_ = false;
        out.transpose_ = !transpose_;
        out.orthogonalized_ = orthogonalized_;
        return out;
    }

    // ---- products --------------------------------------------------------------------------------
    // X^T Y -> host dense (src/StlWrapper.cpp:394-412)
    HostDenseMatrix dot(HipMultiVectorWrapper const &o) const
    {
        HostDenseMatrix out(n_, o.n_);
        if (o.m_ != m_ || o.replicated_ != replicated_) {
            std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << o.M() << "x" << o.N() << std::endl;
            return out;
        }
        if (n_ <= 0 || o.n_ <= 0) return out;
        if (replicated_) {
            for (int j = 0; j < o.n_; ++j)
                for (int i = 0; i < n_; ++i) {
                    double s = 0.0;
                    for (int64_t r = 0; r < m_; ++r) s += hdata()[r + (size_t)i * m_] * o.hdata()[r + (size_t)j * m_];
                    out(i, j) = s;
                }
        } else
            hip_ok(rails_gram(ctx_, panel_->p, c0_, n_, o.panel_->p, o.c0_, o.n_, (double *)out, out.LDA()), "rails_gram");
        return out;
    }

    // this * DenseMatrix (src/StlWrapper.cpp:168-187)
    HipMultiVectorWrapper operator*(HostDenseMatrix const &C) const
    {
        HipMultiVectorWrapper out(*this, C.N());
        out.m_global_ = m_global_;
        if (C.M() != n_) {
            std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << C.M() << "x" << C.N() << std::endl;
            return out;
        }
        if (C.N() <= 0) return out;
        if (replicated_) {
            for (int j = 0; j < C.N(); ++j)
                for (int64_t r = 0; r < m_; ++r) {
                    double s = 0.0;
                    for (int l = 0; l < n_; ++l) s += hdata()[r + (size_t)l * m_] * C(l, j);
                    out.hdata()[r + (size_t)j * m_] = s;
                }
        } else
            // more than 256 output columns (a restart that keeps that many vectors): the sliced form; `out` is a panel of its own
            hip_ok((C.N() > 256 ? rails_panel_gemm_wide : rails_panel_gemm)(ctx_, 1.0, panel_->p, c0_, n_, (double *)C, C.LDA(), C.N(), 0.0, out.panel_->p, 0),
                   "rails_panel_gemm");
        return out;
    }

    // op(this) * other (src/MatrixOrMultiVectorWrapper.hpp:54,59): B^T W (replicated result) or B y (y replicated)
    HipMultiVectorWrapper operator*(HipMultiVectorWrapper const &o) const
    {
        if (transpose_) { // (n_ x m) * (m x o.n_): rows of the result = our columns
            HipMultiVectorWrapper out = Replicated(n_, o.n_, ctx_);
            if (o.m_ != m_ || o.replicated_ || replicated_) {
                std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << o.M() << "x" << o.N() << std::endl;
                return out;
            }
            if (n_ > 0 && o.n_ > 0)
                hip_ok(rails_gram(ctx_, panel_->p, c0_, n_, o.panel_->p, o.c0_, o.n_, out.hdata(), n_), "rails_gram");
            return out;
        }
        // (m x n_) * (n_ x o.n_) with o replicated
        HipMultiVectorWrapper out(*this, o.n_);
        out.m_global_ = m_global_;
        if (!o.replicated_ || o.m_ != n_) {
            std::cerr << "Incomplatible matrices of sizes " << M() << "x" << N() << " and " << o.M() << "x" << o.N() << std::endl;
            return out;
        }
        if (o.n_ > 0)
            hip_ok((o.n_ > 256 ? rails_panel_gemm_wide : rails_panel_gemm)(ctx_, 1.0, panel_->p, c0_, n_, o.hdata(), (int)o.m_, o.n_, 0.0, out.panel_->p, 0),
                   "rails_panel_gemm");
        return out;
    }

    // spectral 2-norm (src/StlWrapper.cpp:265-289)
    double norm() const
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

    // columns [watermark, N) against all previous ones (src/StlWrapper.cpp:305-321)
    void orthogonalize()
    {
        if (replicated_ || !panel_ || c0_ != 0) {
            std::cerr << "rails_amd: orthogonalize() needs a distributed multivector starting at column 0" << std::endl;
            return;
        }
        if (n_ > orthogonalized_) hip_ok(rails_orthogonalize(ctx_, panel_->p, orthogonalized_, n_ - orthogonalized_, 0, nullptr), "rails_orthogonalize");
        orthogonalized_ = n_;
    }

    // host round trips (tests, I/O)
    void from_host(const double *data, int64_t ld)
    {
        if (replicated_)
            for (int j = 0; j < n_; ++j) memcpy(hdata() + (size_t)j * m_, data + (size_t)j * ld, sizeof(double) * m_);
        else if (n_ > 0)
            hip_ok(rails_panel_upload(ctx_, panel_->p, c0_, n_, data, ld), "rails_panel_upload");
        orthogonalized_ = 0;
    }
    void to_host(double *data, int64_t ld) const
    {
        if (replicated_)
            for (int j = 0; j < n_; ++j) memcpy(data + (size_t)j * ld, hdata() + (size_t)j * m_, sizeof(double) * m_);
        else if (n_ > 0)
            hip_ok(rails_panel_download(ctx_, panel_->p, c0_, n_, data, ld), "rails_panel_download");
    }
};

inline HipMultiVectorWrapper operator*(double d, HipMultiVectorWrapper const &o) // src/StlWrapper.cpp:481-487
{
    HipMultiVectorWrapper out(o);
    out *= d;
    return out;
}

struct CsrHandle {
    rails_ctx *ctx;
    rails_csr *A;
    bool own;
    std::function<bool(bool, HipMultiVectorWrapper const &, HipMultiVectorWrapper &)> apply; // callback operators only
    CsrHandle(rails_ctx *c, rails_csr *a, bool o = true) : ctx(c), A(a), own(o) {}
    ~CsrHandle()
    {
        if (A && own) rails_csr_destroy(A);
    }
    CsrHandle(CsrHandle const &) = delete;
    CsrHandle &operator=(CsrHandle const &) = delete;
};

// The Matrix role: a cheap handle (the solver stores copies of A and M, src/LyapunovSolverDecl.hpp:37-39)
class HipOperatorWrapper
{
    rails_ctx *ctx_;
    std::shared_ptr<CsrHandle> h_;
    bool transpose_;
    int64_t m_global_;

public:
    HipOperatorWrapper() : ctx_(default_context()), transpose_(false), m_global_(-1) {}

    // local CSR block (host arrays are copied to the device)
    HipOperatorWrapper(rails_ctx *ctx, int64_t m_local, int64_t n_cols_ext, const int64_t *rowptr, const int32_t *col, const double *val,
                       int64_t m_global = -1)
        : ctx_(ctx ? ctx : default_context()), transpose_(false), m_global_(m_global < 0 ? m_local : m_global)
    {
        rails_csr *A = nullptr;
        if (hip_ok(rails_csr_create(ctx_, m_local, n_cols_ext, rowptr, col, val, &A), "rails_csr_create")) h_ = std::make_shared<CsrHandle>(ctx_, A);
    }

    // borrow an operator created through the C ABI (the caller keeps ownership)
    HipOperatorWrapper(rails_ctx *ctx, rails_csr *A, int64_t m_global = -1)
        : ctx_(ctx ? ctx : default_context()), transpose_(false), m_global_(m_global < 0 && A ? rails_csr_rows(A) : m_global)
    {
        if (A) h_ = std::make_shared<CsrHandle>(ctx_, A, false);
    }

    // An operator given by its action: apply(transposed, X, Y) must write op(A) * X into Y (both are views of device panels, work
    // on the context's stream) and return true.  The counterpart of wrapping any Epetra_Operator (src/Epetra_OperatorWrapper.cpp:75-91).
    typedef std::function<bool(bool, HipMultiVectorWrapper const &, HipMultiVectorWrapper &)> ApplyFunction;
    static HipOperatorWrapper FromCallback(rails_ctx *ctx, int64_t m_local, ApplyFunction apply, int64_t m_global = -1)
    {
        HipOperatorWrapper out;
        out.ctx_ = ctx ? ctx : default_context();
        out.m_global_ = m_global < 0 ? m_local : m_global;
        auto h = std::make_shared<CsrHandle>(out.ctx_, nullptr);
        h->apply = std::move(apply);
        if (hip_ok(rails_csr_create_callback(out.ctx_, m_local, &HipOperatorWrapper::trampoline, h.get(), &h->A), "rails_csr_create_callback")) out.h_ = h;
        return out;
    }

    virtual ~HipOperatorWrapper() {}

    rails_csr *csr() const { return h_ ? h_->A : nullptr; }
    bool transposed() const { return transpose_; }
    rails_ctx *context() const { return ctx_; }

    int M() const { return (int)m_global_; }
    int N() const { return (int)m_global_; }

    HipOperatorWrapper transpose() const // flag flip (src/Epetra_OperatorWrapper.cpp)
    {
        HipOperatorWrapper out(*this);
        out.transpose_ = !transpose_;
        return out;
    }

    // Set-up for many products with panels of nc columns (rails_csr_prepare: the sweep kernel's schedule on banded patterns, which an
    // operator would otherwise only build after its 16th such product).  True when that kernel will take them.  No reference counterpart.
    bool prepare(int nc) const
    {
        int ready = 0;
        if (!h_ || !hip_ok(rails_csr_prepare(ctx_, h_->A, transpose_ ? 1 : 0, nc, &ready), "rails_csr_prepare")) return false;
        return ready != 0;
    }

private:
    static int trampoline(void *user, int trans, const rails_panel *X, int xc0, int nc, rails_panel *Y, int yc0)
    {
        CsrHandle *h = static_cast<CsrHandle *>(user);
        HipMultiVectorWrapper x = HipMultiVectorWrapper::Window(h->ctx, const_cast<rails_panel *>(X), xc0, nc);
        HipMultiVectorWrapper y = HipMultiVectorWrapper::Window(h->ctx, Y, yc0, nc);
        return h->apply(trans != 0, x, y) ? 0 : RAILS_EINVAL;
    }

public:
    // A * X (src/LyapunovSolver.hpp:146)
    HipMultiVectorWrapper operator*(HipMultiVectorWrapper const &X) const
    {
        HipMultiVectorWrapper out(X, X.n_);
        out.m_global_ = X.m_global_;
        if (!h_) {
            std::cerr << "rails_amd: operator* on an empty HipOperatorWrapper" << std::endl;
            return out;
        }
        apply_into(X, out, 0);
        return out;
    }

    // Y[:, yc0:] = op(A) X : lets the solver write A*W straight into AV's tail
    bool apply_into(HipMultiVectorWrapper const &X, HipMultiVectorWrapper &Y, int ycol) const
    {
        if (X.n_ <= 0) return true;
        return hip_ok(rails_spmm(ctx_, h_->A, transpose_ ? 1 : 0, X.panel_->p, X.c0_, X.n_, Y.panel_->p, Y.c0_ + ycol), "rails_spmm");
    }

    // 2-norm of the operator, needed only when B is given as a Matrix (src/MatrixOrMultiVectorWrapper.hpp:33-38):
    // power iteration on A^T A with the device kernels.
    double norm() const
    {
        if (!h_) return 0.0;
        int64_t m = rails_csr_rows(h_->A);
        HipMultiVectorWrapper x(m, 1, ctx_), y(m, 1, ctx_);
        x.random();
        double lam = 0.0;
        for (int it = 0; it < 200; ++it) {
            double nx = x.norm();
            if (nx == 0.0) return 0.0;
            x /= nx;
            rails_spmm(ctx_, h_->A, transpose_ ? 1 : 0, x.panel_->p, 0, 1, y.panel_->p, 0);
            rails_spmm(ctx_, h_->A, transpose_ ? 0 : 1, y.panel_->p, 0, 1, x.panel_->p, 0);
            double l2 = y.norm();
            if (std::abs(l2 - lam) <= 1e-14 * l2) {
                lam = l2;
                break;
            }
            lam = l2;
        }
        return lam;
    }
};

} // namespace rails

#endif
