// solver_capi.cpp -- C ABI (include/rails_solver.h) over rails::Solver instantiated on the HIP backend.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <exception>
#include <new>

#include "rails/HipSolverOps.hpp"
#include "rails/SubspaceSolverOps.hpp"
#include "rails_solver.h"

void rails_set_error(const char *fmt, ...);

namespace
{

// the duck-typed ParameterList of the reference's tests (test/LyapunovSolver_test.cpp:160-179)
class ParameterList
{
    std::map<std::string, double> params_;

public:
    template <typename T>
    T get(std::string const &name, T def)
    {
        auto it = params_.find(name);
        if (it == params_.end()) return def;
        return (T)it->second;
    }
    void set(std::string const &name, double val) { params_[name] = val; }
};

} // namespace

struct rails_solver {
    rails_ctx *ctx = nullptr;
    rails::HipOperatorWrapper A, M;
    rails::HipMultiVectorWrapper B;
    rails::HipSolver *solver = nullptr;
    ParameterList params;
    rails::HipMultiVectorWrapper V;
    rails::HostDenseMatrix T;
    int64_t m_local = 0, m_global = 0;
    bool has_M = false;
    bool mass = false;
    bool ortho_m = false; // V kept M-orthonormal (matlab/RAILSsolver.m opts.ortho = 'M'); needs mass
    rails_trip_fn trip_fn = nullptr;
    void *trip_user = nullptr;
    // coordinate-space back end (rails/SubspaceWrappers.hpp): used by solve() when asked for and applicable
    bool subspace = true, verbose = true, projected = false; // default: coordinate-space back end where applicable (M = I, cold start)
    int max_trips = 0;
    bool have_V0 = false;
    bool last_was_subspace = false;
    int sub_trips = 0;
    std::vector<double> sub_hist;
    std::map<std::string, double> sub_profile;
    std::string sub_stats;
};

extern "C" int rails_solver_create(rails_ctx *ctx, rails_csr *A, rails_csr *M, const double *B_host, int64_t ldb, int p, int64_t m_global,
                                   rails_solver **out)
try {
    if (!ctx || !A || !out || (p > 0 && !B_host) || p < 0) {
        rails_set_error("rails_solver_create: bad argument");
        return RAILS_EINVAL;
    }
    int64_t m = rails_csr_rows(A);
    if (ldb < m) {
        rails_set_error("rails_solver_create: ldb %lld < local rows %lld", (long long)ldb, (long long)m);
        return RAILS_EINVAL;
    }
    if (M && rails_csr_rows(M) != m) {
        rails_set_error("rails_solver_create: M has %lld rows, A %lld", (long long)rails_csr_rows(M), (long long)m);
        return RAILS_EINVAL;
    }
    rails_solver *s = new rails_solver();
    s->ctx = ctx;
    s->m_local = m;
    s->m_global = m_global > 0 ? m_global : m;
    s->A = rails::HipOperatorWrapper(ctx, A, s->m_global);
    s->has_M = (M != nullptr);
    s->M = M ? rails::HipOperatorWrapper(ctx, M, s->m_global) : s->A;
    s->B = rails::HipMultiVectorWrapper(m, p, ctx);
    s->B.set_global_rows(s->m_global);
    s->B.from_host(B_host, ldb);
    s->solver = new rails::HipSolver(s->A, s->B, s->M);
    s->V = rails::HipMultiVectorWrapper(m, 1, ctx);
    s->V.set_global_rows(s->m_global);
    *out = s;
    return RAILS_OK;
} catch (std::bad_alloc const &) {
    rails_set_error("rails_solver_create: out of host memory");
    return RAILS_ENOMEM;
} catch (std::exception const &e) {
    rails_set_error("rails_solver_create: %s", e.what());
    return RAILS_EINVAL;
}

extern "C" int rails_solver_destroy(rails_solver *s)
{
    if (!s) return RAILS_OK;
    delete s->solver;
    delete s;
    return RAILS_OK;
}

extern "C" int rails_solver_set_parameter(rails_solver *s, const char *name, double value)
{
    if (!s || !name) return RAILS_EINVAL;
    s->params.set(name, value);
    return RAILS_OK;
}

extern "C" int rails_solver_apply_parameters(rails_solver *s, int *code)
{
    if (!s) return RAILS_EINVAL;
    int rc = s->solver->set_parameters(s->params);
    if (code) *code = rc;
    return RAILS_OK;
}

extern "C" int rails_solver_set_option(rails_solver *s, const char *name, double value)
{
    if (!s || !name) return RAILS_EINVAL;
    std::string n(name);
    if (n == "mass") {
        if (value != 0.0 && !s->has_M) {
            rails_set_error("rails_solver_set_option: mass requested but the solver was created without M");
            return RAILS_EINVAL;
        }
        s->mass = value != 0.0;
        s->solver->use_mass_matrix(s->mass);
    } else if (n == "mass_orthogonalisation" || n == "ortho_m") {
        if (value != 0.0 && !s->mass) {
            rails_set_error("rails_solver_set_option: mass_orthogonalisation needs the option mass (set it first)");
            return RAILS_EINVAL;
        }
        s->ortho_m = value != 0.0;
        s->solver->use_mass_orthogonalisation(s->ortho_m);
    } else if (n == "verbose") {
        s->verbose = value != 0.0;
        s->solver->set_verbose(s->verbose);
    } else if (n == "max_trips") {
        s->max_trips = (int)value;
        s->solver->set_max_trips(s->max_trips);
    } else if (n == "projected_lanczos") {
        s->projected = value != 0;
        s->solver->set_projected_lanczos(s->projected);
    } else if (n == "subspace")
        s->subspace = value != 0;
    else {
        rails_set_error("rails_solver_set_option: unknown option '%s'", name);
        return RAILS_EINVAL;
    }
    return RAILS_OK;
}

extern "C" int rails_solver_set_trip_callback(rails_solver *s, rails_trip_fn fn, void *user)
{
    if (!s) return RAILS_EINVAL;
    s->trip_fn = fn;
    s->trip_user = user;
    if (fn)
        s->solver->set_trip_callback([s](int trip) { s->trip_fn(s->trip_user, trip); });
    else
        s->solver->set_trip_callback(std::function<void(int)>());
    return RAILS_OK;
}

extern "C" int rails_solver_set_V(rails_solver *s, const double *V_host, int64_t ldv, int k)
{
    if (!s || !V_host || k < 1 || ldv < s->m_local) {
        rails_set_error("rails_solver_set_V: bad argument");
        return RAILS_EINVAL;
    }
    s->V = rails::HipMultiVectorWrapper(s->m_local, k, s->ctx);
    s->V.set_global_rows(s->m_global);
    s->V.from_host(V_host, ldv);
    s->V.set_orthogonalized(k); // the caller's V is assumed orthonormal
    s->have_V0 = true;
    return RAILS_OK;
}

// the same solve on the coordinate-space back end: B and every later vector are expressed in one orthonormal device basis
static int solve_in_coordinates(rails_solver *s)
{
    const int p = s->B.N();
    const int restart = s->params.get("Restart size", -1);
    const int expand = s->params.get("Expand size", 3);
    const int kmax = std::max(restart > 0 ? restart : 100, 1) + expand + 100;
    auto basis = std::make_shared<rails::SubspaceBasis>(s->ctx, s->m_local, s->m_global, 2 * kmax + p + 128);
    if (restart > 0 || s->params.get("Restart iterations", 20) > 0) basis->preallocate_compress_panel(); // restarts will re-base the basis
    rails::SubspaceMultiVector Bc = rails::SubspaceMultiVector::Absorb(basis, s->B);
    rails::SubspaceOperator Ac(s->A, basis);
    rails::SubspaceOperator Mc(s->mass ? s->M : s->A, basis);
    rails::SubspaceSolver solver(Ac, Bc, Mc);
    int prc = solver.set_parameters(s->params);
    if (prc != 0) return prc + 100;
    solver.set_verbose(s->verbose);
    solver.set_max_trips(s->max_trips);
    solver.use_mass_matrix(s->mass);
    solver.use_mass_orthogonalisation(s->ortho_m);
    if (s->trip_fn) solver.set_trip_callback([s](int trip) { s->trip_fn(s->trip_user, trip); });
    solver.set_failure_check([basis]() { return basis->failed; }); // a latched failure of the basis ends the run at the next trip
    rails::SubspaceMultiVector Vc(basis, 1);
    if (s->have_V0) { // warm start: the caller's (orthonormal) V expressed in the basis (src/LyapunovSolver.hpp:116-123)
        Vc = rails::SubspaceMultiVector::Absorb(basis, s->V);
        Vc.set_orthogonalized(Vc.N());
    }
    int rc = solver.solve(Vc, s->T);
    s->V = Vc.materialise();
    s->V.set_orthogonalized(s->V.N());
    s->sub_trips = solver.trips();
    s->sub_hist = solver.residual_history();
    s->sub_profile = solver.profile();
    char buf[1536];
    snprintf(buf, sizeof(buf), "{\"dim\": %d, \"absorb\": %ld, \"absorb_columns\": %ld, \"one_by_one\": %ld, \"dropped\": %ld, \"compress\": %ld, \"materialise\": %ld, \"prefetched_random\": %ld, \"second_rounds\": %ld, \"delicate_blocks\": %ld, \"reprojected_blocks\": %ld, \"overlapped_blocks\": %ld, \"replaced_columns\": %ld, \"verified\": %d, \"verify_representation\": %.3e, \"verify_orthonormality\": %.3e, \"seconds\": {\"materialise\": %.4f, \"absorb\": %.4f, \"compress_qr\": %.4f, \"compress_rotate\": %.4f, "
             "\"compress_coefficients\": %.4f}}",
             basis->dim, basis->n_absorb, basis->n_absorb_cols, basis->n_single, basis->n_dropped, basis->n_compress, basis->n_materialise, basis->n_prefetched, basis->n_second_round, basis->n_delicate, basis->n_reprojected, basis->n_overlapped, basis->n_replaced, basis->verify ? 1 : 0, basis->verify_repr, basis->verify_orth,
             basis->t_materialise, basis->t_absorb, basis->t_qr, basis->t_rotate, basis->t_recoef);
    s->sub_stats = buf;
    if (basis->failed) {
        rails_set_error("coordinate-space back end: a device operation failed (%s)", rails_last_error());
        return -1000;
    }
    return rc;
}

extern "C" int rails_solver_solve(rails_solver *s, int *code, int *k)
{
    if (!s) return RAILS_EINVAL;
    const bool restart_from_solution = s->params.get("Restart from solution", 0.0) != 0.0;
    // a warm start needs the caller's V; "Restart from solution" without one is the direct back end's business (it starts from
    // the single column the solver object holds, like the reference)
    s->last_was_subspace = s->subspace && (s->have_V0 || !restart_from_solution);
    rails::clear_sticky_error();
    int rc;
    try { // the solver templates allocate (std::vector, make_shared): nothing may unwind through the C boundary
        if (s->last_was_subspace) {
            rc = solve_in_coordinates(s);
            if (rc == -1000) return RAILS_EHIP;
        } else
            rc = s->solver->solve(s->V, s->T);
    } catch (std::bad_alloc const &) {
        rails_set_error("rails_solver_solve: out of host memory");
        return RAILS_ENOMEM;
    } catch (std::exception const &e) {
        rails_set_error("rails_solver_solve: %s", e.what());
        return RAILS_EINVAL;
    }
    s->have_V0 = false; // V now holds the result; a later warm start passes its V again (or runs on the direct back end)
    if (code) *code = rc;
    if (k) *k = s->V.N();
    if (rails_ctx_sync(s->ctx) != RAILS_OK) return RAILS_EHIP;
    // a library call that failed somewhere inside the wrappers (they only print, like the reference's): V and T are not to be trusted
    if (rails::sticky_error() != RAILS_OK) {
        const int err = rails::sticky_error();
        rails_set_error("rails_solver_solve: a device operation failed during the solve (code %d, first failure; see stderr): %s", err, rails_last_error());
        return err;
    }
    return RAILS_OK;
}

extern "C" int rails_solver_get_V(rails_solver *s, double *V_host, int64_t ldv)
{
    if (!s || !V_host || ldv < s->m_local) return RAILS_EINVAL;
    s->V.to_host(V_host, ldv);
    return RAILS_OK;
}

extern "C" int rails_solver_get_T(rails_solver *s, double *T_host, int ldt)
{
    if (!s || !T_host) return RAILS_EINVAL;
    int k = s->T.M();
    if (ldt < k) return RAILS_EINVAL;
    for (int j = 0; j < k; ++j)
        for (int i = 0; i < k; ++i) T_host[i + (size_t)j * ldt] = s->T(i, j);
    return RAILS_OK;
}

extern "C" int rails_solver_trips(rails_solver *s) { return s ? (s->last_was_subspace ? s->sub_trips : s->solver->trips()) : -1; }

extern "C" const char *rails_solver_backend_stats(rails_solver *s) { return (s && s->last_was_subspace) ? s->sub_stats.c_str() : "{}"; }

extern "C" int rails_solver_history(rails_solver *s, double *res, int cap)
{
    if (!s) return -1;
    auto const &h = s->last_was_subspace ? s->sub_hist : s->solver->residual_history();
    int n = (int)h.size();
    for (int i = 0; i < n && i < cap; ++i) res[i] = h[i];
    return n;
}

extern "C" int rails_solver_profile(rails_solver *s, char *buf, int cap)
{
    if (!s || !buf || cap < 2) return RAILS_EINVAL;
    std::string out = "{";
    bool first = true;
    for (auto const &kv : (s->last_was_subspace ? s->sub_profile : s->solver->profile())) {
        char tmp[256];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": %.6f", first ? "" : ", ", kv.first.c_str(), kv.second);
        out += tmp;
        first = false;
    }
    out += "}";
    if ((int)out.size() + 1 > cap) return RAILS_EINVAL;
    memcpy(buf, out.c_str(), out.size() + 1);
    return RAILS_OK;
}

// ||R||_F with R = P G P^T, P = [AV MV B], G = [[0 T 0],[T 0 0],[0 0 I]]:  ||R||_F^2 = tr(G S G S), S = P^T P
extern "C" int rails_solver_relative_residual(rails_solver *s, double *rel)
{
    if (!s || !rel) return RAILS_EINVAL;
    int k = s->V.N(), p = s->B.N();
    if (k != s->T.M()) {
        rails_set_error("rails_solver_relative_residual: V has %d columns but T is %d x %d", k, s->T.M(), s->T.N());
        return RAILS_EINVAL;
    }
    rails::HipMultiVectorWrapper AV = s->A * s->V;
    rails::HipMultiVectorWrapper MV = s->mass ? (s->M * s->V) : s->V.view();
    int n = 2 * k + p;
    std::vector<double> S((size_t)n * n, 0.0);
    rails::HipMultiVectorWrapper const *blk[3] = {&AV, &MV, &s->B};
    int off[3] = {0, k, 2 * k};
    for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) {
            rails::HostDenseMatrix C = blk[a]->dot(*blk[b]);
            for (int j = 0; j < C.N(); ++j)
                for (int i = 0; i < C.M(); ++i) {
                    S[(off[a] + i) + (size_t)(off[b] + j) * n] = C(i, j);
                    S[(off[b] + j) + (size_t)(off[a] + i) * n] = C(i, j);
                }
        }
    // GS = G * S
    std::vector<double> GS((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < k; ++i) {
            double s1 = 0.0, s2 = 0.0;
            for (int l = 0; l < k; ++l) {
                s1 += s->T(i, l) * S[(k + l) + (size_t)j * n]; // row block 0: T * S[MV rows]
                s2 += s->T(i, l) * S[l + (size_t)j * n];       // row block 1: T * S[AV rows]
            }
            GS[i + (size_t)j * n] = s1;
            GS[(k + i) + (size_t)j * n] = s2;
        }
        for (int i = 0; i < p; ++i) GS[(2 * k + i) + (size_t)j * n] = S[(2 * k + i) + (size_t)j * n];
    }
    double tr = 0.0; // tr(GS * GS)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) tr += GS[i + (size_t)j * n] * GS[j + (size_t)i * n];
    double bb = 0.0; // ||B B^T||_F^2 = ||B^T B||_F^2
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < p; ++j) {
            double v = S[(2 * k + i) + (size_t)(2 * k + j) * n];
            bb += v * v;
        }
    *rel = std::sqrt(std::fabs(tr)) / std::sqrt(bb);
    return RAILS_OK;
}
