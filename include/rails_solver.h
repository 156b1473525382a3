/* ============================================================================
 * rails_solver.h -- C ABI of the RAILS solver instantiated on the HIP backend:
 *   rails::Solver<HipOperatorWrapper, HipMultiVectorWrapper, HostDenseMatrix>
 * (rails_amd/include/rails/), the drop-in counterpart of the reference's
 *   RAILS::Solver<Matrix, MultiVector, DenseMatrix>   (src/LyapunovSolverDecl.hpp:9-51).
 * This is what a non-C++ host (Python via ctypes, see rails_amd/solver.py) binds.
 * Return conventions as in rails_hip.h.
 * ==========================================================================*/
#ifndef RAILS_SOLVER_H
#define RAILS_SOLVER_H

#include "rails_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rails_solver rails_solver;

/* Solver(A, B, M) -- src/LyapunovSolverDecl.hpp:13-16.  A, M: operators created with rails_csr_create
 * (M may be NULL = identity; the caller keeps ownership).  B: host column-major block of the LOCAL rows,
 * m_local x p with leading dimension ldb.  m_global: global row count (= m_local on one GPU). */
int rails_solver_create(rails_ctx *ctx, rails_csr *A, rails_csr *M, const double *B_host, int64_t ldb, int p,
                        int64_t m_global, rails_solver **out);
int rails_solver_destroy(rails_solver *s);

/* set_parameters -- src/LyapunovSolver.hpp:72-98.  Names are the reference's ("Maximum iterations",
 * "Tolerance", "Expand size", "Lanczos iterations", "Restart size", "Reduced size", "Restart iterations",
 * "Restart tolerance", "Minimize solution space", "Restart from solution"; any capitalisation the
 * reference accepts).  Values are stored until rails_solver_apply_parameters, which returns the
 * reference's code (0 ok, 1 = Lanczos iterations <= Expand size). */
int rails_solver_set_parameter(rails_solver *s, const char *name, double value);
int rails_solver_apply_parameters(rails_solver *s, int *code);

/* extensions: "mass" (use M, generalized equation), "mass_orthogonalisation" (with mass: keep V M-orthonormal, V'MV = I, so that the
 * projected equation is the standard one -- `opts.ortho = 'M'` of matlab/RAILSsolver.m:38-41,384,583-597; default 0: orthonormal V and the
 * generalized projected equation by Cholesky reduction), "verbose", "max_trips", "projected_lanczos" (M = I only: carry the
 * residual Lanczos recurrence in coefficient space, see rails/HipSolverOps.hpp), "subspace" (default 1: run the solver template on the
 * coordinate-space back end of rails/SubspaceWrappers.hpp -- all multivectors as coordinates in one orthonormal device basis;
 * 0: the direct back end of rails/HipWrappers.hpp) */
int rails_solver_set_option(rails_solver *s, const char *name, double value);
/* called at the start of every loop trip with the index of that trip, and once after the last */
typedef void (*rails_trip_fn)(void *user, int trip);
int rails_solver_set_trip_callback(rails_solver *s, rails_trip_fn fn, void *user);

/* warm start: set V (host column-major, local rows x k); assumed orthonormal ("Restart from solution") */
int rails_solver_set_V(rails_solver *s, const double *V_host, int64_t ldv, int k);

/* solve(V, T) -- src/LyapunovSolver.hpp:100-346.  *code = 0 converged, -1 not converged, 1 loop exhausted,
 * 2 stopped by max_trips; *k = V.N(). */
int rails_solver_solve(rails_solver *s, int *code, int *k);

int rails_solver_get_V(rails_solver *s, double *V_host, int64_t ldv); /* local rows x k */
int rails_solver_get_T(rails_solver *s, double *T_host, int ldt);     /* k x k */
int rails_solver_trips(rails_solver *s);
int rails_solver_history(rails_solver *s, double *res, int cap);      /* Lanczos estimates per trip; returns count */

/* host wall-clock seconds per solver section of the last solve (JSON object; names follow the reference's profile
 * sections, src/Timer.hpp:101-106: "Apply A", "Apply B", "Compute VAV", "dense_solve", "Residual Lanczos", ...) */
int rails_solver_profile(rails_solver *s, char *buf, int cap);
/* JSON counters of the coordinate-space back end for the last solve ("{}" when the direct back end ran). */
const char *rails_solver_backend_stats(rails_solver *s);

/* ||A X + X A' + B B'||_F / ||B B'||_F for X = V T V' evaluated on the device without forming X
 * (uses R = [AV V B] G [AV V B]'; test / reporting helper).  The norm comes out of a trace of Gram products, i.e. as the
 * square root of a difference of O(||X||^2 ||A||^2) terms: it bottoms out near 1e-8 relative.  Below that use products with R
 * itself (tests/test_gpu_fullsize.py does a power iteration on R). */
int rails_solver_relative_residual(rails_solver *s, double *rel);

#ifdef __cplusplus
}
#endif
#endif
