/* ============================================================================
 * rails_hip.h -- C ABI of librails_hip.so: the MI355X (gfx950) back end of the
 * RAILS inner loop.
 *
 * This is the drop-in boundary.  The reference (Sbte/RAILS) has no FFI: its
 * plug-in seam is C++ template duck typing, Solver<Matrix, MultiVector,
 * DenseMatrix> (src/LyapunovSolverDecl.hpp:9-51).  The header-only wrappers in
 * rails_amd/include/rails/ (HipOperatorWrapper, HipMultiVectorWrapper,
 * HostDenseMatrix) satisfy that contract and are implemented purely in terms of
 * the entry points below, so this C ABI is exactly what a binding for the hot
 * path binds.  Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative RAILS_E* code otherwise
 *     (never throws); rails_last_error() gives the message of the calling
 *     thread's last failure.  This mirrors the reference's "no exceptions, int
 *     codes, message on stderr" behaviour (src/StlWrapper.cpp:173-179).
 *   - host matrices cross the boundary COLUMN-MAJOR with a leading dimension,
 *     like the reference's StlWrapper / DenseMatrix storage
 *     (src/StlVector.cpp:47-50, operator double* at src/StlWrapper.cpp:201).
 *   - device panels are ROW-MAJOR m_local x ld (ld = padded column capacity):
 *     one sparse nonzero gathers one contiguous row segment, reductions run
 *     down the slow index, and view/resize/push_back are O(1) column-window
 *     changes.  The solver never sees MultiVector memory
 *     (src/LyapunovSolver.hpp uses double* only from DenseMatrix, :357,:458),
 *     so the layout is free.
 *   - the library owns device buffers until *_destroy; the caller owns host
 *     buffers.  Calls that return host values are synchronisation points;
 *     everything else is asynchronous on the context's stream.
 *   - row-partitioned multi-GPU runs: every rank holds a contiguous block of
 *     rows of A, V, AV, B; small (k x k, p x k) objects are replicated.  All
 *     reductions over rows go through the all-reduce hook (sum of doubles).
 * ==========================================================================*/
#ifndef RAILS_HIP_H
#define RAILS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAILS_OK 0
#define RAILS_EINVAL -1   /* bad argument / shape mismatch                      */
#define RAILS_EHIP -2     /* a HIP runtime call failed                          */
#define RAILS_ENOMEM -3   /* device or host allocation failed                   */
#define RAILS_ELAPACK -4  /* host LAPACK missing or returned an error           */
#define RAILS_ECOMM -5    /* all-reduce / halo hook failed                      */
#define RAILS_ENODEV -6   /* no usable gfx950 device                            */

typedef struct rails_ctx rails_ctx;     /* device, stream, workspaces, RNG, collectives */
typedef struct rails_csr rails_csr;     /* device CSR operator (the Matrix role)        */
typedef struct rails_panel rails_panel; /* device row-partitioned panel (MultiVector)   */

const char *rails_last_error(void);
const char *rails_version(void);

/* ---------------------------------------------------------------- context --- */

/* device: HIP device ordinal.  stream: a hipStream_t to run on (e.g. the caller's
 * torch stream), or NULL to let the library create its own. */
int rails_ctx_create(int device, void *stream, rails_ctx **out);
int rails_ctx_destroy(rails_ctx *ctx);
int rails_ctx_sync(rails_ctx *ctx);
void *rails_ctx_stream(rails_ctx *ctx);
/* call counters of this context as a JSON object (block vs column-wise orthogonalisations, SpMM kernel choice, ...) */
int rails_ctx_stats(rails_ctx *ctx, char *buf, int cap);
/* Busy meter: with on != 0 every kernel launch of this context is bracketed by a pair of events, read at the next synchronisation of
 * its stream; "gpu_busy_ms" of rails_ctx_stats is the sum: the time the GPU was at work for this context.  Costs a few microseconds
 * per launch: for measurement runs. */
int rails_ctx_set_meter(rails_ctx *ctx, int on);

/* Counter-based RNG: value = f(seed, stream id, GLOBAL row, column); every random
 * fill consumes one stream id.  Replaces StlWrapper::random's std::rand()-seeded
 * generator (src/StlWrapper.cpp:414-423) by one that is identical on CPU and GPU
 * and independent of the row partition. */
int rails_ctx_set_seed(rails_ctx *ctx, uint64_t seed, uint64_t first_stream);
/* Current generator position (seed, id of the next stream): lets a caller repeat a draw. */
int rails_ctx_rng_state(rails_ctx *ctx, uint64_t *seed, uint64_t *next_stream);

/* Row partition of this rank: local rows are global rows [row0, row0 + m_local). */
int rails_ctx_set_partition(rails_ctx *ctx, int rank, int nranks, int64_t row0, int64_t m_global);

/* Sum-all-reduce hook over the ranks, in place on a DEVICE buffer of n doubles,
 * ordered on `stream`.  With nranks == 1 no hook is needed.  This is the one
 * collective of the path: Epetra hides it in Multiply('T','N')/Norm2
 * (src/Epetra_MultiVectorWrapper.cpp:238,312,431). */
typedef int (*rails_allreduce_fn)(void *user, double *dev_buf, size_t n, void *stream);
int rails_ctx_set_allreduce(rails_ctx *ctx, rails_allreduce_fn fn, void *user);

/* The same all-reduce (and the ghost-row exchange of rails_spmm) done by the library itself over RCCL -- xGMI inside a node --
 * on the context's stream, with no hook and no Python in the loop.  Either let the library make the communicator: rank 0 calls
 * rails_rccl_unique_id, the application hands the RAILS_RCCL_ID_BYTES bytes to every rank (any way it likes: MPI, files, a
 * torch.distributed broadcast), then EVERY rank calls rails_ctx_init_rccl (collective; device = the context's); or adopt one
 * the application already has (an ncclComm_t; the caller keeps ownership).  A hook installed with rails_ctx_set_allreduce
 * takes precedence.  librccl.so.1 is loaded at run time (RAILS_RCCL_LIB overrides the name). */
#define RAILS_RCCL_ID_BYTES 128
int rails_rccl_unique_id(void *id_out);
int rails_ctx_init_rccl(rails_ctx *ctx, const void *id, int nranks, int rank);
int rails_ctx_set_rccl(rails_ctx *ctx, void *nccl_comm);
int rails_ctx_rccl_size(const rails_ctx *ctx); /* ranks of the communicator, 0 without one */

/* Halo hook for the row-partitioned operator apply: given the packed rows this rank
 * must send (send_buf, concatenated per destination rank as described by the counts
 * passed to rails_csr_set_halo) fill recv_buf with the ghost rows, on `stream`.
 * Replaces the import inside Epetra_CrsMatrix::Apply (src/Epetra_OperatorWrapper.cpp:87). */
typedef int (*rails_halo_fn)(void *user, const double *send_buf, double *recv_buf, int ncols, void *stream);

/* ------------------------------------------------------------ CSR operator --- */

/* Upload a local CSR block: m_local rows, column indices in [0, n_cols_ext) where
 * columns [0, m_local) are the local rows and [m_local, n_cols_ext) are ghost rows
 * (n_cols_ext == m_local on one GPU).  rowptr has m_local+1 entries.
 * Role: the Matrix template parameter (src/LyapunovSolverDecl.hpp:37,
 * Epetra_OperatorWrapper.cpp:75-91; StlWrapper.cpp:168-187 for the dense Stl form). */
int rails_csr_create(rails_ctx *ctx, int64_t m_local, int64_t n_cols_ext, const int64_t *rowptr,
                     const int32_t *col, const double *val, rails_csr **out);
/* An operator given by its action instead of a CSR block (composite operators such as the reference's Schur complement
 * A22 - A21 A11^-1 A12, src/SchurOperator.cpp:181-214; matrix-free operators): rails_spmm on the returned handle calls
 * fn(user, trans, X, xc0, nc, Y, yc0), which must set Y[:, yc0:yc0+nc] = op(A) X[:, xc0:xc0+nc] with the C-ABI panel functions on
 * the context's stream and return 0.  The handle takes the Matrix role everywhere a CSR handle does (single GPU). */
typedef int (*rails_apply_fn)(void *user, int trans, const rails_panel *X, int xc0, int nc, rails_panel *Y, int yc0);
int rails_csr_create_callback(rails_ctx *ctx, int64_t m_local, rails_apply_fn fn, void *user, rails_csr **out);
int rails_csr_destroy(rails_csr *A);
int64_t rails_csr_rows(const rails_csr *A);
int64_t rails_csr_nnz(const rails_csr *A);

/* Ghost-row plan for multi-GPU: send_rows[0..n_send) are LOCAL row indices to pack (in
 * the order the halo hook expects), n_ghost rows are received. */
int rails_csr_set_halo(rails_csr *A, int64_t n_send, const int64_t *send_rows, int64_t n_ghost,
                       rails_halo_fn fn, void *user);

/* Rows per neighbour rank of that plan: send_rows is grouped by destination rank (ranks ascending, send_counts[r] rows to rank r) and
 * the ghost rows by owning rank (recv_counts[r] from rank r).  With these and an RCCL communicator on the context the exchange
 * is one group of ncclSend / ncclRecv per product and rails_csr_set_halo may be called with fn = NULL. */
int rails_csr_set_halo_counts(rails_csr *A, int nranks, const int64_t *send_counts, const int64_t *recv_counts);

/* Y[:, yc0:yc0+nc] = op(A) * X[:, xc0:xc0+nc]; trans != 0 applies A^T (single GPU only).
 * Replaces `A_ * W` (src/LyapunovSolver.hpp:146).  X and Y must not alias. */
int rails_spmm(rails_ctx *ctx, rails_csr *A, int trans, const rails_panel *X, int xc0, int nc,
               rails_panel *Y, int yc0);

/* Kernel variant control for benchmarking and tests: 0 = auto, 1 = row-gather kernel (column chunking by the window
 * heuristic), 2 = LDS-staged footprint kernel, 3 = row-gather over the whole width, 4 / 5 = row-gather in 32 / 64 column
 * chunks inside one launch, 6 = LDS-staged footprint kernel with 16-column chunks (whole 128-B lines per staged row),
 * 7 = sweep kernel (banded patterns: X streamed once per XCD through LDS rings, partial sums in registers; fails when the
 * pattern or the column count does not fit), 8 = never the sweep kernel (auto otherwise). */
int rails_csr_set_variant(rails_csr *A, int variant);
/* Set-up for products of nc columns with A (trans != 0: with its transpose): builds now what the automatic kernel choice would
 * otherwise put off -- the sweep kernel's schedule (banded patterns, 64 to 256 columns) costs about a third of a second of host time per
 * million rows, a few hundred times what it saves one product, so without this call an operator only builds it after it has been asked for
 * RAILS_SWEEP_AFTER (default 16) products of that width.  *kernel_ready (may be null) = 1 when the sweep kernel will take them.  A
 * no-op for operators and widths the kernel does not apply to.  The reference has no counterpart: Epetra's FillComplete is the
 * nearest (set-up work of the matrix class before `A_ * W`, src/LyapunovSolver.hpp:146). */
int rails_csr_prepare(rails_ctx *ctx, rails_csr *A, int trans, int nc, int *kernel_ready);
/* Statistics of the sweep kernel's schedule for nc columns, once it has been built:
 * out[0] slot efficiency, [1] X rows staged per matrix row and chunk, [2] lock-step trips, [3] 1 if the schedule exists. */
int rails_csr_sweep_stats(rails_csr *A, int nc, double *out);
/* A rectangular operator, n_rows x n_cols, all columns local: in rails_spmm X has n_cols rows and Y n_rows; no transposed apply, no
 * ghost rows.  Role: the off-diagonal blocks A12, A21 of the reference's Schur complement operator (src/SchurOperator.cpp:181-214). */
int rails_csr_create_rect(rails_ctx *ctx, int64_t n_rows, int64_t n_cols, const int64_t *rowptr, const int32_t *col, const double *val,
                          rails_csr **out);
/* name of the kernel the last rails_spmm on A launched */
const char *rails_csr_last_kernel(const rails_csr *A);

/* Host-side schedule of the sweep kernel (rails_amd/csrc/sweep_plan.h), exposed for tests and diagnostics: no device is
 * touched.  params = {waves, groups, rows per step, ring segments, parts, phases, segments being filled at any time, trips per
 * schedule entry (4 or 2; 0 = the library's choice)} (8 entries) or NULL for the kernel's own geometry.
 * info (iinfo has room for 16): iinfo[0..5] = params, [6] entries per step record, [7] lock-step trips, [8] nnz, [9] batches,
 * [10] entries in the busiest (wave, step), [11] slots per wave, [12] segments being filled, [13] trips per entry; dinfo[0] = slot efficiency
 * nnz / (slots x trips), dinfo[1] = X rows staged per matrix row and column chunk.  rails_sweep_plan_array lends the
 * arrays of the plan (which = 0 part_row0 i64, 1 sweep0 i64, 2 nsteps i32, 3 hdr_off i64, 4 batch_off i64, 5 flush_off i64,
 * 6 codes u32, 7 vals f64, 8 offs u16, 9 flush_rows i32); they live until rails_sweep_plan_destroy. */
/* ---- sparse triangular solves on device panels (rails_amd/csrc/sptrsv.hip) -----------------------------------------------------------
 * The A11 systems of the Schur-complement operator: the reference factorises A11 with Amesos KLU and solves on the host inside every
 * product (src/SchurOperator.cpp:171-176 set-up, :181-214 Apply); here a host factorisation's L and U are applied on the device by level
 * scheduling, so that the blocks of a product stay where the SpMM kernels left them.  A triangle comes in CSR (columns of a row in any
 * order); `lower` != 0: entries on or below the diagonal; `unit_diag` != 0: ones on the diagonal, not stored. */
typedef struct rails_sptrsv rails_sptrsv;
int rails_sptrsv_create(rails_ctx *ctx, int64_t n, const int64_t *rowptr, const int32_t *col, const double *val, int lower, int unit_diag,
                        rails_sptrsv **out);
void rails_sptrsv_destroy(rails_sptrsv *T);
int64_t rails_sptrsv_levels(const rails_sptrsv *T); /* number of levels of the dependency graph (diagnostics) */
/* X[:, c0:c0+nc] <- T^-1 X[:, c0:c0+nc] in place, queued on the context's stream */
int rails_sptrsv_solve(rails_ctx *ctx, const rails_sptrsv *T, rails_panel *X, int c0, int nc);
/* row permutations of a factorisation: Y row i <- X row perm[i] (scatter == 0) or Y row perm[i] <- X row i (scatter != 0); perm lives in
 * device memory (rails_index_upload / rails_index_free) */
int rails_panel_permute_rows(rails_ctx *ctx, const rails_panel *X, int xc0, int nc, const int32_t *perm_dev, int scatter, rails_panel *Y, int yc0);
int rails_index_upload(rails_ctx *ctx, const int32_t *host, int64_t n, int32_t **out_dev);
void rails_index_free(rails_ctx *ctx, int32_t *dev);

typedef struct rails_sweep_plan rails_sweep_plan;
int rails_sweep_plan_create(int64_t m, int64_t ncols, const int64_t *rowptr, const int32_t *col, const double *val,
                            const int *params, rails_sweep_plan **out);
int rails_sweep_plan_destroy(rails_sweep_plan *plan);
int rails_sweep_plan_info(const rails_sweep_plan *plan, int64_t *iinfo, double *dinfo);
int rails_sweep_plan_array(const rails_sweep_plan *plan, int which, const void **ptr, int64_t *count);

/* ------------------------------------------------------------------ panels --- */

/* A panel holds m_local rows x capacity columns (ld >= capacity, padded).
 * Role: the MultiVector template parameter's storage (StlWrapper m_max x n_max buffer,
 * src/StlWrapper.hpp:13-21). */
int rails_panel_create(rails_ctx *ctx, int64_t m_local, int capacity, rails_panel **out);
int rails_panel_destroy(rails_panel *P);
int64_t rails_panel_rows(const rails_panel *P);
int rails_panel_capacity(const rails_panel *P);
int rails_panel_ld(const rails_panel *P);
void *rails_panel_device_ptr(const rails_panel *P);
/* grow capacity preserving contents (StlWrapper::resize re-allocation, src/StlWrapper.cpp:238-248) */
int rails_panel_reserve(rails_ctx *ctx, rails_panel *P, int capacity);

/* host (column-major, ldh) <-> device columns [c0, c0+nc) */
int rails_panel_upload(rails_ctx *ctx, rails_panel *P, int c0, int nc, const double *host, int64_t ldh);
int rails_panel_download(rails_ctx *ctx, const rails_panel *P, int c0, int nc, double *host, int64_t ldh);

int rails_panel_fill(rails_ctx *ctx, rails_panel *P, int c0, int nc, double value);     /* operator=(double) :123 */
int rails_panel_scale(rails_ctx *ctx, rails_panel *P, int c0, int nc, double s);        /* operator*=  :131      */
int rails_panel_copy(rails_ctx *ctx, const rails_panel *X, int xc0, int nc, rails_panel *Y, int yc0); /* = / push_back :65,:367 */
int rails_panel_axpy(rails_ctx *ctx, double alpha, const rails_panel *X, int xc0, int nc, rails_panel *Y,
                     int yc0);                                                            /* += -=  :145-158        */
int rails_panel_random(rails_ctx *ctx, rails_panel *P, int c0, int nc);                  /* random() :414         */

/* C (a x b, host, column-major ldc) = X[:, xc0:+a]^T * Y[:, yc0:+b], summed over all ranks.
 * Replaces MultiVector::dot (src/StlWrapper.cpp:394-412; call sites
 * src/LyapunovSolver.hpp:173,187,394,400,406).  Synchronises. */
int rails_gram(rails_ctx *ctx, const rails_panel *X, int xc0, int a, const rails_panel *Y, int yc0, int b,
               double *C_host, int ldc);

/* Y[:, yc0:+r] = beta * Y[:, yc0:+r] + alpha * X[:, xc0:+k] * C   (C host, k x r column-major).
 * Replaces MultiVector * DenseMatrix (src/StlWrapper.cpp:168-187; call sites
 * src/LyapunovSolver.hpp:265,290,396,402,443).  X and Y may be the same panel only if the
 * column windows coincide exactly (in-place, row-local) or are disjoint. */
int rails_panel_gemm(rails_ctx *ctx, double alpha, const rails_panel *X, int xc0, int k, const double *C_host,
                     int ldc, int r, double beta, rails_panel *Y, int yc0);

/* Opt-in since round 3 (the library's own one-pass MFMA kernel, k_panel_gemm_wide, is the default and the faster one: 49 against 43-46
 * TFLOP/s at k = 324, r = 268): with RAILS_WIDE_GEMM=rocblas in the environment this call creates a rocBLAS handle (resolved with dlopen;
 * ~0.3 s in a warm process, seconds in a cold one, once per process, for the calling context's device, on the CALLING thread) and
 * rails_panel_gemm_wide with k, r >= 64 goes through rocblas_dgemm from then on (rails_ctx_library_gemm_ready).  Without the variable
 * the call does nothing. */
int rails_ctx_enable_library_gemm(rails_ctx *ctx);
int rails_ctx_library_gemm_ready(const rails_ctx *ctx);

/* Deferred small results: a chain Gram -> update -> Gram -> Cholesky -> update on the device without the host in between (the block
 * orthogonalisation of the coordinate-space back end behind the host's projected solve).  An arena of `nslots` slots of
 * `doubles_per_slot` doubles on the device with a pinned mirror; rails_gram_deferred leaves X'Y (a x b, leading dimension a, summed
 * over the ranks) in a slot and sends it on its way to the mirror; rails_panel_gemm_deferred takes the coefficient matrix of
 * Y = beta Y + alpha X C from a slot (rows [0, k) of its r columns, leading dimension ld); rails_chol_inverse_deferred turns the w x w
 * Gram matrix G of a slot (w <= 48) into D^-1 R^-1 (D = sqrt(diag G), R'R = D^-1 G D^-1) in another slot.  None of them synchronises;
 * rails_deferred_fetch copies from the mirror and is valid after a rails_ctx_sync that follows the call which filled the slot. */
int rails_deferred_reserve(rails_ctx *ctx, int nslots, int64_t doubles_per_slot);
int rails_gram_deferred(rails_ctx *ctx, const rails_panel *X, int xc0, int a, const rails_panel *Y, int yc0, int b, int slot);
int rails_panel_gemm_deferred(rails_ctx *ctx, double alpha, const rails_panel *X, int xc0, int k, int slot, int ld, int r, double beta,
                              rails_panel *Y, int yc0);
int rails_chol_inverse_deferred(rails_ctx *ctx, int slot_in, int w, int slot_out);
/* First update and second projection of a block in one pass over the basis (the second sweep of the reference's block
 * Gram-Schmidt, src/StlWrapper.cpp:314-344): Y[:, yc0:yc0+r] += alpha X[:, xc0:xc0+k] C with C on the host (k x r, leading dimension
 * ldc), then slot <- X[:, xc0:xc0+k]' Y[:, yc0:yc0+r2] for the leading r2 <= r updated columns (k x r2, leading dimension k, summed
 * over the ranks, on its way to the mirror).  Fused for r <= 17, r2 <= 16, 32 <= k <= 512; the two separate kernels otherwise. */
int rails_update_gram_deferred(rails_ctx *ctx, double alpha, const rails_panel *X, int xc0, int k, const double *C_host, int ldc, int r,
                               rails_panel *Y, int yc0, int r2, int slot);
int rails_deferred_fetch(rails_ctx *ctx, int slot, int64_t n, double *host_out);

/* The same product for any number r of output columns: one upload of C, launches in slices of 128 columns with no host wait in
 * between (the basis rotation of the coordinate-space back end).  X and Y: different panels or disjoint windows. */
int rails_panel_gemm_wide(rails_ctx *ctx, double alpha, const rails_panel *X, int xc0, int k, const double *C_host, int ldc, int r,
                          double beta, rails_panel *Y, int yc0);

/* Orthonormalise columns [k_old, k_old+w) of V against columns [0, k_old) and among themselves,
 * equivalent to the reference's column-wise CGS2 with pre/post normalisation
 * (src/StlWrapper.cpp:305-321): block CGS2 + CholQR2 on MFMA, falling back to the column-wise
 * form when the block Gram matrix is numerically rank deficient.  method: 0 = auto,
 * 1 = force column-wise, 2 = force block.  *used (may be NULL) receives the method used (1 column-wise, 2 block,
 * 3 block after one repair round: dependent columns replaced by their normalised residuals, see orth.hip). */
int rails_orthogonalize(rails_ctx *ctx, rails_panel *V, int k_old, int w, int method, int *used);

/* Fused residual Lanczos (src/LyapunovSolver.hpp:367-447): L steps of Lanczos on the implicit
 *   R = AV*T*MV^T + MV*T*AV^T + B*B^T      (MV == V for M = I)
 * started from a fresh random unit vector (one RNG stream is consumed, as Q.random() does at :374).
 * One pass over [AV MV B] per step; alpha, beta and the breakdown test stay on the device.
 *   T_host: k x k column-major (ldt).  H_host: out, (L+1) x (L+1) column-major (ldh >= L+1),
 *   zero-filled then the tridiagonal entries set exactly where the reference sets them.
 *   *steps: Lanczos steps done (< L on breakdown, beta < 1e-14, :419-426).
 * The orthonormal Lanczos vectors q_0..q_{steps-1} stay in a device side buffer of the library until the
 * next call on the same context.  Column windows must start at even columns; k <= 512, p <= 128.  Synchronises. */
int rails_resid_lanczos(rails_ctx *ctx, const rails_panel *AV, int avc0, const rails_panel *MV, int mvc0, int k,
                        const double *T_host, int ldt, const rails_panel *B, int bc0, int p, int L,
                        double *H_host, int ldh, int *steps);

/* Start of a residual Lanczos run only: draws the random start vector q0 (one RNG stream, as Q.random() at :374), and
 * returns sums_host = [AV^T q0 (k) | MV^T q0 (k) | B^T q0 (p) | q0^T q0] (summed over the ranks) in ONE pass over the
 * panels.  q0 (un-normalised) becomes Lanczos vector 0 for rails_lanczos_vectors.  Used by the projected-space Lanczos
 * of rails/HipSolverOps.hpp, which carries the recurrence itself in the (2k+p+1)-dimensional coefficient space. */
int rails_lanczos_start(rails_ctx *ctx, const rails_panel *AV, int avc0, const rails_panel *MV, int mvc0, int k,
                        const rails_panel *B, int bc0, int p, double *sums_host);

/* Out[:, oc0:oc0+w] = Q * S with Q the Lanczos vectors of the last rails_resid_lanczos and S (steps x w,
 * host column-major, lds).  Replaces `eigenvectors = Q * v` (src/LyapunovSolver.hpp:443); passing only the
 * selected columns of v writes the expansion vectors straight into V's tail (:338-339). */
int rails_lanczos_vectors(rails_ctx *ctx, const double *S_host, int lds, int w, rails_panel *Out, int oc0);
int rails_lanczos_release(rails_ctx *ctx); /* frees the Lanczos vectors kept by the context (also done by rails_ctx_destroy) */

/* ---------------------------------------------------------- timing helpers --- */
/* HIP-event timing on the context's stream (bench.py's roofline leg). */
/* Make room for `bytes` of coefficient uploads / small results now (pinned staging buffer and its device counterpart) instead
 * of at the call that first needs it. */
int rails_ctx_reserve_staging(rails_ctx *ctx, size_t bytes);
int rails_timer_start(rails_ctx *ctx);
int rails_timer_stop(rails_ctx *ctx, double *ms);

/* ------------------------------------------------------ host numerics (C ABI) --- */
/* Kept on the host per the north star; thin C ABI with the argument meaning of the reference's
 * shims.  LAPACK is resolved at run time (env RAILS_LAPACK_LIB, scipy's OpenBLAS, MKL, system). */

/* src/SlicotWrapper.hpp:14-16.  Only (dico,job,fact) = ('C','X','N') is supported (the only use,
 * src/LyapunovSolver.hpp:357).  trans='T' solves A*X + X*A^T = scale*C, trans='N' A^T*X + X*A.
 * SLICOT itself is not available: Bartels-Stewart on dgees + dtrsyl; info = n+1 when the
 * triangular solve had to perturb (SLICOT's convention).  Two faster routes to the same X are tried first
 * where they apply, each checked before it is accepted: the eigen-decomposition for a symmetric A, and a
 * squared Smith iteration (level-3 BLAS, residual-verified to the level Bartels-Stewart reaches) for a
 * nonsymmetric A whose spectrum is clustered.  A holds its Schur form afterwards only
 * on the Bartels-Stewart route. */
void rails_sb03md(char dico, char job, char fact, char trans, int n, double *A, int lda, double *X, int ldx,
                  double *scale, int *info);
/* how many calls of this process took the squared-Smith route and the Bartels-Stewart route (diagnostics) */
void rails_sb03md_counts(long *smith, long *schur);
/* calls of the factored ADI form that built on the call before (a bordered extension of its matrix: shifts kept, LU factors extended)
 * and calls that started from scratch */
void rails_sb03md_adi_counts(long *extended, long *fresh);
/* after an attempt of the squared-Smith route that did not apply, the calling thread skips the attempt for the next 30 calls;
 * this sets that counter (0: try again at the next call) */
void rails_sb03md_set_pause(int calls);
/* src/LapackWrapper.hpp:21-22 */
void rails_dsyev(char jobz, char uplo, int n, double *a, int lda, double *w, int *info);
/* src/LapackWrapper.hpp:18-19 */
void rails_dsteqr(char compz, int n, double *d, double *e, double *z, int ldz, double *work, int *info);
/* src/BlasWrapper.hpp:34-42 (DGEMM), for the small host products of the restart (X' (VAV X), src/LyapunovSolver.hpp:286) */
void rails_dgemm(char transa, char transb, int m, int n, int k, double alpha, const double *A, int lda, const double *B,
                 int ldb, double beta, double *C, int ldc);
/* Triangular solve with many right-hand sides (BLAS DTRSM; src/BlasWrapper.hpp has no counterpart -- the generalized projected
 * solve, matlab/mex/lyap.c:125-133, reduces with the Cholesky factor of V'MV through it) */
void rails_dtrsm(char side, char uplo, char transa, char diag, int m, int n, double alpha, const double *A, int lda, double *B, int ldb);
/* Cholesky (generalized projected solve, block orthogonalisation) */
void rails_dpotrf(char uplo, int n, double *a, int lda, int *info);
/* Cholesky with complete pivoting of a positive semi-definite matrix (LAPACK dpstrf): P'AP = R'R, stops at the first pivot
 * <= tol; *rank = pivots taken, piv 0-based (column j of the factor belongs to column piv[j] of A); info 1 = rank < n. */
void rails_dpstrf(char uplo, int n, double *a, int lda, int *piv, int *rank, double tol, int *info);
/* Orthonormal basis of the column space of A (m x n, overwritten): pivoted Householder QR (dgeqp3 + dorgqr), rank = pivots with
 * |r_ii| > tol*|r_11|; q (m x rank, ldq >= m).  info = -100 when the host LAPACK lacks dgeqp3/dorgqr. */
void rails_range_basis(int m, int n, double *a, int lda, double tol, double *q, int ldq, int *rank, int *info);
int rails_host_lapack_init(const char *path);
const char *rails_host_lapack_path(void);

#ifdef __cplusplus
}
#endif
#endif /* RAILS_HIP_H */
