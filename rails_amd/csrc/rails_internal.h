// Internal declarations shared by the HIP translation units of librails_hip.so.
#ifndef RAILS_INTERNAL_H
#define RAILS_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "rails_hip.h"

void rails_set_error(const char *fmt, ...);

#define RAILS_HIP_CHECK(expr)                                                                      \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            rails_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return RAILS_EHIP;                                                                     \
        }                                                                                          \
    } while (0)

#define RAILS_REQUIRE(cond, ...)                                                                   \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            rails_set_error(__VA_ARGS__);                                                          \
            return RAILS_EINVAL;                                                                   \
        }                                                                                          \
    } while (0)

#define RAILS_TRY(expr)                                                                            \
    do {                                                                                           \
        int rc__ = (expr);                                                                         \
        if (rc__ != RAILS_OK) return rc__;                                                         \
    } while (0)

// pad the leading dimension to a multiple of 16 doubles (128 B): every row starts on a cache line
static inline int rails_pad_ld(int capacity) { return ((capacity < 1 ? 1 : capacity) + 15) / 16 * 16; }

struct rails_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream + fork / join events: work that may run beside the first stream's (the interior rows of a row-partitioned product
    // while the ghost rows travel); made on first use (rails_ctx_second_stream)
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int num_cu = 256;
    // RNG
    uint64_t seed = 1;
    uint64_t next_stream = 0;
    // partition
    int rank = 0, nranks = 1;
    int64_t row0 = 0, m_global = -1;
    rails_allreduce_fn allreduce = nullptr;
    void *allreduce_user = nullptr;
    // RCCL communicator over the ranks of the partition (rccl_comm.hip); used when no hook is installed
    void *rccl = nullptr;
    bool own_rccl = false;
    int rccl_nranks = 0, rccl_rank = 0;
    // device workspace for reduction partials / small matrices
    double *ws = nullptr;
    size_t ws_bytes = 0;
    // device buffers of destroyed panels, kept for the next panel of the same size (stream-ordered re-use: no synchronisation,
    // no allocator call inside a solve that creates and drops temporaries of a few recurring sizes)
    std::vector<std::pair<size_t, double *>> free_panels;
    size_t free_panel_bytes = 0;
    // second device scratch (small matrices uploaded per call)
    double *small = nullptr;
    size_t small_bytes = 0;
    // pinned host staging
    double *pinned = nullptr;
    size_t pinned_bytes = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_h2d = nullptr; // recorded after an asynchronous upload out of `pinned`; waited for before the host writes there again
    bool h2d_pending = false;
    // deferred small results (dense.hip: rails_gram_deferred ...): slots of `defer_slot` doubles on the device with a pinned mirror
    double *defer_dev = nullptr, *defer_pin = nullptr;
    size_t defer_slot = 0;
    int defer_nslots = 0;
    // busy meter (rails_ctx_set_meter): a pair of events around every launch; the pairs are read at the next synchronisation of the stream
    bool meter = false;
    std::vector<hipEvent_t> meter_events; // 2 x RAILS_METER_PAIRS
    int meter_open = 0;                   // pairs recorded since the last synchronisation
    long meter_missed = 0;                // launches that found no free pair
    double gpu_busy_ms = 0.0;
    // counters (rails_ctx_stats)
    void *lz = nullptr; // rails_lanczos_state (lanczos.hip), released by rails_lanczos_release
    long n_orth_block = 0, n_orth_columnwise = 0, n_spmm_tiled = 0, n_spmm_planes = 0, n_spmm_overlapped = 0, n_spmm_sweep = 0, n_spmm_rowgather = 0, n_spmm_callback = 0, n_dev_alloc = 0, n_allreduce = 0, n_lanczos = 0, n_lanczos_start = 0, n_orth_repair = 0, n_update_gram_fused = 0;
};

// the busy meter (see rails_ctx): RAILS_LAUNCH brackets a launch, rails_stream_sync reads what has been bracketed since the last one
constexpr int RAILS_METER_PAIRS = 512;
inline bool rails_meter_before(rails_ctx *c)
{
    if (!c->meter) return false;
    if (c->meter_open >= RAILS_METER_PAIRS) {
        c->meter_missed++;
        return false;
    }
    hipEventRecord(c->meter_events[2 * c->meter_open], c->stream);
    return true;
}
inline void rails_meter_after(rails_ctx *c) { hipEventRecord(c->meter_events[2 * c->meter_open++ + 1], c->stream); }
inline hipError_t rails_stream_sync(rails_ctx *c)
{
    const hipError_t e = hipStreamSynchronize(c->stream);
    for (int i = 0; i < c->meter_open; ++i) {
        float ms = 0.0f;
        if (e == hipSuccess && hipEventElapsedTime(&ms, c->meter_events[2 * i], c->meter_events[2 * i + 1]) == hipSuccess) c->gpu_busy_ms += ms;
    }
    c->meter_open = 0;
    return e;
}
#define RAILS_LAUNCH(...)                                                                                                                  \
    do {                                                                                                                                   \
        const bool rails_metered = rails_meter_before(c);                                                                                  \
        hipLaunchKernelGGL(__VA_ARGS__);                                                                                                   \
        if (rails_metered) rails_meter_after(c);                                                                                           \
    } while (0)

struct rails_panel {
    rails_ctx *ctx = nullptr;
    double *d = nullptr;
    int64_t m = 0; // local rows (+ ghost rows for operator inputs are held separately)
    int cap = 0;
    int ld = 0;
};

struct rails_planes_plan; // spmm_planes.hip: coefficient records of a structured-grid stencil
struct rails_sweep_cache; // spmm_sweep.hip: device copies of the sweep kernel's schedules, one per column-chunk count

struct rails_csr {
    rails_ctx *ctx = nullptr;
    rails_sweep_cache *sweep = nullptr;
    rails_planes_plan *planes = nullptr;
    int64_t m = 0, ncols_ext = 0, nnz = 0;
    bool rect = false; // rails_csr_create_rect: n_rows x n_cols with n_cols != n_rows, all columns local (X has ncols_ext rows, Y has m)
    int64_t *rowptr = nullptr;
    int32_t *col = nullptr;
    double *val = nullptr;
    int max_row_nnz = 0;
    int64_t window_rows = 0; // mean (max col - min col + 1) over a sample of rows: the sliding window of X rows a row block gathers from
    // transposed copy, built lazily
    rails_csr *AT = nullptr;
    // host copy kept for building the transpose / tiling analysis
    std::vector<int64_t> h_rowptr;
    std::vector<int32_t> h_col;
    std::vector<double> h_val;
    // halo
    int64_t n_send = 0, n_ghost = 0;
    int64_t int_lo = 0, int_hi = 0; // interior rows [int_lo, int_hi): no ghost columns (rails_csr_set_halo)
    int64_t window_rows_int = 0;    // window_rows of the interior rows alone
    int64_t *send_rows = nullptr;
    double *send_buf = nullptr;
    double *ext = nullptr; // [m + n_ghost] x ld_ext staging of X with ghosts appended
    size_t send_cap = 0, ext_cap = 0;
    rails_halo_fn halo = nullptr;
    void *halo_user = nullptr;
    std::vector<int64_t> send_counts, recv_counts; // rows per neighbour rank (rails_csr_set_halo_counts): the RCCL form of the exchange
    // LDS-staged footprint kernel (spmm.hip): per row-block column footprints
    int variant = 0;
    bool tiled_ready = false;
    int is_grid = -1; // structured-grid stencil? (-1: not looked at yet; rails_csr_is_grid, spmm.hip)
    bool tiled_ok = false;
    int tile_rows = 0;
    int64_t n_tiles = 0;
    int32_t *t_fp_ptr = nullptr; // [n_tiles+1] offsets into t_fp
    int32_t *t_fp = nullptr;     // footprint column lists
    uint16_t *t_lcol = nullptr;  // [nnz] footprint-relative column of every nonzero, tile-major
    int32_t *t_rowptr = nullptr; // [n_tiles+1] offsets into t_rows
    int32_t *t_rows = nullptr;   // [m] rows of every tile
    int64_t *t_nzptr = nullptr;  // [n_tiles+1] offsets into t_val / t_lcol
    int32_t *t_rp = nullptr;     // [m + n_tiles] per-tile local row offsets (rows+1 per tile)
    double *t_val = nullptr;     // [nnz] values, tile-major
    uint16_t *t_fpos = nullptr;  // LDS row of every footprint entry
    int max_fp = 0, max_nz = 0, max_pos = 0;
    double tile_reuse = 0.0;
    bool tile_grid = false;
    const char *last_kernel = "";
    // operator given by its action (rails_csr_create_callback): rails_spmm hands the panels over
    rails_apply_fn apply_cb = nullptr;
    void *apply_user = nullptr;
};

// Diagnostics: with RAILS_TRACE_SLOW_MS=x every guarded entry point synchronises before and after its work and reports on stderr
// when it took longer than x ms (finds the one call behind a slow trip; off by default: no synchronisation, no cost).
struct rails_slow_guard {
    rails_ctx *c;
    const char *what;
    long long a, b;
    double t0 = 0.0;
    static double limit_ms()
    {
        static const double v = getenv("RAILS_TRACE_SLOW_MS") ? atof(getenv("RAILS_TRACE_SLOW_MS")) : 0.0;
        return v;
    }
    static double now()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    }
    rails_slow_guard(rails_ctx *ctx, const char *w, long long a_ = 0, long long b_ = 0) : c(ctx), what(w), a(a_), b(b_)
    {
        if (limit_ms() > 0.0 && c) {
            hipStreamSynchronize(c->stream);
            t0 = now();
        }
    }
    ~rails_slow_guard()
    {
        if (limit_ms() > 0.0 && c) {
            hipStreamSynchronize(c->stream);
            const double dt = now() - t0;
            if (dt > limit_ms()) fprintf(stderr, "[rails slow] %s(%lld, %lld): %.2f ms\n", what, a, b, dt);
        }
    }
};

// ---- helpers implemented in ctx.hip ----
int rails_ws_reserve(rails_ctx *ctx, size_t bytes);
void rails_library_gemm_release(rails_ctx *ctx);
int rails_small_reserve(rails_ctx *ctx, size_t bytes);
int rails_pinned_reserve(rails_ctx *ctx, size_t bytes);
// host -> pinned -> device staging without a stream synchronisation: begin_write waits until the previous upload out of the
// pinned buffer has been consumed (and reserves), end_write marks the upload just enqueued
int rails_pinned_begin_write(rails_ctx *ctx, size_t bytes);
int rails_pinned_end_write(rails_ctx *ctx);
int rails_allreduce_dev(rails_ctx *ctx, double *dev, size_t n);
int rails_ctx_second_stream(rails_ctx *ctx); // makes stream2 / ev_fork / ev_join when missing

// rccl_comm.hip
int rails_rccl_allreduce(rails_ctx *c, double *dev, size_t n);
int rails_rccl_halo(rails_ctx *c, const rails_csr *A, const double *send_buf, double *recv_buf, int ncols);
void rails_rccl_release(rails_ctx *c);

// ---- kernels / launchers across translation units ----
// spmm_sweep.hip: the sweep kernel for banded patterns; *done tells whether it computed the product
int rails_spmm_sweep(rails_ctx *c, rails_csr *A, const double *X, int ldx, const double *Xg, int ldg, double *Y, int ldy, int nc, bool aligned,
                     bool force, bool *done);
int rails_sweep_prepare(rails_ctx *c, rails_csr *A, int nc, bool *fits); // the schedule for nc columns, now
void rails_sweep_release(rails_csr *A);
// the interior rows of a row-partitioned operator as a sweep of their own (on stream st, beside the halo exchange)
int rails_spmm_sweep_interior(rails_ctx *c, rails_csr *A, const double *X, int ldx, double *Y, int ldy, int nc, bool aligned, hipStream_t st, bool *done);
int rails_sweep_prepare_interior(rails_ctx *c, rails_csr *A, int nc, bool *fits);
// spmm_planes.hip: the plane-sweep kernel for structured-grid stencils; *done tells whether it computed the product
int rails_spmm_planes(rails_ctx *c, rails_csr *A, const double *X, int ldx, double *Y, int ldy, int nc, bool aligned, bool build, bool *done);
void rails_planes_release(rails_csr *A);
// the interior planes of a z-slab of a grid stencil (rows without ghost columns) on stream st; *done = false: not that kind of operator
int rails_spmm_planes_interior(rails_ctx *c, rails_csr *A, const double *X, int ldx, double *Y, int ldy, int nc, bool aligned, hipStream_t st, bool *done);
bool planes_last_interior(const rails_csr *A);
bool rails_detect_grid(const rails_csr *A, int64_t *nx, int64_t *ny, int64_t *nz); // spmm.hip
// dense.hip: partial Gram into device memory (no host copy / all-reduce): C_dev (a x b col-major, ldc = a)
int rails_gram_dev(rails_ctx *ctx, const double *X, int ldx, const double *Y, int ldy, int64_t m, int a, int b,
                   double *C_dev);
int rails_panel_gemm_dev(rails_ctx *ctx, double alpha, const double *X, int ldx, int k, const double *C_dev,
                         int r, double beta, double *Y, int ldy, int64_t m);

#endif
