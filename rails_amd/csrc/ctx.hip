// ctx.hip -- context, panels, BLAS-1 style panel kernels, host<->device transfer, RNG.
// Part of librails_hip.so (gfx950).  See include/rails_hip.h for the ABI contract.
#include "rails_internal.h"

static thread_local char g_err[1024] = "";

void rails_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *rails_last_error(void) { return g_err; }
extern "C" const char *rails_version(void) { return "rails_amd 0.1 (gfx950)"; }

// ------------------------------------------------------------------ context ---

extern "C" int rails_ctx_create(int device, void *stream, rails_ctx **out)
{
    RAILS_REQUIRE(out != nullptr, "rails_ctx_create: out is null");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        rails_set_error("rails_ctx_create: no HIP device visible (%s); the HIP path has no CPU fallback",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return RAILS_ENODEV;
    }
    RAILS_REQUIRE(device >= 0 && device < ndev, "rails_ctx_create: device %d out of range [0,%d)", device, ndev);
    RAILS_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    RAILS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rails_set_error("rails_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                        prop.gcnArchName);
        return RAILS_ENODEV;
    }
    rails_ctx *c = new rails_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    hipError_t he = hipSuccess;
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        c->own_stream = he == hipSuccess;
    }
    if (he == hipSuccess) he = hipEventCreate(&c->ev0);
    if (he == hipSuccess) he = hipEventCreate(&c->ev1);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming);
    if (he != hipSuccess) { // nothing of a half-made context is left behind
        rails_set_error("rails_ctx_create: stream / event creation failed: %s", hipGetErrorString(he));
        if (c->ev0) hipEventDestroy(c->ev0);
        if (c->ev1) hipEventDestroy(c->ev1);
        if (c->ev_h2d) hipEventDestroy(c->ev_h2d);
    for (hipEvent_t e : c->meter_events) hipEventDestroy(e);
        if (c->own_stream) hipStreamDestroy(c->stream);
        delete c;
        return RAILS_EHIP;
    }
    rails_host_lapack_init(nullptr); // load the host LAPACK now (not inside the first solve); its absence is reported by the calls that need it
    *out = c;
    return RAILS_OK;
}

extern "C" int rails_ctx_destroy(rails_ctx *c)
{
    if (!c) return RAILS_OK;
    hipSetDevice(c->device);
    rails_stream_sync(c);
    rails_lanczos_release(c);
    rails_rccl_release(c);
    rails_library_gemm_release(c);
    for (auto &fp : c->free_panels) hipFree(fp.second);
    c->free_panels.clear();
    if (c->ws) hipFree(c->ws);
    if (c->small) hipFree(c->small);
    if (c->pinned) hipHostFree(c->pinned);
    if (c->defer_dev) hipFree(c->defer_dev);
    if (c->defer_pin) hipHostFree(c->defer_pin);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev_h2d) hipEventDestroy(c->ev_h2d);
    for (hipEvent_t e : c->meter_events) hipEventDestroy(e);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return RAILS_OK;
}

int rails_ctx_second_stream(rails_ctx *c)
{
    if (c->stream2) return RAILS_OK;
    RAILS_HIP_CHECK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    RAILS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    RAILS_HIP_CHECK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    return RAILS_OK;
}

extern "C" int rails_ctx_sync(rails_ctx *c)
{
    RAILS_REQUIRE(c, "rails_ctx_sync: null context");
    RAILS_HIP_CHECK(rails_stream_sync(c));
    return RAILS_OK;
}

extern "C" int rails_ctx_stats(rails_ctx *c, char *buf, int cap)
{
    RAILS_REQUIRE(c && buf && cap > 0, "rails_ctx_stats: bad argument");
    int n = snprintf(buf, (size_t)cap,
                     "{\"orth_block\": %ld, \"orth_columnwise\": %ld, \"spmm_tiled\": %ld, \"spmm_planes\": %ld, \"spmm_halo_overlapped\": %ld, \"spmm_sweep\": %ld, \"spmm_rowgather\": %ld, \"spmm_callback\": %ld, \"device_allocations\": %ld, \"allreduce\": %ld, "
                     "\"lanczos\": %ld, \"lanczos_start\": %ld, \"orth_repair\": %ld, \"update_gram_fused\": %ld, \"gpu_busy_ms\": %.3f}",
                     c->n_orth_block, c->n_orth_columnwise, c->n_spmm_tiled, c->n_spmm_planes, c->n_spmm_overlapped, c->n_spmm_sweep, c->n_spmm_rowgather, c->n_spmm_callback, c->n_dev_alloc, c->n_allreduce, c->n_lanczos, c->n_lanczos_start, c->n_orth_repair, c->n_update_gram_fused, c->gpu_busy_ms);
    RAILS_REQUIRE(n > 0 && n < cap, "rails_ctx_stats: buffer too small");
    return RAILS_OK;
}

extern "C" void *rails_ctx_stream(rails_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int rails_ctx_set_meter(rails_ctx *c, int on)
{
    RAILS_REQUIRE(c, "rails_ctx_set_meter: null context");
    hipSetDevice(c->device);
    RAILS_HIP_CHECK(rails_stream_sync(c)); // what is pending is read (or, switching on, not bracketed: nothing to read)
    if (on && c->meter_events.empty()) {
        c->meter_events.resize(2 * RAILS_METER_PAIRS, nullptr);
        for (hipEvent_t &e : c->meter_events) RAILS_HIP_CHECK(hipEventCreate(&e));
    }
    c->meter = on != 0;
    return RAILS_OK;
}

extern "C" int rails_ctx_rng_state(rails_ctx *c, uint64_t *seed, uint64_t *next_stream)
{
    RAILS_REQUIRE(c && seed && next_stream, "rails_ctx_rng_state: null argument");
    *seed = c->seed;
    *next_stream = c->next_stream;
    return RAILS_OK;
}

extern "C" int rails_ctx_set_seed(rails_ctx *c, uint64_t seed, uint64_t first_stream)
{
    RAILS_REQUIRE(c, "null context");
    c->seed = seed;
    c->next_stream = first_stream;
    return RAILS_OK;
}

extern "C" int rails_ctx_set_partition(rails_ctx *c, int rank, int nranks, int64_t row0, int64_t m_global)
{
    RAILS_REQUIRE(c, "null context");
    RAILS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks && row0 >= 0, "rails_ctx_set_partition: bad rank %d/%d row0 %lld",
                  rank, nranks, (long long)row0);
    c->rank = rank;
    c->nranks = nranks;
    c->row0 = row0;
    c->m_global = m_global;
    return RAILS_OK;
}

extern "C" int rails_ctx_set_allreduce(rails_ctx *c, rails_allreduce_fn fn, void *user)
{
    RAILS_REQUIRE(c, "null context");
    c->allreduce = fn;
    c->allreduce_user = user;
    return RAILS_OK;
}

int rails_allreduce_dev(rails_ctx *c, double *dev, size_t n)
{
    if (n == 0) return RAILS_OK;
    if (c->nranks <= 1 && !c->allreduce && !c->rccl) return RAILS_OK; // a hook or communicator on a single rank is still honoured
    if (!c->allreduce && c->rccl) { // native: RCCL on the context's stream
        c->n_allreduce++;
        return rails_rccl_allreduce(c, dev, n);
    }
    if (!c->allreduce) {
        rails_set_error("row-partitioned run (nranks=%d) with neither an RCCL communicator (rails_ctx_init_rccl) nor an all-reduce hook", c->nranks);
        return RAILS_ECOMM;
    }
    c->n_allreduce++;
    int rc = c->allreduce(c->allreduce_user, dev, n, (void *)c->stream);
    if (rc != 0) {
        rails_set_error("all-reduce hook failed with code %d", rc);
        return RAILS_ECOMM;
    }
    return RAILS_OK;
}

static int grow(rails_ctx *c, double **p, size_t *have, size_t need, bool host)
{
    if (need <= *have) return RAILS_OK;
    RAILS_HIP_CHECK(rails_stream_sync(c));
    if (*p) {
        if (host)
            RAILS_HIP_CHECK(hipHostFree(*p));
        else
            RAILS_HIP_CHECK(hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    size_t sz = 2 * need + 4096; // doubling: these buffers follow the basis dimension, which creeps up restart by restart
    c->n_dev_alloc++;
    if (getenv("RAILS_TRACE_ALLOC")) fprintf(stderr, "rails alloc: %s buffer grows to %zu bytes\n", host ? "pinned" : "device", sz);
    hipError_t e = host ? hipHostMalloc((void **)p, sz, hipHostMallocDefault) : hipMalloc((void **)p, sz);
    if (e != hipSuccess) {
        rails_set_error("allocation of %zu bytes failed: %s", sz, hipGetErrorString(e));
        return RAILS_ENOMEM;
    }
    *have = sz;
    return RAILS_OK;
}

int rails_ws_reserve(rails_ctx *c, size_t bytes) { return grow(c, &c->ws, &c->ws_bytes, bytes, false); }
int rails_small_reserve(rails_ctx *c, size_t bytes) { return grow(c, &c->small, &c->small_bytes, bytes, false); }
int rails_pinned_reserve(rails_ctx *c, size_t bytes) { return grow(c, &c->pinned, &c->pinned_bytes, bytes, true); }

// room for coefficient uploads / small results of `bytes` in both the pinned staging buffer and its device counterpart, made now
// rather than in the middle of a solve (every growth synchronises the stream and calls the allocator)
extern "C" int rails_ctx_reserve_staging(rails_ctx *c, size_t bytes)
{
    RAILS_REQUIRE(c, "null context");
    RAILS_TRY(rails_small_reserve(c, bytes));
    return rails_pinned_reserve(c, bytes);
}

int rails_pinned_begin_write(rails_ctx *c, size_t bytes)
{
    if (c->h2d_pending) {
        RAILS_HIP_CHECK(hipEventSynchronize(c->ev_h2d));
        c->h2d_pending = false;
    }
    return rails_pinned_reserve(c, bytes);
}

int rails_pinned_end_write(rails_ctx *c)
{
    RAILS_HIP_CHECK(hipEventRecord(c->ev_h2d, c->stream));
    c->h2d_pending = true;
    return RAILS_OK;
}

extern "C" int rails_timer_start(rails_ctx *c)
{
    RAILS_REQUIRE(c, "null context");
    RAILS_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    return RAILS_OK;
}

extern "C" int rails_timer_stop(rails_ctx *c, double *ms)
{
    RAILS_REQUIRE(c && ms, "null argument");
    RAILS_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    RAILS_HIP_CHECK(hipEventSynchronize(c->ev1));
    float f = 0.f;
    RAILS_HIP_CHECK(hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = (double)f;
    return RAILS_OK;
}

// ------------------------------------------------------------------- panels ---

extern "C" int rails_panel_create(rails_ctx *c, int64_t m_local, int capacity, rails_panel **out)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    RAILS_REQUIRE(c && out, "rails_panel_create: null argument");
    RAILS_REQUIRE(m_local >= 0 && capacity >= 0, "rails_panel_create: bad shape %lld x %d", (long long)m_local, capacity);
    rails_panel *P = new rails_panel();
    P->ctx = c;
    P->m = m_local;
    P->cap = capacity;
    P->ld = rails_pad_ld(capacity);
    size_t bytes = (size_t)(m_local > 0 ? m_local : 1) * P->ld * sizeof(double);
    for (size_t q = 0; q < c->free_panels.size() && !P->d; ++q)
        if (c->free_panels[q].first == bytes) {
            P->d = c->free_panels[q].second;
            c->free_panel_bytes -= bytes;
            c->free_panels.erase(c->free_panels.begin() + (long)q);
        }
    if (!P->d) {
        c->n_dev_alloc++;
        if (getenv("RAILS_TRACE_ALLOC")) fprintf(stderr, "rails alloc: panel %lld x %d (%zu bytes)\n", (long long)m_local, capacity, bytes);
        hipError_t e = hipMalloc((void **)&P->d, bytes);
        if (e != hipSuccess && !c->free_panels.empty()) { // give the cached buffers back and try again
            rails_stream_sync(c);
            for (auto &fp : c->free_panels) hipFree(fp.second);
            c->free_panels.clear();
            c->free_panel_bytes = 0;
            e = hipMalloc((void **)&P->d, bytes);
        }
        if (e != hipSuccess) {
            rails_set_error("rails_panel_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            delete P;
            return RAILS_ENOMEM;
        }
    }
    // padding columns are read by vectorised kernels: keep them finite
    RAILS_HIP_CHECK(hipMemsetAsync(P->d, 0, bytes, c->stream));
    *out = P;
    return RAILS_OK;
}

extern "C" int rails_panel_destroy(rails_panel *P)
{
    if (!P) return RAILS_OK;
    if (P->d) {
        rails_ctx *c = P->ctx;
        const size_t bytes = (size_t)(P->m > 0 ? P->m : 1) * P->ld * sizeof(double);
        // keep up to 8 GiB of buffers for re-use (work queued on the stream may still read this one: the next user is on the same stream)
        if (bytes <= ((size_t)2 << 30) && c->free_panel_bytes + bytes <= ((size_t)8 << 30) && c->free_panels.size() < 64) {
            c->free_panels.push_back(std::make_pair(bytes, P->d));
            c->free_panel_bytes += bytes;
        } else {
            rails_stream_sync(c);
            hipFree(P->d);
        }
    }
    delete P;
    return RAILS_OK;
}

extern "C" int64_t rails_panel_rows(const rails_panel *P) { return P ? P->m : -1; }
extern "C" int rails_panel_capacity(const rails_panel *P) { return P ? P->cap : -1; }
extern "C" int rails_panel_ld(const rails_panel *P) { return P ? P->ld : -1; }
extern "C" void *rails_panel_device_ptr(const rails_panel *P) { return P ? (void *)P->d : nullptr; }

__global__ void k_copy2d(const double *__restrict__ X, int ldx, double *__restrict__ Y, int ldy, int64_t m, int nc)
{
    int64_t total = m * nc;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx / nc;
        int cidx = (int)(idx - r * nc);
        Y[r * ldy + cidx] = X[r * ldx + cidx];
    }
}

static inline int grid_for(rails_ctx *c, int64_t total, int block)
{
    int64_t g = (total + block - 1) / block;
    int64_t cap = (int64_t)c->num_cu * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int rails_panel_reserve(rails_ctx *c, rails_panel *P, int capacity)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    RAILS_REQUIRE(c && P, "null argument");
    if (capacity <= P->cap) return RAILS_OK;
    int nld = rails_pad_ld(capacity);
    if (nld == P->ld) {
        P->cap = capacity;
        return RAILS_OK;
    }
    double *nd = nullptr;
    size_t bytes = (size_t)(P->m > 0 ? P->m : 1) * nld * sizeof(double);
    c->n_dev_alloc++;
    if (getenv("RAILS_TRACE_ALLOC")) fprintf(stderr, "rails alloc: panel reserve %lld x %d -> %d (%zu bytes)\n", (long long)P->m, P->cap, capacity, bytes);
    hipError_t e = hipMalloc((void **)&nd, bytes);
    if (e != hipSuccess) {
        rails_set_error("rails_panel_reserve: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return RAILS_ENOMEM;
    }
    RAILS_HIP_CHECK(hipMemsetAsync(nd, 0, bytes, c->stream));
    if (P->m > 0 && P->cap > 0)
        RAILS_LAUNCH(k_copy2d, dim3(grid_for(c, P->m * P->cap, 256)), dim3(256), 0, c->stream, P->d, P->ld, nd, nld,
                           P->m, P->cap);
    RAILS_HIP_CHECK(rails_stream_sync(c));
    RAILS_HIP_CHECK(hipFree(P->d));
    P->d = nd;
    P->ld = nld;
    P->cap = capacity;
    return RAILS_OK;
}

static int check_window(const rails_panel *P, int c0, int nc, const char *what)
{
    RAILS_REQUIRE(P && P->d, "%s: null panel", what);
    RAILS_REQUIRE(c0 >= 0 && nc >= 0 && c0 + nc <= P->cap, "%s: columns [%d,%d) outside capacity %d", what, c0, c0 + nc,
                  P->cap);
    return RAILS_OK;
}

// tmp is column-major m x nc (contiguous); panel window is row-major
__global__ void k_scatter_cm_to_panel(const double *__restrict__ tmp, double *__restrict__ P, int ld, int64_t m, int nc)
{
    __shared__ double tile[32][33];
    // tile of 32 rows x 32 cols; grid.x over row tiles, grid.y over col tiles
    int64_t r0 = (int64_t)blockIdx.x * 32;
    int c0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 256 threads: ty in 0..7
    for (int j = ty; j < 32; j += 8) {
        int64_t r = r0 + tx;
        int cc = c0 + j;
        tile[j][tx] = (r < m && cc < nc) ? tmp[(int64_t)cc * m + r] : 0.0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int64_t r = r0 + i;
        int cc = c0 + tx;
        if (r < m && cc < nc) P[r * ld + cc] = tile[tx][i];
    }
}

__global__ void k_gather_panel_to_cm(const double *__restrict__ P, int ld, double *__restrict__ tmp, int64_t m, int nc)
{
    __shared__ double tile[32][33];
    int64_t r0 = (int64_t)blockIdx.x * 32;
    int c0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        int64_t r = r0 + i;
        int cc = c0 + tx;
        tile[i][tx] = (r < m && cc < nc) ? P[r * ld + cc] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int64_t r = r0 + tx;
        int cc = c0 + j;
        if (r < m && cc < nc) tmp[(int64_t)cc * m + r] = tile[tx][j];
    }
}

extern "C" int rails_panel_upload(rails_ctx *c, rails_panel *P, int c0, int nc, const double *host, int64_t ldh)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    RAILS_TRY(check_window(P, c0, nc, "rails_panel_upload"));
    RAILS_REQUIRE(host || nc == 0 || P->m == 0, "rails_panel_upload: null host buffer");
    RAILS_REQUIRE(ldh >= P->m, "rails_panel_upload: ldh %lld < rows %lld", (long long)ldh, (long long)P->m);
    if (nc == 0 || P->m == 0) return RAILS_OK;
    const int chunk = 32;
    RAILS_TRY(rails_ws_reserve(c, (size_t)P->m * chunk * sizeof(double)));
    for (int j0 = 0; j0 < nc; j0 += chunk) {
        int n = nc - j0 < chunk ? nc - j0 : chunk;
        RAILS_HIP_CHECK(hipMemcpy2DAsync(c->ws, (size_t)P->m * sizeof(double), host + (int64_t)j0 * ldh, (size_t)ldh * sizeof(double),
                                         (size_t)P->m * sizeof(double), n, hipMemcpyHostToDevice, c->stream));
        dim3 grid((unsigned)((P->m + 31) / 32), (unsigned)((n + 31) / 32));
        RAILS_LAUNCH(k_scatter_cm_to_panel, grid, dim3(256), 0, c->stream, c->ws, P->d + c0 + j0, P->ld, P->m, n);
        RAILS_HIP_CHECK(rails_stream_sync(c));
    }
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

extern "C" int rails_panel_download(rails_ctx *c, const rails_panel *P, int c0, int nc, double *host, int64_t ldh)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    RAILS_TRY(check_window(P, c0, nc, "rails_panel_download"));
    RAILS_REQUIRE(host || nc == 0 || P->m == 0, "rails_panel_download: null host buffer");
    RAILS_REQUIRE(ldh >= P->m, "rails_panel_download: ldh %lld < rows %lld", (long long)ldh, (long long)P->m);
    if (nc == 0 || P->m == 0) return RAILS_OK;
    const int chunk = 32;
    RAILS_TRY(rails_ws_reserve(c, (size_t)P->m * chunk * sizeof(double)));
    for (int j0 = 0; j0 < nc; j0 += chunk) {
        int n = nc - j0 < chunk ? nc - j0 : chunk;
        dim3 grid((unsigned)((P->m + 31) / 32), (unsigned)((n + 31) / 32));
        RAILS_LAUNCH(k_gather_panel_to_cm, grid, dim3(256), 0, c->stream, P->d + c0 + j0, P->ld, c->ws, P->m, n);
        RAILS_HIP_CHECK(hipMemcpy2DAsync(host + (int64_t)j0 * ldh, (size_t)ldh * sizeof(double), c->ws, (size_t)P->m * sizeof(double),
                                         (size_t)P->m * sizeof(double), n, hipMemcpyDeviceToHost, c->stream));
        RAILS_HIP_CHECK(rails_stream_sync(c));
    }
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

// --------------------------------------------------------- BLAS-1 on windows ---

__global__ void k_fill(double *__restrict__ P, int ld, int64_t m, int nc, double v)
{
    int64_t total = m * nc;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx / nc;
        int cidx = (int)(idx - r * nc);
        P[r * ld + cidx] = v;
    }
}

__global__ void k_scale(double *__restrict__ P, int ld, int64_t m, int nc, double s)
{
    int64_t total = m * nc;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx / nc;
        int cidx = (int)(idx - r * nc);
        P[r * ld + cidx] *= s;
    }
}

__global__ void k_axpy(double alpha, const double *__restrict__ X, int ldx, double *__restrict__ Y, int ldy, int64_t m, int nc)
{
    int64_t total = m * nc;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx / nc;
        int cidx = (int)(idx - r * nc);
        Y[r * ldy + cidx] += alpha * X[r * ldx + cidx];
    }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// value = f(seed, stream, global row, column) in (-1, 1); the same function is restated on the CPU side of the tests
__global__ void k_random(double *__restrict__ P, int ld, int64_t m, int nc, uint64_t seed, uint64_t stream, int64_t row0)
{
    uint64_t hs = splitmix64(seed ^ splitmix64(stream * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
    int64_t total = m * nc;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx / nc;
        int cidx = (int)(idx - r * nc);
        uint64_t h = splitmix64(hs ^ splitmix64((uint64_t)(row0 + r) * 0x9E3779B97F4A7C15ull + (uint64_t)cidx * 0xC2B2AE3D27D4EB4Full + 1));
        double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
        P[r * ld + cidx] = 2.0 * u - 1.0;
    }
}

extern "C" int rails_panel_fill(rails_ctx *c, rails_panel *P, int c0, int nc, double value)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    RAILS_TRY(check_window(P, c0, nc, "rails_panel_fill"));
    if (nc == 0 || P->m == 0) return RAILS_OK;
    RAILS_LAUNCH(k_fill, dim3(grid_for(c, P->m * nc, 256)), dim3(256), 0, c->stream, P->d + c0, P->ld, P->m, nc, value);
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

extern "C" int rails_panel_scale(rails_ctx *c, rails_panel *P, int c0, int nc, double s)
{
    RAILS_TRY(check_window(P, c0, nc, "rails_panel_scale"));
    if (nc == 0 || P->m == 0) return RAILS_OK;
    RAILS_LAUNCH(k_scale, dim3(grid_for(c, P->m * nc, 256)), dim3(256), 0, c->stream, P->d + c0, P->ld, P->m, nc, s);
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

extern "C" int rails_panel_copy(rails_ctx *c, const rails_panel *X, int xc0, int nc, rails_panel *Y, int yc0)
{
    RAILS_TRY(check_window(X, xc0, nc, "rails_panel_copy(X)"));
    RAILS_TRY(check_window(Y, yc0, nc, "rails_panel_copy(Y)"));
    RAILS_REQUIRE(X->m == Y->m, "rails_panel_copy: row mismatch %lld vs %lld", (long long)X->m, (long long)Y->m);
    if (nc == 0 || X->m == 0) return RAILS_OK;
    if (X->d == Y->d) {
        if (xc0 == yc0) return RAILS_OK;
        RAILS_REQUIRE(xc0 + nc <= yc0 || yc0 + nc <= xc0, "rails_panel_copy: overlapping windows of one panel");
    }
    RAILS_LAUNCH(k_copy2d, dim3(grid_for(c, X->m * nc, 256)), dim3(256), 0, c->stream, X->d + xc0, X->ld, Y->d + yc0,
                       Y->ld, X->m, nc);
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

extern "C" int rails_panel_axpy(rails_ctx *c, double alpha, const rails_panel *X, int xc0, int nc, rails_panel *Y, int yc0)
{
    RAILS_TRY(check_window(X, xc0, nc, "rails_panel_axpy(X)"));
    RAILS_TRY(check_window(Y, yc0, nc, "rails_panel_axpy(Y)"));
    RAILS_REQUIRE(X->m == Y->m, "rails_panel_axpy: row mismatch %lld vs %lld", (long long)X->m, (long long)Y->m);
    if (nc == 0 || X->m == 0) return RAILS_OK;
    RAILS_LAUNCH(k_axpy, dim3(grid_for(c, X->m * nc, 256)), dim3(256), 0, c->stream, alpha, X->d + xc0, X->ld,
                       Y->d + yc0, Y->ld, X->m, nc);
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}

extern "C" int rails_panel_random(rails_ctx *c, rails_panel *P, int c0, int nc)
{
    if (c) hipSetDevice(c->device); // allocations and launches go to the context's device whatever the caller's current device is
    RAILS_TRY(check_window(P, c0, nc, "rails_panel_random"));
    uint64_t s = c->next_stream++;
    if (nc == 0 || P->m == 0) return RAILS_OK;
    RAILS_LAUNCH(k_random, dim3(grid_for(c, P->m * nc, 256)), dim3(256), 0, c->stream, P->d + c0, P->ld, P->m, nc,
                       c->seed, s, c->row0);
    RAILS_HIP_CHECK(hipGetLastError());
    return RAILS_OK;
}
