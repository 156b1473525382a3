// rccl_comm.hip -- the two exchanges of the row-partitioned path done by the library itself over RCCL (xGMI inside a node):
//   * the sum all-reduce of the small projected blocks and Lanczos sums -- what Epetra hides inside Multiply('T','N') and Norm2
//     (src/Epetra_MultiVectorWrapper.cpp:238,312,431) -- as ncclAllReduce on the context's stream;
//   * the ghost rows of W that the local rows of A reference -- the import inside Epetra_CrsMatrix::Apply
//     (src/Epetra_OperatorWrapper.cpp:87) -- as one group of ncclSend / ncclRecv per product.
// With a communicator on the context the hooks (rails_ctx_set_allreduce, the halo hook of rails_csr_set_halo) are not needed: a
// C++ user of the drop-in classes gets the multi-GPU path without Python or torch.distributed in the loop.  The hooks stay as the
// fallback and take precedence when installed.  librccl is loaded at run time (dlopen): the library itself has no link-time
// dependency on it and single-GPU users never touch it.
#include "rails_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;
std::mutex g_rccl_mutex;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return RAILS_OK;
    const char *names[] = {getenv("RAILS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
    void *h = nullptr;
    for (int i = 0; i < 5 && !h; ++i)
        if (names[i] && *names[i]) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        rails_set_error("RCCL not found (librccl.so.1; RAILS_RCCL_LIB overrides): %s", dlerror());
        return RAILS_ECOMM;
    }
    RcclApi a;
    a.handle = h;
#define RAILS_RCCL_SYM(field, name)                                       \
    a.field = (decltype(a.field))dlsym(h, name);                          \
    if (!a.field) {                                                       \
        rails_set_error("RCCL library lacks %s", name);                   \
        dlclose(h);                                                       \
        return RAILS_ECOMM;                                               \
    }
    RAILS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    RAILS_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    RAILS_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RAILS_RCCL_SYM(CommCount, "ncclCommCount")
    RAILS_RCCL_SYM(CommUserRank, "ncclCommUserRank")
    RAILS_RCCL_SYM(AllReduce, "ncclAllReduce")
    RAILS_RCCL_SYM(Send, "ncclSend")
    RAILS_RCCL_SYM(Recv, "ncclRecv")
    RAILS_RCCL_SYM(GroupStart, "ncclGroupStart")
    RAILS_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    RAILS_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RAILS_RCCL_SYM
    g_rccl = a;
    return RAILS_OK;
}

#define RAILS_RCCL_CHECK(expr)                                                                                   \
    do {                                                                                                         \
        ncclResult_t r__ = (expr);                                                                               \
        if (r__ != ncclSuccess) {                                                                                \
            rails_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r__));      \
            return RAILS_ECOMM;                                                                                  \
        }                                                                                                        \
    } while (0)

} // namespace

static_assert(sizeof(ncclUniqueId) == RAILS_RCCL_ID_BYTES, "rails_hip.h declares the size of RCCL's unique id");

extern "C" int rails_rccl_unique_id(void *id_out)
{
    RAILS_REQUIRE(id_out, "rails_rccl_unique_id: null argument");
    RAILS_TRY(load_rccl());
    ncclUniqueId id;
    RAILS_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return RAILS_OK;
}

extern "C" int rails_ctx_init_rccl(rails_ctx *c, const void *id_in, int nranks, int rank)
{
    RAILS_REQUIRE(c && id_in && nranks >= 1 && rank >= 0 && rank < nranks, "rails_ctx_init_rccl: bad argument");
    RAILS_REQUIRE(!c->rccl, "rails_ctx_init_rccl: the context has a communicator already");
    RAILS_TRY(load_rccl());
    RAILS_HIP_CHECK(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ncclComm_t comm = nullptr;
    RAILS_RCCL_CHECK(g_rccl.CommInitRank(&comm, nranks, id, rank));
    c->rccl = (void *)comm;
    c->own_rccl = true;
    c->rccl_nranks = nranks;
    c->rccl_rank = rank;
    return RAILS_OK;
}

extern "C" int rails_ctx_set_rccl(rails_ctx *c, void *nccl_comm)
{
    RAILS_REQUIRE(c, "rails_ctx_set_rccl: null context");
    if (c->rccl && c->own_rccl) {
        RAILS_HIP_CHECK(rails_stream_sync(c));
        g_rccl.CommDestroy((ncclComm_t)c->rccl);
    }
    c->rccl = nullptr;
    c->own_rccl = false;
    if (!nccl_comm) return RAILS_OK;
    RAILS_TRY(load_rccl());
    int n = 0, r = 0;
    RAILS_RCCL_CHECK(g_rccl.CommCount((ncclComm_t)nccl_comm, &n));
    RAILS_RCCL_CHECK(g_rccl.CommUserRank((ncclComm_t)nccl_comm, &r));
    c->rccl = nccl_comm;
    c->rccl_nranks = n;
    c->rccl_rank = r;
    return RAILS_OK;
}

extern "C" int rails_ctx_rccl_size(const rails_ctx *c) { return c && c->rccl ? c->rccl_nranks : 0; }

void rails_rccl_release(rails_ctx *c)
{
    if (c->rccl && c->own_rccl && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c->rccl);
    c->rccl = nullptr;
}

// in place on a device buffer, ordered on the context's stream
int rails_rccl_allreduce(rails_ctx *c, double *dev, size_t n)
{
    RAILS_RCCL_CHECK(g_rccl.AllReduce(dev, dev, n, ncclDouble, ncclSum, (ncclComm_t)c->rccl, c->stream));
    return RAILS_OK;
}

// packed rows out, ghost rows in: one message per neighbour and direction, all in one group (no ordering between them)
int rails_rccl_halo(rails_ctx *c, const rails_csr *A, const double *send_buf, double *recv_buf, int ncols)
{
    RAILS_REQUIRE((int)A->send_counts.size() == c->rccl_nranks && (int)A->recv_counts.size() == c->rccl_nranks,
                  "rails_spmm: the ghost-row plan was made for %d ranks, the communicator has %d", (int)A->send_counts.size(), c->rccl_nranks);
    RAILS_RCCL_CHECK(g_rccl.GroupStart());
    int64_t so = 0, ro = 0;
    ncclResult_t first = ncclSuccess; // a failing call inside the group still has to be followed by GroupEnd: an open group swallows every later call
    for (int r = 0; r < c->rccl_nranks && first == ncclSuccess; ++r) {
        const int64_t ns = A->send_counts[r] * ncols, nr = A->recv_counts[r] * ncols;
        if (r != c->rccl_rank) {
            if (nr) first = g_rccl.Recv(recv_buf + ro, (size_t)nr, ncclDouble, r, (ncclComm_t)c->rccl, c->stream);
            if (ns && first == ncclSuccess) first = g_rccl.Send(send_buf + so, (size_t)ns, ncclDouble, r, (ncclComm_t)c->rccl, c->stream);
        }
        so += ns;
        ro += nr;
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (first != ncclSuccess || end != ncclSuccess) {
        rails_set_error("rails_spmm: the ghost-row exchange failed: %s", g_rccl.GetErrorString(first != ncclSuccess ? first : end));
        return RAILS_ECOMM;
    }
    return RAILS_OK;
}
