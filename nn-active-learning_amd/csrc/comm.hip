// RCCL leg of the sharded pool: ONE collective, the all-reduce(sum) of the L x L fp64 Fisher sum
// over xGMI.  The library is not linked against RCCL: the process already holds one (PyTorch-ROCm's librccl.so.1,
// loaded with torch before libalq - see _lib.py), and a second copy beside it would be a second runtime.  The
// entry points are resolved at the first use with dlopen("librccl.so.1"), which returns the loaded object.
#include <dlfcn.h>

#include <cstring>

#include "alq_internal.h"

namespace alq {

// the slice of rccl.h used here (NCCL ABI: stable enums / handle types)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclFloat64 = 8, kNcclSum = 0 };

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static Rccl g_rccl;

static int rccl_load() {
    if (g_rccl.h) return ALQ_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    ALQ_REQUIRE(h != nullptr, ALQ_EUNSUPPORTED, "RCCL is not loadable: %s", dlerror());
    Rccl r;
    r.h = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    ALQ_REQUIRE(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString, ALQ_EUNSUPPORTED,
                "librccl lacks an expected entry point");
    g_rccl = r;
    return ALQ_OK;
}

#define ALQ_RCCL(expr)                                                                         \
    do {                                                                                       \
        int r_ = (expr);                                                                       \
        if (r_ != kNcclSuccess) {                                                              \
            ::alq::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r_)); \
            return ALQ_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

}  // namespace alq

using namespace alq;

extern "C" {

int alq_comm_unique_id(void *h_id) {
    ALQ_REQUIRE(h_id != nullptr, ALQ_EINVAL, "alq_comm_unique_id: null buffer");
    ALQ_TRY(rccl_load());
    ncclUniqueId id;
    ALQ_RCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(h_id, id.internal, sizeof(id.internal));
    return ALQ_OK;
}

int alq_comm_init(alq_ctx *ctx, const void *h_id, int rank, int world) {
    ALQ_REQUIRE(ctx && h_id && world >= 1 && rank >= 0 && rank < world, ALQ_EINVAL, "alq_comm_init: bad argument");
    ALQ_REQUIRE(ctx->comm == nullptr, ALQ_EINVAL, "alq_comm_init: the context already has a communicator");
    ALQ_TRY(rccl_load());
    ALQ_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(id.internal, h_id, sizeof(id.internal));
    ncclComm_t c = nullptr;
    ALQ_RCCL(g_rccl.CommInitRank(&c, world, id, rank));
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return ALQ_OK;
}

int alq_comm_destroy(alq_ctx *ctx) {
    ALQ_REQUIRE(ctx != nullptr, ALQ_EINVAL, "null ctx");
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        ALQ_RCCL(g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(ctx->comm)));
        ctx->comm = nullptr;
    }
    return ALQ_OK;
}

int alq_allreduce_sum(alq_ctx *ctx, double *d_buf, int64_t count) {
    ALQ_REQUIRE(ctx && (count == 0 || d_buf) && count >= 0, ALQ_EINVAL, "alq_allreduce_sum: bad argument");
    ALQ_REQUIRE(ctx->comm != nullptr, ALQ_EINVAL, "alq_allreduce_sum: no communicator (alq_comm_init first)");
    if (count == 0) return ALQ_OK;
    ALQ_HIP(hipSetDevice(ctx->device));
    ALQ_RCCL(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, kNcclFloat64, kNcclSum, reinterpret_cast<ncclComm_t>(ctx->comm),
                              ctx->stream));
    return ALQ_OK;
}

}  // extern "C"
