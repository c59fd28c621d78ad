// comm_api.hip -- the multi-GPU exchange of the navigation tick for a C host: RCCL called directly.
//
// The path shards without a data-path collective up to ONE step per tick:
// every rank has stepped its slab of the dense entity order and all ranks need all rows of
// [new position | velocity] before the next tick's neighbour queries -- an all-gather over xGMI.  The
// reference engine is C; it cannot reach torch.distributed.  These entry points give it the same
// exchange through librccl (ncclAllGather for equal slabs, one grouped ncclBroadcast per rank for ragged
// ones), plus the all-gather of baked flow tiles for hosts that let any agent sample any field.
//
// librccl is loaded on first use (dlopen): a single-GPU host never needs it.
#include "navhip_internal.h"

#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

#define HIPCHK(ctx, expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if(_e != hipSuccess) {                                                              \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);          \
            return NAVHIP_ERR_DEVICE;                                                       \
        }                                                                                   \
    } while(0)

namespace {

struct rccl_api {
    void *handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char  *(*GetErrorString)(ncclResult_t);
};

rccl_api g_rccl;
std::string g_rccl_error;

bool rccl_load()
{
    if(g_rccl.handle) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for(const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if(h) break; }
    if(!h) {
        const char *why = dlerror();               // (a second call would return NULL: the first clears it)
        g_rccl_error = std::string("dlopen(librccl): ") + (why ? why : "?");
        return false;
    }
    rccl_api a;
    a.handle = h;
#define SYM(field, name) *(void**)(&a.field) = dlsym(h, name); if(!a.field) { g_rccl_error = "librccl lacks " name; dlclose(h); return false; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather") SYM(Broadcast, "ncclBroadcast") SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl = a;
    return true;
}

}  // namespace

struct nh_comm {
    ncclComm_t comm;
    int        rank, world;
    float     *pack;   size_t pack_cap;       // [n][4] staging of navhip_comm_allgather_step_dev
};

#define NCCLCHK(ctx, expr)                                                                          \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if(_r != ncclSuccess) {                                                                     \
            (ctx)->last_error = std::string(#expr) + ": " + g_rccl.GetErrorString(_r);              \
            return NAVHIP_ERR_DEVICE;                                                               \
        }                                                                                           \
    } while(0)

static_assert(sizeof(ncclUniqueId) == NAVHIP_COMM_ID_BYTES, "NAVHIP_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

// [n][4] <- [n][2] | [n][2] for the rows [b, e), and back for the rows outside [b, e)
__global__ void k_comm_pack(float4 *pack, const float2 *pos, const float2 *vel, int b, int e)
{
    const int i = b + blockIdx.x * 256 + threadIdx.x;
    if(i < e) { const float2 p = pos[i], v = vel[i]; pack[i] = make_float4(p.x, p.y, v.x, v.y); }
}

__global__ void k_comm_unpack(const float4 *pack, float2 *pos, float2 *vel, int n, int b, int e)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i < n && (i < b || i >= e)) { const float4 r = pack[i]; pos[i] = make_float2(r.x, r.y); vel[i] = make_float2(r.z, r.w); }
}

static int allgather_rows(navhip_ctx *ctx, char *rows, size_t row_bytes, const int32_t *bounds, hipStream_t s)
{
    nh_comm *C = ctx->comm;
    const int world = C->world;
    bool equal = true;
    const int32_t per = bounds[1] - bounds[0];
    for(int r = 0; r < world; r++) {
        if(bounds[r + 1] < bounds[r]) return NAVHIP_ERR_INVALID;
        equal = equal && bounds[r] == r * per && bounds[r + 1] - bounds[r] == per;
    }
    if(equal) {
        // in place: every rank's send buffer is its own slab of the receive buffer
        NCCLCHK(ctx, g_rccl.AllGather(rows + (size_t)bounds[C->rank] * row_bytes, rows, (size_t)per * row_bytes, ncclChar,
                                      C->comm, s));
        return NAVHIP_OK;
    }
    // ragged slabs (movement.c:3759: the last slab of a ceil split is shorter): one broadcast per rank,
    // grouped into one launch
    NCCLCHK(ctx, g_rccl.GroupStart());
    for(int r = 0; r < world; r++) {
        const size_t cnt = (size_t)(bounds[r + 1] - bounds[r]) * row_bytes;
        if(!cnt) continue;
        char *p = rows + (size_t)bounds[r] * row_bytes;
        ncclResult_t rc = g_rccl.Broadcast(p, p, cnt, ncclChar, r, C->comm, s);
        if(rc != ncclSuccess) { g_rccl.GroupEnd(); ctx->last_error = std::string("ncclBroadcast: ") + g_rccl.GetErrorString(rc); return NAVHIP_ERR_DEVICE; }
    }
    NCCLCHK(ctx, g_rccl.GroupEnd());
    return NAVHIP_OK;
}

extern "C" {

int navhip_comm_unique_id(uint8_t out_id[NAVHIP_COMM_ID_BYTES])
{
    if(!out_id) return NAVHIP_ERR_INVALID;
    if(!rccl_load()) return NAVHIP_ERR_DEVICE;
    ncclUniqueId id;
    if(g_rccl.GetUniqueId(&id) != ncclSuccess) return NAVHIP_ERR_DEVICE;
    memcpy(out_id, &id, sizeof(id));
    return NAVHIP_OK;
}

int navhip_comm_init(navhip_ctx *ctx, int rank, int world, const uint8_t id[NAVHIP_COMM_ID_BYTES])
{
    if(!ctx || !id || world < 1 || rank < 0 || rank >= world) return NAVHIP_ERR_INVALID;
    if(!rccl_load()) { ctx->last_error = g_rccl_error; return NAVHIP_ERR_DEVICE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    navhip_comm_destroy(ctx);
    nh_comm *C = new (std::nothrow) nh_comm();
    if(!C) return NAVHIP_ERR_NOMEM;
    C->rank = rank; C->world = world; C->pack = nullptr; C->pack_cap = 0;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t rc = g_rccl.CommInitRank(&C->comm, world, uid, rank);
    if(rc != ncclSuccess) {
        ctx->last_error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc);
        delete C;
        return NAVHIP_ERR_DEVICE;
    }
    ctx->comm = C;
    return NAVHIP_OK;
}

void navhip_comm_destroy(navhip_ctx *ctx)
{
    if(!ctx || !ctx->comm) return;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    g_rccl.CommDestroy(ctx->comm->comm);
    if(ctx->comm->pack) hipFree(ctx->comm->pack);
    delete ctx->comm;
    ctx->comm = nullptr;
}

int navhip_comm_rank(const navhip_ctx *ctx)  { return ctx && ctx->comm ? ctx->comm->rank : -1; }
int navhip_comm_world(const navhip_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->world : 0; }

int navhip_comm_allgather_rows_dev(navhip_ctx *ctx, void *dev_rows, size_t row_bytes, const int32_t *bounds, void *stream)
{
    if(!ctx || !ctx->comm || !dev_rows || !bounds || row_bytes == 0) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return allgather_rows(ctx, (char*)dev_rows, row_bytes, bounds, stream ? (hipStream_t)stream : ctx->stream);
}

int navhip_comm_allgather_step_dev(navhip_ctx *ctx, float *dev_new_pos_xz, float *dev_vel_xz, const int32_t *bounds,
                                   void *stream)
{
    if(!ctx || !ctx->comm || !dev_new_pos_xz || !dev_vel_xz || !bounds) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    nh_comm *C = ctx->comm;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int n = bounds[C->world], b = bounds[C->rank], e = bounds[C->rank + 1];
    if(n <= 0 || bounds[0] != 0) return NAVHIP_ERR_INVALID;
    if(C->pack_cap < (size_t)n) {
        if(C->pack) HIPCHK(ctx, hipFree(C->pack));
        C->pack = nullptr; C->pack_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&C->pack, (size_t)n * 16));
        C->pack_cap = (size_t)n;
    }
    // ONE collective per tick: this rank's rows of [new position | velocity], 16 B per agent
    if(e > b)
        hipLaunchKernelGGL(k_comm_pack, dim3((e - b + 255) / 256), dim3(256), 0, s, (float4*)C->pack,
                           (const float2*)dev_new_pos_xz, (const float2*)dev_vel_xz, b, e);
    int rc = allgather_rows(ctx, (char*)C->pack, 16, bounds, s);
    if(rc) return rc;
    hipLaunchKernelGGL(k_comm_unpack, dim3((n + 255) / 256), dim3(256), 0, s, (const float4*)C->pack,
                       (float2*)dev_new_pos_xz, (float2*)dev_vel_xz, n, b, e);
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

}  // extern "C"
