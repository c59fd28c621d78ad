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
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// a host without the RCCL headers still builds the library (librccl is only looked for at run time): the few
// types and constants of the entry points bound below, as rccl.h declares them
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
}
#endif

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

// How rows travel between ranks.  RCCL over xGMI in production; the MAILBOX transport for bringing the exchange
// step up (and testing it) where there is one device or no RCCL: the "network" is a device buffer laid out like
// the receive buffer, into which the caller has put the other ranks' rows -- a rank's own rows are deposited
// there, everybody else's are taken from there.  Everything around the transport (packing, slab bounds, the
// ragged grouping, unpacking) is the same code on both.
struct nh_comm {
    ncclComm_t comm;                          // (null: mailbox transport)
    int        rank, world;
    float     *pack;   size_t pack_cap;       // [n][4] staging of navhip_comm_allgather_step_dev
    char      *mailbox; size_t mailbox_bytes; // mailbox transport: laid out like the receive buffer
    char      *cur_base;                      // (mailbox transport) the receive buffer of the collective in flight
};

// the two collectives the exchange uses, on either transport
static int tr_allgather(navhip_ctx *ctx, const void *send, void *recv, size_t cnt, hipStream_t s)
{
    nh_comm *C = ctx->comm;
    if(C->comm) {
        ncclResult_t r = g_rccl.AllGather(send, recv, cnt, ncclChar, C->comm, s);
        if(r != ncclSuccess) { ctx->last_error = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return NAVHIP_ERR_DEVICE; }
        return NAVHIP_OK;
    }
    if(cnt * (size_t)C->world > C->mailbox_bytes) { ctx->last_error = "mailbox smaller than the gathered rows"; return NAVHIP_ERR_INVALID; }
    if(cnt) HIPCHK(ctx, hipMemcpyAsync(C->mailbox + (size_t)C->rank * cnt, send, cnt, hipMemcpyDeviceToDevice, s));
    for(int r = 0; r < C->world; r++)
        if(r != C->rank && cnt)
            HIPCHK(ctx, hipMemcpyAsync((char*)recv + (size_t)r * cnt, C->mailbox + (size_t)r * cnt, cnt, hipMemcpyDeviceToDevice, s));
    return NAVHIP_OK;
}

static int tr_broadcast(navhip_ctx *ctx, void *buf, size_t cnt, int root, hipStream_t s)
{
    nh_comm *C = ctx->comm;
    if(C->comm) {
        ncclResult_t r = g_rccl.Broadcast(buf, buf, cnt, ncclChar, root, C->comm, s);
        if(r != ncclSuccess) { ctx->last_error = std::string("ncclBroadcast: ") + g_rccl.GetErrorString(r); return NAVHIP_ERR_DEVICE; }
        return NAVHIP_OK;
    }
    const size_t off = (size_t)((char*)buf - C->cur_base);
    if(off + cnt > C->mailbox_bytes) { ctx->last_error = "mailbox smaller than the gathered rows"; return NAVHIP_ERR_INVALID; }
    if(root == C->rank) HIPCHK(ctx, hipMemcpyAsync(C->mailbox + off, buf, cnt, hipMemcpyDeviceToDevice, s));
    else                HIPCHK(ctx, hipMemcpyAsync(buf, C->mailbox + off, cnt, hipMemcpyDeviceToDevice, s));
    return NAVHIP_OK;
}

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
    if(bounds[0] != 0) return NAVHIP_ERR_INVALID;
    bool equal = true;
    const int32_t per = bounds[1] - bounds[0];
    for(int r = 0; r < world; r++) {
        if(bounds[r + 1] < bounds[r]) return NAVHIP_ERR_INVALID;
        equal = equal && bounds[r] == r * per && bounds[r + 1] - bounds[r] == per;
    }
    C->cur_base = rows;
    if(equal)         // in place: every rank's send buffer is its own slab of the receive buffer
        return tr_allgather(ctx, rows + (size_t)bounds[C->rank] * row_bytes, rows, (size_t)per * row_bytes, s);
    // ragged slabs (movement.c:3759: the last slab of a ceil split is shorter): one broadcast per rank,
    // grouped into one launch
    if(C->comm) NCCLCHK(ctx, g_rccl.GroupStart());
    int rc = NAVHIP_OK;
    for(int r = 0; r < world && !rc; r++) {
        const size_t cnt = (size_t)(bounds[r + 1] - bounds[r]) * row_bytes;
        if(!cnt) continue;
        rc = tr_broadcast(ctx, rows + (size_t)bounds[r] * row_bytes, cnt, r, s);
    }
    if(C->comm) { if(rc) g_rccl.GroupEnd(); else NCCLCHK(ctx, g_rccl.GroupEnd()); }
    return rc;
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
    C->mailbox = nullptr; C->mailbox_bytes = 0; C->cur_base = nullptr;
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

int navhip_comm_init_mailbox(navhip_ctx *ctx, int rank, int world, void *dev_mailbox, size_t mailbox_bytes)
{
    if(!ctx || !dev_mailbox || world < 1 || rank < 0 || rank >= world) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    navhip_comm_destroy(ctx);
    nh_comm *C = new (std::nothrow) nh_comm();
    if(!C) return NAVHIP_ERR_NOMEM;
    C->comm = nullptr; C->rank = rank; C->world = world; C->pack = nullptr; C->pack_cap = 0;
    C->mailbox = (char*)dev_mailbox; C->mailbox_bytes = mailbox_bytes; C->cur_base = nullptr;
    ctx->comm = C;
    return NAVHIP_OK;
}

void navhip_comm_destroy(navhip_ctx *ctx)
{
    if(!ctx || !ctx->comm) return;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    if(ctx->comm->comm) g_rccl.CommDestroy(ctx->comm->comm);
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
