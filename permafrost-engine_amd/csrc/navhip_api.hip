// navhip_api.hip -- the C ABI of libnavhip.so (include/navhip.h): context, map-plane
// residency in HBM, and the host-/device-buffer entry points that launch the kernels.
//
// Data layout in HBM (per nav layer, allocated on first upload; sized for the reference's
// maximum 64x64-chunk map this is 64 MB cost + 128 MB blockers + 128 MB local islands +
// 960 MB factions -- a fraction of the 288 GB part, so everything stays resident):
//   cost          u8  [chunks][64][64]       struct nav_chunk.cost_base      nav_data.h:123
//   blockers      u16 [chunks][64][64]       struct nav_chunk.blockers       nav_data.h:134
//   local_islands u16 [chunks][64][64]       struct nav_chunk.local_islands  nav_data.h:157
//   factions      u8  [chunks][15][64][64]   struct nav_chunk.factions       nav_data.h:141
//   passmask      u64 [chunks][64]           derived: row bitmasks of field_tile_passable
//   unit_cost     u8  [chunks]               derived: BFS kernel eligibility
#include "navhip_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#define HIPCHK(ctx, expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if(_e != hipSuccess) {                                                              \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);          \
            return NAVHIP_ERR_DEVICE;                                                       \
        }                                                                                   \
    } while(0)

static size_t plane_elem_bytes(int plane)
{
    switch(plane) {
    case NAVHIP_PLANE_COST_BASE:     return 1;
    case NAVHIP_PLANE_BLOCKERS:      return 2;
    case NAVHIP_PLANE_LOCAL_ISLANDS: return 2;
    case NAVHIP_PLANE_FACTIONS:      return NAVHIP_MAX_FACTIONS;
    default: return 0;
    }
}

static void **plane_slot(navhip_layer &L, int plane)
{
    switch(plane) {
    case NAVHIP_PLANE_COST_BASE:     return (void**)&L.cost;
    case NAVHIP_PLANE_BLOCKERS:      return (void**)&L.blockers;
    case NAVHIP_PLANE_LOCAL_ISLANDS: return (void**)&L.local_islands;
    case NAVHIP_PLANE_FACTIONS:      return (void**)&L.factions;
    default: return nullptr;
    }
}

static int ensure_cap(navhip_ctx *ctx, void **p, size_t *cap, size_t need)
{
    if(*cap >= need) return NAVHIP_OK;
    if(*p) HIPCHK(ctx, hipFree(*p));
    *p = nullptr; *cap = 0;
    size_t want = need + need / 2;
    HIPCHK(ctx, hipMalloc(p, want));
    *cap = want;
    return NAVHIP_OK;
}

extern "C" {

int navhip_ctx_create(navhip_ctx **out, int chunk_w, int chunk_h, int device)
{
    if(!out || chunk_w < 1 || chunk_h < 1 || chunk_w > 64 || chunk_h > 64)   // 6-bit chunk ids, nav.c:841-848
        return NAVHIP_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
        fprintf(stderr, "navhip: no usable HIP device (count=%d, asked for %d); this library has "
                        "no CPU fallback\n", ndev, device);
        return NAVHIP_ERR_DEVICE;
    }
    navhip_ctx *ctx = new (std::nothrow) navhip_ctx();
    if(!ctx) return NAVHIP_ERR_NOMEM;
    ctx->device = device;
    ctx->w = chunk_w; ctx->h = chunk_h; ctx->nchunks = chunk_w * chunk_h;
    ctx->field_kernel_mode = 0;
    memset(ctx->layers, 0, sizeof(ctx->layers));
    ctx->d_reqs = nullptr; ctx->d_reqs_cap = 0;
    ctx->d_dirs = nullptr; ctx->d_dirs_cap = 0;
    ctx->d_integ = nullptr; ctx->d_integ_cap = 0;
    ctx->d_reqmask = nullptr; ctx->d_reqmask_cap = 0;
    ctx->d_dirty_list = nullptr; ctx->d_dirty_cap = 0;
    if(hipSetDevice(device) != hipSuccess
    || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return NAVHIP_ERR_DEVICE;
    }
    *out = ctx;
    return NAVHIP_OK;
}

void navhip_ctx_destroy(navhip_ctx *ctx)
{
    if(!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        navhip_layer &L = ctx->layers[l];
        hipFree(L.cost); hipFree(L.blockers); hipFree(L.local_islands); hipFree(L.factions);
        hipFree(L.passmask); hipFree(L.unit_cost);
        free(L.dirty);
    }
    hipFree(ctx->d_reqs); hipFree(ctx->d_dirs); hipFree(ctx->d_integ); hipFree(ctx->d_reqmask);
    hipFree(ctx->d_dirty_list);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *navhip_last_error(const navhip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }
int   navhip_device(const navhip_ctx *ctx) { return ctx ? ctx->device : -1; }
void *navhip_stream(const navhip_ctx *ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int navhip_sync(navhip_ctx *ctx)
{
    if(!ctx) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NAVHIP_OK;
}

int navhip_set_field_kernel(navhip_ctx *ctx, int mode)
{
    if(!ctx || mode < 0 || mode > 1) return NAVHIP_ERR_INVALID;
    ctx->field_kernel_mode = mode;
    return NAVHIP_OK;
}

static int layer_prepare(navhip_ctx *ctx, int layer, int plane)
{
    navhip_layer &L = ctx->layers[layer];
    void **slot = plane_slot(L, plane);
    if(!*slot) {
        size_t bytes = (size_t)ctx->nchunks * NH_CELLS * plane_elem_bytes(plane);
        HIPCHK(ctx, hipMalloc(slot, bytes));
        if(plane != NAVHIP_PLANE_COST_BASE)
            HIPCHK(ctx, hipMemsetAsync(*slot, 0, bytes, ctx->stream));
    }
    if(!L.passmask) {
        HIPCHK(ctx, hipMalloc((void**)&L.passmask, (size_t)ctx->nchunks * 64 * sizeof(uint64_t)));
        HIPCHK(ctx, hipMalloc((void**)&L.unit_cost, (size_t)ctx->nchunks));
        L.dirty = (uint8_t*)calloc(ctx->nchunks, 1);
        if(!L.dirty) return NAVHIP_ERR_NOMEM;
    }
    return NAVHIP_OK;
}

int navhip_upload_plane(navhip_ctx *ctx, int layer, int plane, const void *host, size_t bytes)
{
    if(!ctx || !host || layer < 0 || layer >= NAVHIP_NAV_LAYER_MAX
    || plane < 0 || plane >= NAVHIP_PLANE_COUNT)
        return NAVHIP_ERR_INVALID;
    size_t want = (size_t)ctx->nchunks * NH_CELLS * plane_elem_bytes(plane);
    if(bytes != want) {
        ctx->last_error = "navhip_upload_plane: size mismatch";
        return NAVHIP_ERR_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = layer_prepare(ctx, layer, plane);
    if(rc) return rc;
    navhip_layer &L = ctx->layers[layer];
    HIPCHK(ctx, hipMemcpyAsync(*plane_slot(L, plane), host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // caller may reuse `host` immediately
    if(plane == NAVHIP_PLANE_COST_BASE || plane == NAVHIP_PLANE_BLOCKERS) {
        memset(L.dirty, 1, ctx->nchunks);
        L.any_dirty = true;
    }
    return NAVHIP_OK;
}

int navhip_upload_chunk(navhip_ctx *ctx, int layer, int plane, int chunk_r, int chunk_c,
                        const void *host, size_t bytes)
{
    if(!ctx || !host || layer < 0 || layer >= NAVHIP_NAV_LAYER_MAX
    || plane < 0 || plane >= NAVHIP_PLANE_COUNT
    || chunk_r < 0 || chunk_r >= ctx->h || chunk_c < 0 || chunk_c >= ctx->w)
        return NAVHIP_ERR_INVALID;
    size_t per = (size_t)NH_CELLS * plane_elem_bytes(plane);
    if(bytes != per) {
        ctx->last_error = "navhip_upload_chunk: size mismatch";
        return NAVHIP_ERR_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = layer_prepare(ctx, layer, plane);
    if(rc) return rc;
    navhip_layer &L = ctx->layers[layer];
    int chunk = chunk_r * ctx->w + chunk_c;
    HIPCHK(ctx, hipMemcpyAsync((char*)*plane_slot(L, plane) + (size_t)chunk * per, host, per,
                               hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if(plane == NAVHIP_PLANE_COST_BASE || plane == NAVHIP_PLANE_BLOCKERS) {
        L.dirty[chunk] = 1;
        L.any_dirty = true;
    }
    return NAVHIP_OK;
}

void *navhip_plane_dev(navhip_ctx *ctx, int layer, int plane)
{
    if(!ctx || layer < 0 || layer >= NAVHIP_NAV_LAYER_MAX || plane < 0 || plane >= NAVHIP_PLANE_COUNT)
        return nullptr;
    return *plane_slot(ctx->layers[layer], plane);
}

// rebuild passmask / unit_cost of chunks whose cost or blockers changed
static int refresh_derived(navhip_ctx *ctx, hipStream_t s)
{
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        navhip_layer &L = ctx->layers[l];
        if(!L.any_dirty || !L.cost) continue;
        std::vector<uint32_t> list;
        for(int i = 0; i < ctx->nchunks; i++)
            if(L.dirty[i]) list.push_back((uint32_t)i);
        if(!list.empty()) {
            if((int)list.size() == ctx->nchunks) {
                nh_launch_derive(ctx, l, nullptr, ctx->nchunks, s);
            }else{
                int rc = ensure_cap(ctx, (void**)&ctx->d_dirty_list, &ctx->d_dirty_cap,
                                    list.size() * sizeof(uint32_t));
                if(rc) return rc;
                HIPCHK(ctx, hipMemcpyAsync(ctx->d_dirty_list, list.data(),
                                           list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
                nh_launch_derive(ctx, l, ctx->d_dirty_list, (int)list.size(), s);
                HIPCHK(ctx, hipStreamSynchronize(s));   // list buffer is reused per layer
            }
            HIPCHK(ctx, hipGetLastError());
        }
        memset(L.dirty, 0, ctx->nchunks);
        L.any_dirty = false;
    }
    return NAVHIP_OK;
}

static int validate_reqs(navhip_ctx *ctx, const navhip_field_req *reqs, int n)
{
    for(int i = 0; i < n; i++) {
        const navhip_field_req &r = reqs[i];
        bool ok = r.layer < NAVHIP_NAV_LAYER_MAX && r.chunk_r < ctx->h && r.chunk_c < ctx->w
               && (r.type == NAVHIP_TARGET_TILE || r.type == NAVHIP_TARGET_PORTAL);
        if(ok && r.type == NAVHIP_TARGET_TILE)
            ok = r.tile_r < 64 && r.tile_c < 64;
        if(ok && r.type == NAVHIP_TARGET_PORTAL)
            ok = r.port_r0 <= r.port_r1 && r.port_r1 < 64 && r.port_c0 <= r.port_c1 && r.port_c1 < 64
              && r.next_r0 <= r.next_r1 && r.next_r1 < 64 && r.next_c0 <= r.next_c1 && r.next_c1 < 64
              && r.next_chunk_r < ctx->h && r.next_chunk_c < ctx->w;
        if(!ok) {
            ctx->last_error = "navhip_build_fields: malformed request " + std::to_string(i);
            return NAVHIP_ERR_INVALID;
        }
        const navhip_layer &L = ctx->layers[r.layer];
        if(!L.cost || (r.type == NAVHIP_TARGET_PORTAL && !L.local_islands)
        || (r.faction_id != NAVHIP_FACTION_ID_NONE && !L.factions && L.blockers)) {
            ctx->last_error = "navhip_build_fields: request " + std::to_string(i)
                            + " needs a plane that was never uploaded";
            return NAVHIP_ERR_NOT_UPLOADED;
        }
    }
    return NAVHIP_OK;
}

int navhip_build_fields_dev(navhip_ctx *ctx, const navhip_field_req *dev_reqs, int n,
                            uint8_t *dev_inout_dirs, float *dev_out_integ, void *stream)
{
    if(!ctx || n < 0 || (n > 0 && (!dev_reqs || !dev_inout_dirs))) return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int rc = refresh_derived(ctx, s);
    if(rc) return rc;
    nh_launch_fields(ctx, dev_reqs, n, dev_inout_dirs, dev_out_integ, s);
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_build_fields(navhip_ctx *ctx, const navhip_field_req *reqs, int n,
                        uint8_t *inout_dirs, float *out_integ)
{
    if(!ctx || n < 0 || (n > 0 && (!reqs || !inout_dirs))) return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = validate_reqs(ctx, reqs, n);
    if(rc) return rc;
    hipStream_t s = ctx->stream;
    rc = ensure_cap(ctx, &ctx->d_reqs, &ctx->d_reqs_cap, (size_t)n * sizeof(navhip_field_req));
    if(rc) return rc;
    rc = ensure_cap(ctx, (void**)&ctx->d_dirs, &ctx->d_dirs_cap, (size_t)n * NH_CELLS);
    if(rc) return rc;
    if(out_integ) {
        rc = ensure_cap(ctx, (void**)&ctx->d_integ, &ctx->d_integ_cap,
                        (size_t)n * NH_CELLS * sizeof(float));
        if(rc) return rc;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_reqs, reqs, (size_t)n * sizeof(navhip_field_req),
                               hipMemcpyHostToDevice, s));
    bool any_inout = false;
    for(int i = 0; i < n; i++) any_inout |= (reqs[i].flags & NAVHIP_REQ_INOUT) != 0;
    if(any_inout)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_dirs, inout_dirs, (size_t)n * NH_CELLS,
                                   hipMemcpyHostToDevice, s));
    rc = navhip_build_fields_dev(ctx, (const navhip_field_req*)ctx->d_reqs, n, ctx->d_dirs,
                                 out_integ ? ctx->d_integ : nullptr, s);
    if(rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(inout_dirs, ctx->d_dirs, (size_t)n * NH_CELLS,
                               hipMemcpyDeviceToHost, s));
    if(out_integ)
        HIPCHK(ctx, hipMemcpyAsync(out_integ, ctx->d_integ, (size_t)n * NH_CELLS * sizeof(float),
                                   hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

uint64_t navhip_flow_field_id(const navhip_field_req *r)
{
    // N_FlowFieldID, field.c:1952-1975
    if(r->type == NAVHIP_TARGET_PORTAL) {
        return (((uint64_t)r->layer)            << 60)
             | (((uint64_t)r->type)             << 56)
             | (((uint64_t)(r->next_iid & 0xf)) << 48)
             | (((uint64_t)(r->port_iid & 0xf)) << 40)
             | (((uint64_t)r->port_r0)          << 34)
             | (((uint64_t)r->port_c0)          << 28)
             | (((uint64_t)r->port_r1)          << 22)
             | (((uint64_t)r->port_c1)          << 16)
             | (((uint64_t)r->chunk_r)          <<  8)
             | (((uint64_t)r->chunk_c)          <<  0);
    }
    return (((uint64_t)r->layer)   << 60)
         | (((uint64_t)r->type)    << 56)
         | (((uint64_t)r->tile_r)  << 24)
         | (((uint64_t)r->tile_c)  << 16)
         | (((uint64_t)r->chunk_r) <<  8)
         | (((uint64_t)r->chunk_c) <<  0);
}

} // extern "C"
