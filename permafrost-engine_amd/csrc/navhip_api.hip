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
//   probemask     u64 [chunks][64][2]        derived: row bitmasks cost_base != 0xff | blockers > 0 (tile probes)
//   unit_cost     u8  [chunks]               derived: BFS kernel eligibility
#include "navhip_internal.h"
#include "agent_internal.h"
#include "agent_thread.h"
#include <cmath>

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

static int refresh_derived(navhip_ctx *ctx, hipStream_t s);
static int ensure_buf(navhip_ctx *ctx, navhip_ctx::buf &b, size_t need);

static size_t plane_elem_bytes(int plane)
{
    switch(plane) {
    case NAVHIP_PLANE_COST_BASE:     return 1;
    case NAVHIP_PLANE_BLOCKERS:      return 2;
    case NAVHIP_PLANE_LOCAL_ISLANDS: return 2;
    case NAVHIP_PLANE_FACTIONS:      return NAVHIP_MAX_FACTIONS;
    case NAVHIP_PLANE_ISLANDS:       return 2;
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
    case NAVHIP_PLANE_ISLANDS:       return (void**)&L.islands;
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
    memset(ctx->sp, 0, sizeof(ctx->sp));
    memset(&ctx->coh, 0, sizeof(ctx->coh));
    memset(&ctx->coh_plan, 0, sizeof(ctx->coh_plan));
    memset(&ctx->gen_list, 0, sizeof(ctx->gen_list));
    memset(&ctx->counters, 0, sizeof(ctx->counters));
    ctx->gen_launches = 0;
    ctx->coh_flocks = ctx->coh_members = -1;
    ctx->coh_parity = 0; ctx->coh_unique = 0;
    ctx->ev_regroup = nullptr;
    ctx->regroup_pending = false;
    memset(ctx->coh_regroup_key, 0xff, sizeof(ctx->coh_regroup_key)); ctx->coh_regroup_age = 0;
    memset(&ctx->midrec, 0, sizeof(ctx->midrec));
    memset(ctx->nbr, 0, sizeof(ctx->nbr));
    memset(ctx->wl, 0, sizeof(ctx->wl));
    ctx->wl_parity = 0;
    memset(ctx->stage, 0, sizeof(ctx->stage));
    ctx->profiling = false; ctx->ev_valid = false;
    ctx->aux[0] = ctx->aux[1] = nullptr; ctx->aux_main = nullptr;
    ctx->front_stream = nullptr; ctx->join0_signalled = ctx->lists_signalled = false; ctx->step_end_on = nullptr; ctx->step_end_signalled = false; ctx->start_seq = 0; ctx->start_flag = NH_HO_START;
    memset(&ctx->pre, 0, sizeof(ctx->pre));
    ctx->pool = nullptr; ctx->async = nullptr; ctx->comm = nullptr; ctx->ho = nullptr; ctx->sp_builds = 0; ctx->lists_pinned = nullptr;
    memset(ctx->ev, 0, sizeof(ctx->ev));
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
    navhip_comm_destroy(ctx);
    if(ctx->lists_pinned) hipHostFree(ctx->lists_pinned);
    navhip_pool_destroy(ctx);
    nh_async_destroy(ctx);
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        navhip_layer &L = ctx->layers[l];
        hipFree(L.cost); hipFree(L.blockers); hipFree(L.local_islands); hipFree(L.factions);
        hipFree(L.islands);
        hipFree(L.passmask); hipFree(L.probemask); hipFree(L.unit_cost); hipFree(L.touched); hipFree(L.changed);
        free(L.dirty);
    }
    hipFree(ctx->d_reqs); hipFree(ctx->d_dirs); hipFree(ctx->d_integ); hipFree(ctx->d_reqmask);
    hipFree(ctx->d_dirty_list);
    for(auto &b : ctx->sp) hipFree(b.p);
    for(auto &b : ctx->stage) hipFree(b.p);
    hipFree(ctx->coh.p); hipFree(ctx->coh_plan.p); hipFree(ctx->midrec.p); hipFree(ctx->gen_list.p);
    for(auto &b : ctx->nbr) hipFree(b.p);
    for(auto &b : ctx->arrived) hipFree(b.p);
    for(auto &b : ctx->wl) hipFree(b.p);
    for(auto &e : ctx->ev) if(e) hipEventDestroy(e);
    if(nh_streams_alive(ctx->device))
        for(auto &a : ctx->aux) if(a) hipStreamSynchronize(a);   // (borrowed: the process's own set, csrc/stream_set.hip)
    nh_handover_destroy(ctx);
    if(ctx->ev_regroup) hipEventDestroy(ctx->ev_regroup);
    nh_streams_forget(ctx->device, ctx->stream);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *navhip_last_error(const navhip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }
int   navhip_device(const navhip_ctx *ctx) { return ctx ? ctx->device : -1; }
void *navhip_stream(const navhip_ctx *ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int navhip_stream_beside(navhip_ctx *ctx, void *main_stream, int cu_begin, int cu_count, void **out_stream)
{
    if(!ctx || !out_stream || cu_begin < 0) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    if(cu_count <= 0) {
        hipStream_t all[NH_STREAM_FIXED];
        int rc = nh_streams_for(ctx, (hipStream_t)main_stream, all);
        if(rc) return rc;
        st = all[NH_STREAM_FIELDS];
    }else{
        st = nh_stream_partial_for(ctx, (hipStream_t)main_stream, cu_begin, cu_count);
        if(!st) return ctx->last_error.empty() ? NAVHIP_ERR_INVALID : NAVHIP_ERR_DEVICE;
    }
    *out_stream = (void*)st;
    return NAVHIP_OK;
}

int navhip_stream_main(navhip_ctx *ctx, void **out_stream)
{
    if(!ctx || !out_stream) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t all[NH_STREAM_FIXED];
    int rc = nh_streams_for(ctx, nullptr, all);
    if(rc) return rc;
    *out_stream = (void*)all[NH_STREAM_MAIN];
    return NAVHIP_OK;
}

int navhip_stream_create_partial(navhip_ctx *ctx, int cu_begin, int cu_count, void **out_stream)
{
    if(cu_count <= 0) return NAVHIP_ERR_INVALID;
    return navhip_stream_beside(ctx, ctx ? (void*)ctx->stream : nullptr, cu_begin, cu_count, out_stream);
}

int navhip_sync(navhip_ctx *ctx)
{
    if(!ctx) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for(auto a : ctx->aux) if(a) HIPCHK(ctx, hipStreamSynchronize(a));       // prefetch side streams
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
        HIPCHK(ctx, hipMalloc((void**)&L.probemask, (size_t)ctx->nchunks * 128 * sizeof(uint64_t)));
        HIPCHK(ctx, hipMalloc((void**)&L.unit_cost, (size_t)ctx->nchunks));
        HIPCHK(ctx, hipMalloc((void**)&L.touched, (size_t)ctx->nchunks));
        HIPCHK(ctx, hipMalloc((void**)&L.changed, (size_t)ctx->nchunks));
        HIPCHK(ctx, hipMemsetAsync(L.touched, 0, (size_t)ctx->nchunks, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(L.changed, 0, (size_t)ctx->nchunks, ctx->stream));
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
    if(plane == NAVHIP_PLANE_COST_BASE) {
        // does the layer hold costs other than 1 / impassable?  (The reference's cost writers produce no
        // others, nav.c:339-342,416; a host that does sends its requests to the relaxation kernel, whose
        // launch is then sized for real work instead of for an empty list.)
        const uint8_t *c = (const uint8_t*)host;
        bool other = false;
        for(size_t i = 0; i < bytes && !other; i++) other = c[i] != 1 && c[i] != NAVHIP_COST_IMPASSABLE;
        L.nonunit_costs = other;
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
    if(plane == NAVHIP_PLANE_COST_BASE) {
        const uint8_t *c = (const uint8_t*)host;
        for(size_t i = 0; i < per; i++) if(c[i] != 1 && c[i] != NAVHIP_COST_IMPASSABLE) { L.nonunit_costs = true; break; }
    }
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

// rebuild passmask / probemask / unit_cost of chunks whose cost or blockers changed (after an upload: the
// device-side blocker updates refresh their chunks themselves).  Whatever it launches has completed when it
// returns, so consumers on any stream may follow.
static int refresh_derived(navhip_ctx *ctx, hipStream_t s)
{
    bool launched = false;
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
            launched = true;
        }
        memset(L.dirty, 0, ctx->nchunks);
        L.any_dirty = false;
    }
    if(launched) HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

int navhip_download_plane(navhip_ctx *ctx, int layer, int plane, void *host, size_t bytes)
{
    if(!ctx || !host || layer < 0 || layer >= NAVHIP_NAV_LAYER_MAX
    || plane < 0 || plane >= NAVHIP_PLANE_COUNT)
        return NAVHIP_ERR_INVALID;
    void *src = *plane_slot(ctx->layers[layer], plane);
    if(!src) return NAVHIP_ERR_NOT_UPLOADED;
    if(bytes != (size_t)ctx->nchunks * NH_CELLS * plane_elem_bytes(plane)) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NAVHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// dynamic obstacles
// ---------------------------------------------------------------------------------------------
int navhip_blockers_circles_dev(navhip_ctx *ctx, const navhip_circle *dev_circles, int n,
                                float map_pos_x, float map_pos_z, void *stream)
{
    if(!ctx || n < 0 || (n > 0 && !dev_circles)) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    ctx->counters.blocker_circles += (uint64_t)n;
    int rc = refresh_derived(ctx, s);          // the "before" masks must be current
    if(rc) return rc;
    bool any = false;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) any |= ctx->layers[l].blockers != nullptr;
    if(!any) {
        ctx->last_error = "navhip_blockers_circles: no blockers plane resident";
        return NAVHIP_ERR_NOT_UPLOADED;
    }
    nh_launch_blockers_circles(ctx, dev_circles, n, map_pos_x, map_pos_z, s);
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_blockers_circles(navhip_ctx *ctx, const navhip_circle *circles, int n,
                            float map_pos_x, float map_pos_z)
{
    if(!ctx || n < 0 || (n > 0 && !circles)) return NAVHIP_ERR_INVALID;
    for(int i = 0; i < n; i++) {
        const navhip_circle &c = circles[i];
        if(!(c.radius >= 0.0f) || std::ceil((double)(c.radius / 4)) > 28.0
        || (c.delta != 1 && c.delta != -1) || c.faction_id < 0 || c.faction_id >= NAVHIP_MAX_FACTIONS) {
            ctx->last_error = "navhip_blockers_circles: circle " + std::to_string(i)
                            + " outside the device path (radius > 112, delta != +-1 or bad faction)";
            return NAVHIP_ERR_INVALID;
        }
    }
    if(n == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_buf(ctx, ctx->stage[23], (size_t)n * sizeof(navhip_circle));
    if(rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[23].p, circles, (size_t)n * sizeof(navhip_circle),
                               hipMemcpyHostToDevice, ctx->stream));
    rc = navhip_blockers_circles_dev(ctx, (const navhip_circle*)ctx->stage[23].p, n, map_pos_x,
                                     map_pos_z, ctx->stream);
    if(rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NAVHIP_OK;
}

int navhip_relabel_local_islands(navhip_ctx *ctx, int layer)
{
    if(!ctx || layer < 0 || layer >= NAVHIP_NAV_LAYER_MAX) return NAVHIP_ERR_INVALID;
    if(!ctx->layers[layer].cost) return NAVHIP_ERR_NOT_UPLOADED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = layer_prepare(ctx, layer, NAVHIP_PLANE_LOCAL_ISLANDS);
    if(rc) return rc;
    rc = refresh_derived(ctx, ctx->stream);
    if(rc) return rc;
    nh_launch_local_islands(ctx, layer, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NAVHIP_OK;
}

int navhip_changed_chunks(navhip_ctx *ctx, int layer, uint8_t *host_flags, int clear)
{
    if(!ctx || !host_flags || layer < 0 || layer >= NAVHIP_NAV_LAYER_MAX) return NAVHIP_ERR_INVALID;
    navhip_layer &L = ctx->layers[layer];
    if(!L.changed) { memset(host_flags, 0, (size_t)ctx->nchunks); return NAVHIP_OK; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(host_flags, L.changed, (size_t)ctx->nchunks, hipMemcpyDeviceToHost,
                               ctx->stream));
    if(clear) HIPCHK(ctx, hipMemsetAsync(L.changed, 0, (size_t)ctx->nchunks, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NAVHIP_OK;
}

int navhip_clear_changed(navhip_ctx *ctx, void *stream)
{
    if(!ctx) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++)
        if(ctx->layers[l].changed)
            HIPCHK(ctx, hipMemsetAsync(ctx->layers[l].changed, 0, (size_t)ctx->nchunks, s));
    return NAVHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// region fields
// ---------------------------------------------------------------------------------------------
int navhip_build_region_fields_dev(navhip_ctx *ctx, const navhip_region_req *dev_reqs, int n,
                                   int max_dim, const int16_t *dev_seeds, const int16_t *dev_overlay,
                                   uint8_t *dev_inout, size_t out_stride, void *stream)
{
    if(!ctx || n < 0 || (n > 0 && (!dev_reqs || !dev_inout)) || max_dim < 2 || max_dim > 128)
        return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    ctx->counters.region_fields += (uint64_t)n;
    nh_launch_region_fields(ctx, dev_reqs, n, max_dim, dev_seeds, dev_overlay, dev_inout, out_stride, s);
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_build_region_fields(navhip_ctx *ctx, const navhip_region_req *reqs, int n,
                               const int16_t *seeds, size_t n_seeds,
                               const int16_t *overlay, size_t n_overlay,
                               uint8_t *inout, size_t out_stride)
{
    if(!ctx || n < 0 || (n > 0 && (!reqs || !inout))) return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    int max_dim = 0;
    bool any_window = false;
    for(int i = 0; i < n; i++) {
        const navhip_region_req &r = reqs[i];
        bool ok = r.layer < NAVHIP_NAV_LAYER_MAX && r.out_mode <= 1 && r.rdim == r.cdim
               && r.rdim >= 2 && r.rdim <= 128 && (r.rdim % 2) == 0
               && (size_t)r.seed_begin + r.seed_count <= n_seeds
               && (size_t)r.overlay_begin + r.overlay_count <= n_overlay
               && (r.seed_count == 0 || seeds) && (r.overlay_count == 0 || overlay);
        if(ok && r.out_mode == 1)
            ok = r.roff + (r.rdim < 64 ? r.rdim : 64) <= r.rdim && r.coff + (r.cdim < 64 ? r.cdim : 64) <= r.cdim
              && out_stride >= NH_CELLS;
        if(ok && r.out_mode == 0) ok = out_stride >= (size_t)r.rdim * r.cdim / 2;
        if(!ok) {
            ctx->last_error = "navhip_build_region_fields: malformed request " + std::to_string(i);
            return NAVHIP_ERR_INVALID;
        }
        if(!ctx->layers[r.layer].cost) return NAVHIP_ERR_NOT_UPLOADED;
        max_dim = r.rdim > max_dim ? r.rdim : max_dim;
        any_window |= r.out_mode == 1;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc = ensure_buf(ctx, ctx->stage[32], (size_t)n * sizeof(navhip_region_req));
    if(!rc) rc = ensure_buf(ctx, ctx->stage[33], n_seeds * 4);
    if(!rc) rc = ensure_buf(ctx, ctx->stage[34], n_overlay * 4);
    if(!rc) rc = ensure_buf(ctx, ctx->stage[35], (size_t)n * out_stride);
    if(rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[32].p, reqs, (size_t)n * sizeof(navhip_region_req),
                               hipMemcpyHostToDevice, s));
    if(n_seeds) HIPCHK(ctx, hipMemcpyAsync(ctx->stage[33].p, seeds, n_seeds * 4, hipMemcpyHostToDevice, s));
    if(n_overlay) HIPCHK(ctx, hipMemcpyAsync(ctx->stage[34].p, overlay, n_overlay * 4, hipMemcpyHostToDevice, s));
    if(any_window)
        HIPCHK(ctx, hipMemcpyAsync(ctx->stage[35].p, inout, (size_t)n * out_stride, hipMemcpyHostToDevice, s));
    rc = navhip_build_region_fields_dev(ctx, (const navhip_region_req*)ctx->stage[32].p, n, max_dim,
                                        (const int16_t*)ctx->stage[33].p, (const int16_t*)ctx->stage[34].p,
                                        (uint8_t*)ctx->stage[35].p, out_stride, s);
    if(rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(inout, ctx->stage[35].p, (size_t)n * out_stride, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// LOS fields
// ---------------------------------------------------------------------------------------------
int navhip_build_los_dev(navhip_ctx *ctx, const navhip_los_req *dev_reqs, int n,
                         const uint8_t *dev_prev_fields, uint8_t *dev_out_fields,
                         float map_pos_x, float map_pos_z, void *stream)
{
    if(!ctx || n < 0 || (n > 0 && (!dev_reqs || !dev_out_fields))) return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    ctx->counters.los_fields += (uint64_t)n;
    nh_launch_los(ctx, dev_reqs, n, dev_prev_fields, dev_out_fields, map_pos_x, map_pos_z, s);
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_build_los(navhip_ctx *ctx, const navhip_los_req *reqs, int n,
                     const uint8_t *prev_fields, uint8_t *out_fields,
                     float map_pos_x, float map_pos_z)
{
    if(!ctx || n < 0 || (n > 0 && (!reqs || !out_fields))) return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    bool any_prev = false;
    for(int i = 0; i < n; i++) {
        const navhip_los_req &r = reqs[i];
        const bool has_prev = r.prev_dr != 0 || r.prev_dc != 0;
        bool ok = r.layer < NAVHIP_NAV_LAYER_MAX && r.chunk_r < ctx->h && r.chunk_c < ctx->w
               && r.target_chunk_r < ctx->h && r.target_chunk_c < ctx->w
               && r.target_tile_r < 64 && r.target_tile_c < 64
               && (!has_prev || ((r.prev_dr == 0) != (r.prev_dc == 0)
                                 && r.prev_dr >= -1 && r.prev_dr <= 1 && r.prev_dc >= -1 && r.prev_dc <= 1))
               && (has_prev || (r.chunk_r == r.target_chunk_r && r.chunk_c == r.target_chunk_c));
        if(!ok) {
            ctx->last_error = "navhip_build_los: malformed request " + std::to_string(i);
            return NAVHIP_ERR_INVALID;
        }
        if(!ctx->layers[r.layer].cost) return NAVHIP_ERR_NOT_UPLOADED;
        any_prev |= has_prev;
    }
    if(any_prev && !prev_fields) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc = ensure_buf(ctx, ctx->stage[29], (size_t)n * sizeof(navhip_los_req));
    if(!rc) rc = ensure_buf(ctx, ctx->stage[30], (size_t)n * NH_CELLS);
    if(!rc) rc = ensure_buf(ctx, ctx->stage[31], (size_t)n * NH_CELLS);
    if(rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage[29].p, reqs, (size_t)n * sizeof(navhip_los_req),
                               hipMemcpyHostToDevice, s));
    if(any_prev)
        HIPCHK(ctx, hipMemcpyAsync(ctx->stage[30].p, prev_fields, (size_t)n * NH_CELLS,
                                   hipMemcpyHostToDevice, s));
    rc = navhip_build_los_dev(ctx, (const navhip_los_req*)ctx->stage[29].p, n,
                              (const uint8_t*)ctx->stage[30].p, (uint8_t*)ctx->stage[31].p,
                              map_pos_x, map_pos_z, s);
    if(rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_fields, ctx->stage[31].p, (size_t)n * NH_CELLS,
                               hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

int nh_validate_field_reqs(navhip_ctx *ctx, const navhip_field_req *reqs, int n)
{
    for(int i = 0; i < n; i++) {
        const navhip_field_req &r = reqs[i];
        bool ok = r.layer < NAVHIP_NAV_LAYER_MAX && r.chunk_r < ctx->h && r.chunk_c < ctx->w
               && (r.type == NAVHIP_TARGET_TILE || r.type == NAVHIP_TARGET_PORTAL
                || r.type == NAVHIP_TARGET_NEAREST_PATHABLE);
        if(ok && r.type != NAVHIP_TARGET_PORTAL)
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
        || ((r.flags & NAVHIP_REQ_ISLAND_NEAREST) && (!L.local_islands || !L.islands))
        || (r.faction_id != NAVHIP_FACTION_ID_NONE && !L.factions && L.blockers)) {
            ctx->last_error = "navhip_build_fields: request " + std::to_string(i)
                            + " needs a plane that was never uploaded";
            return NAVHIP_ERR_NOT_UPLOADED;
        }
    }
    return NAVHIP_OK;
}

static int build_fields_on(navhip_ctx *ctx, const navhip_field_req *dev_reqs, int n, uint8_t *dev_inout_dirs,
                           float *dev_out_integ, const int32_t *dev_slots, hipStream_t s)
{
    int rc = refresh_derived(ctx, s);
    if(rc) return rc;
    // work list of the generic kernel: the header is zero between launches (the kernel resets it)
    const void *old_list = ctx->gen_list.p;
    rc = ensure_buf(ctx, ctx->gen_list, ((size_t)n + 2) * sizeof(int32_t));
    if(rc) return rc;
    if(ctx->gen_list.p != old_list) HIPCHK(ctx, hipMemsetAsync(ctx->gen_list.p, 0, 2 * sizeof(int32_t), s));
    nh_launch_fields(ctx, dev_reqs, n, dev_inout_dirs, dev_out_integ, (int32_t*)ctx->gen_list.p, s, dev_slots);
    HIPCHK(ctx, hipGetLastError());
    ctx->counters.field_calls++; ctx->counters.chunk_fields += (uint64_t)n;
    return NAVHIP_OK;
}

int navhip_build_fields_dev(navhip_ctx *ctx, const navhip_field_req *dev_reqs, int n,
                            uint8_t *dev_inout_dirs, float *dev_out_integ, void *stream)
{
    if(!ctx || n < 0 || (n > 0 && (!dev_reqs || !dev_inout_dirs))) return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return build_fields_on(ctx, dev_reqs, n, dev_inout_dirs, dev_out_integ, nullptr,
                           stream ? (hipStream_t)stream : ctx->stream);
}

}  // extern "C"

// request i is built into slot dev_slots[i] of dev_fields (navhip_pool_build)
int navhip_build_fields_slots_dev(navhip_ctx *ctx, const navhip_field_req *dev_reqs, int n, uint8_t *dev_fields,
                                  const int32_t *dev_slots, hipStream_t s)
{
    if(n == 0) return NAVHIP_OK;
    return build_fields_on(ctx, dev_reqs, n, dev_fields, nullptr, dev_slots, s);
}

int navhip_stage_reserve(navhip_ctx *ctx, int slot, size_t bytes, void **dev)
{
    int rc = ensure_buf(ctx, ctx->stage[slot], bytes);
    if(rc) return rc;
    *dev = ctx->stage[slot].p;
    return NAVHIP_OK;
}

extern "C" {

int navhip_build_fields(navhip_ctx *ctx, const navhip_field_req *reqs, int n,
                        uint8_t *inout_dirs, float *out_integ)
{
    if(!ctx || n < 0 || (n > 0 && (!reqs || !inout_dirs))) return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = nh_validate_field_reqs(ctx, reqs, n);
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
    for(int i = 0; i < n; i++)      // skipped (IF_CHANGED) slots must come back unchanged too
        any_inout |= (reqs[i].flags & (NAVHIP_REQ_INOUT | NAVHIP_REQ_IF_CHANGED | NAVHIP_REQ_ISLAND_NEAREST)) != 0
                  || reqs[i].type == NAVHIP_TARGET_NEAREST_PATHABLE;
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


// ---------------------------------------------------------------------------------------------
// agent step
// ---------------------------------------------------------------------------------------------
static int ensure_buf(navhip_ctx *ctx, navhip_ctx::buf &b, size_t need)
{
    return ensure_cap(ctx, &b.p, &b.cap, need ? need : 16);
}

// cohesion scratch: grow on demand; a new buffer or another flock count has no lane grouping yet
static int coh_scratch_ensure(navhip_ctx *ctx, int n_flocks, int n_members, hipStream_t s)
{
    const void *old = ctx->coh_plan.p;
    int rc = ensure_buf(ctx, ctx->coh_plan, nh_cohesion_scratch_bytes(n_flocks, n_members));
    if(rc) return rc;
    if(ctx->coh_plan.p != old || ctx->coh_flocks != n_flocks || ctx->coh_members != n_members) {
        // (scratch may still be in use by a regrouping on a side stream: order behind it)
        if(ctx->aux[1] && s != ctx->aux[1]) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_regroup, ctx->aux[1]));
            HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_regroup, 0));
        }
        HIPCHK(ctx, nh_cohesion_scratch_reset((int32_t*)ctx->coh_plan.p, n_flocks, n_members, s));
        ctx->coh_flocks = n_flocks; ctx->coh_members = n_members; ctx->coh_parity = 0;
    }
    return NAVHIP_OK;
}

// bg_<name>_init geometry, bitmap_grid.h:959-990
static bool grid_geometry(const navhip_world *w, nh_grid *g)
{
    int32_t ox = (int32_t)lrintf(w->grid_xmin * 256.0f), oy = (int32_t)lrintf(w->grid_zmin * 256.0f);
    int32_t span_x = (int32_t)lrintf(w->grid_xmax * 256.0f) - ox;
    int32_t span_y = (int32_t)lrintf(w->grid_zmax * 256.0f) - oy;
    if(span_x <= 0 || span_y <= 0) return false;
    g->origin_x = ox; g->origin_y = oy;
    g->grid_w = (int)(((uint32_t)span_x + 4095u) >> 12);
    g->grid_h = (int)(((uint32_t)span_y + 4095u) >> 12);
    if(g->grid_w < 1) g->grid_w = 1;
    if(g->grid_h < 1) g->grid_h = 1;
    return true;
}

static void fill_map_view(const navhip_ctx *ctx, nh_map_view *mv)
{
    mv->w = ctx->w; mv->h = ctx->h;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        const navhip_layer &L = ctx->layers[l];
        mv->layers[l] = nh_layer_view{L.cost, L.blockers, L.local_islands, L.factions,
                                      L.passmask, L.unit_cost, L.changed, L.islands, L.probemask};
    }
}

// like ensure_buf, but a fresh allocation is zeroed (counters that the kernels keep at zero themselves)
static int ensure_zeroed(navhip_ctx *ctx, navhip_ctx::buf &b, size_t need, hipStream_t s)
{
    const void *old = b.p;
    int rc = ensure_buf(ctx, b, need);
    if(rc) return rc;
    if(b.p != old) HIPCHK(ctx, hipMemsetAsync(b.p, 0, b.cap, s));
    return NAVHIP_OK;
}

static int spatial_build(navhip_ctx *ctx, const navhip_world *w, nh_grid *g, hipStream_t s,
                         int slab_begin = 0, int slab_end = -1, bool with_records = true)
{
    if(!grid_geometry(w, g)) {
        ctx->last_error = "agent step: empty spatial-grid bounds";
        return NAVHIP_ERR_INVALID;
    }
    const size_t n = (size_t)w->n_ents, ncells = (size_t)g->grid_w * g->grid_h;
    // ent_cell, ent_rank, cell_count, cell_start, tmp_id, block_sum, box, recA, recV, pool_of
    const size_t bytes[10] = {4 * n, 4 * n, 4 * ncells, 4 * (ncells + 1), 4 * n, 4 * ((ncells + NH_SCAN_T - 1) / NH_SCAN_T),
                              48, 16 * n, 8 * n, 4 * n};        // ([6]: two slab boxes + the length of the slab's list of walks)
    for(int i = 0; i < 10; i++) {
        const void *old = ctx->sp[i].p;
        int rc = (i == 2) ? ensure_zeroed(ctx, ctx->sp[i], bytes[i], s) : ensure_buf(ctx, ctx->sp[i], bytes[i]);
        if(rc) return rc;
        // the two slab boxes start empty (INT_MIN); afterwards every build re-initialises its successor's
        if(i == 6 && ctx->sp[i].p != old)
            HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->sp[i].p, (int)0x80000000, 8, s));
    }
    nh_spatial_scratch S = {(int32_t*)ctx->sp[0].p, (int32_t*)ctx->sp[1].p, (int32_t*)ctx->sp[2].p,
                            (int32_t*)ctx->sp[3].p, (int32_t*)ctx->sp[4].p, (int32_t*)ctx->sp[5].p,
                            (int32_t*)ctx->sp[6].p, 0, (float4*)ctx->sp[7].p, (float2*)ctx->sp[8].p,
                            (int32_t*)ctx->sp[9].p,
                            {w->vel_xz, w->radius, w->flags, w->state, w->arrival_sink_xz, w->arrival_flags}};
    if(!with_records) S.src = nh_pack_src{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // (positions only)
    g->n = w->n_ents;
    if(slab_end < 0) slab_end = w->n_ents;
    // (the two slab boxes alternate between the builds that USE one: such a build cleans the other)
    if(slab_begin > 0 || slab_end < w->n_ents) S.box_parity = (int)(ctx->sp_builds++ & 1u);
    nh_launch_spatial_build(*g, w->pos_xz, S, slab_begin, slab_end, s);
    return NAVHIP_OK;
}

// scratch of the neighbour walk and of the work lists
static int step_scratch(navhip_ctx *ctx, int n_ents, nh_nbr *NB, nh_worklists *WL, hipStream_t s)
{
    const size_t n = (size_t)n_ents;
    int rc = ensure_buf(ctx, ctx->nbr[0], 8 * n);
    if(!rc) rc = ensure_zeroed(ctx, ctx->nbr[1], 4 * n, s);
    if(!rc) rc = ensure_buf(ctx, ctx->nbr[2], 4 * (size_t)(64 * 5) * n);
    if(!rc) rc = ensure_buf(ctx, ctx->midrec, sizeof(nh_mid_rec) * n);
    const int cap = nh_worklist_cap(n_ents);
    if(!rc) rc = ensure_zeroed(ctx, ctx->wl[0], 4 * 2 * NH_WL_COUNTERS, s);
    if(!rc) rc = ensure_buf(ctx, ctx->wl[1], 4 * (size_t)NH_WL_LISTS * NH_WL_SUB * cap);
    if(rc) return rc;
    NB->sep = (float2*)ctx->nbr[0].p; NB->cnt = (uint32_t*)ctx->nbr[1].p; NB->rec = (float*)ctx->nbr[2].p;
    NB->stride = 64 * 5;
    WL->count = (int32_t*)ctx->wl[0].p; WL->ids = (int32_t*)ctx->wl[1].p; WL->cap = cap;
    return NAVHIP_OK;
}

static int step_check_world(navhip_ctx *ctx, const navhip_world *w)
{
    if(!w || w->n_ents < 0 || (w->hz != 20 && w->hz != 10 && w->hz != 5 && w->hz != 1))
        return NAVHIP_ERR_INVALID;
    if(w->n_ents == 0) return NAVHIP_OK;
    if(w->n_ents >= (1 << 24)) {            // pool records carry the uid in 24 bits
        ctx->last_error = "agent step: more than 2^24 entities";
        return NAVHIP_ERR_INVALID;
    }
    if((w->arrival_flags != nullptr) != (w->arrival_sink_xz != nullptr)) return NAVHIP_ERR_INVALID;
    if(!w->pos_xz || !w->vel_xz || !w->radius || !w->max_speed || !w->speed || !w->flags
    || !w->state || !w->has_dest_los || !w->flock
    || (w->n_flocks > 0 && (!w->flock_target_xz || !w->flock_offsets || !w->flock_members)))
        return NAVHIP_ERR_INVALID;
    if(!ctx->layers[0].cost && !ctx->layers[4].cost && !ctx->layers[8].cost) {
        ctx->last_error = "agent step: no cost_base plane uploaded";
        return NAVHIP_ERR_NOT_UPLOADED;
    }
    return NAVHIP_OK;
}

static int step_fill_params(navhip_ctx *ctx, const navhip_world *w, nh_step_params *Pp)
{
    nh_step_params &P = *Pp;
    memset(&P, 0, sizeof(P));
    int rc_masks = refresh_derived(ctx, ctx->stream);      // the tile probes read the derived row masks
    if(rc_masks) return rc_masks;
    fill_map_view(ctx, &P.map);
    P.map_x = w->map_pos_x; P.map_z = w->map_pos_z;
    P.n_ents = w->n_ents; P.n_flocks = w->n_flocks; P.hz = w->hz;
    P.n_members = w->n_ents;          // every entity belongs to at most one flock
    P.work_begin = w->work_begin; P.work_end = w->work_end;
    if(P.work_begin == 0 && P.work_end == 0) P.work_end = w->n_ents;
    if(P.work_begin < 0 || P.work_end > w->n_ents || P.work_begin > P.work_end)
        return NAVHIP_ERR_INVALID;
    // (the cohesion term's lane grouping, carried from tick to tick: see k_coh_bin)
    if(P.work_begin == 0 && P.work_end == w->n_ents) P.members_key = 0;
    else if(w->static_epoch)                          P.members_key = (int)((w->static_epoch & 0x3fffffffu) | 0x40000000u);
    else                                              P.members_key = (int)(0x80000000u | ++ctx->coh_unique);
    P.pos_xz = w->pos_xz; P.vel_xz = w->vel_xz; P.radius = w->radius; P.max_speed = w->max_speed;
    P.speed = w->speed; P.flags = w->flags; P.state = w->state; P.has_dest_los = w->has_dest_los;
    P.flock = w->flock; P.vdes_xz = w->vdes_xz; P.flock_target_xz = w->flock_target_xz;
    P.flock_offsets = w->flock_offsets; P.flock_members = w->flock_members;
    P.flock_field_slot = w->flock_field_slot; P.field_pool = w->field_pool;
    if(w->n_field_slots == NAVHIP_POOL_RESIDENT) {
        // sample the context's resident pool: row = flock index of the (dest, chunk) -> slot table
        if(!ctx->pool || w->n_flocks > nh_pool_dests(ctx)) {
            ctx->last_error = "agent step: NAVHIP_POOL_RESIDENT without a pool that has a row per flock";
            return NAVHIP_ERR_INVALID;
        }
        P.flock_field_slot = nh_pool_map(ctx); P.field_pool = nh_pool_fields(ctx);
    }
    P.form_ready = w->form_ready; P.cell_pos_xz = w->cell_pos_xz;
    P.form_cohesion_xz = w->form_cohesion_xz; P.form_align_xz = w->form_align_xz;
    P.form_drag_xz = w->form_drag_xz;
    P.arrival_sink_xz = w->arrival_sink_xz; P.arrival_flags = w->arrival_flags;
    P.los_pool = w->los_pool; P.flock_los_slot = w->flock_los_slot; P.los_pos_xz = w->los_pos_xz;
    if((P.los_pool != nullptr) != (P.flock_los_slot != nullptr)) return NAVHIP_ERR_INVALID;
    P.region_row = w->region_row; P.region_field_slot = w->region_field_slot;
    if(P.region_row && w->n_field_slots == NAVHIP_POOL_RESIDENT) {
        // rows of the resident pool's mapping table (the caller keeps its region rows behind the flock rows)
        if(w->n_region_rows > nh_pool_dests(ctx)) {
            ctx->last_error = "agent step: more region rows than the resident pool's mapping table has";
            return NAVHIP_ERR_INVALID;
        }
        P.region_field_slot = nh_pool_map(ctx);
    }else if(P.region_row && !P.region_field_slot) {
        ctx->last_error = "agent step: region_row given without region_field_slot";
        return NAVHIP_ERR_INVALID;
    }
    if(P.form_ready && (!P.cell_pos_xz || !P.form_cohesion_xz || !P.form_align_xz || !P.form_drag_xz)) {
        ctx->last_error = "agent step: form_ready given without the other formation arrays";
        return NAVHIP_ERR_INVALID;
    }
    return NAVHIP_OK;
}

// what a prefetch was started for: the step that follows joins it only for the very same snapshot
static void pre_key_fill(navhip_ctx *ctx, const navhip_world *w, const nh_step_params &P)
{
    auto &k = ctx->pre;
    k.pos_xz = w->pos_xz; k.vel_xz = w->vel_xz; k.radius = w->radius; k.flags = w->flags;
    k.state = w->state; k.flock_members = w->flock_members; k.flock_offsets = w->flock_offsets;
    k.arrival_flags = w->arrival_flags; k.arrival_sink_xz = w->arrival_sink_xz;
    k.n_ents = w->n_ents; k.n_flocks = w->n_flocks; k.hz = w->hz;
    k.work_begin = P.work_begin; k.work_end = P.work_end;
    k.g.origin_x = P.grid.origin_x; k.g.origin_y = P.grid.origin_y;
    k.g.grid_w = P.grid.grid_w; k.g.grid_h = P.grid.grid_h;
}

static bool pre_key_matches(const navhip_ctx *ctx, const navhip_world *w, const nh_step_params &P,
                            const nh_grid &g)
{
    const auto &k = ctx->pre;
    return k.pos_xz == w->pos_xz && k.vel_xz == w->vel_xz && k.radius == w->radius && k.flags == w->flags
        && k.state == w->state && k.flock_members == w->flock_members && k.flock_offsets == w->flock_offsets
        && k.arrival_flags == w->arrival_flags && k.arrival_sink_xz == w->arrival_sink_xz
        && k.n_ents == w->n_ents && k.n_flocks == w->n_flocks && k.hz == w->hz
        && k.work_begin == P.work_begin && k.work_end == P.work_end
        && k.g.origin_x == g.origin_x && k.g.origin_y == g.origin_y && k.g.grid_w == g.grid_w
        && k.g.grid_h == g.grid_h;
}

// the side streams of the agent step (snapshot-only work beside the field builds; the ClearPath launches beside each
// other) and the words in device memory they hand over through (nh_handover).  The streams are the process's (nh_streams_for): borrowed, and chosen for the stream the
// step's main chain runs on -- the ones whose hardware queues sit on other pipes than that stream's.
static int ensure_side_streams(navhip_ctx *ctx, hipStream_t main)
{
    if(ctx->aux[0] && ctx->aux_main == main) return NAVHIP_OK;
    hipStream_t st[NH_STREAM_FIXED];
    int rc = nh_streams_for(ctx, main, st);
    if(rc) return rc;
    if(!ctx->ev_regroup) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_regroup, hipEventDisableTiming));
    if(ctx->aux[0] && (ctx->aux[0] != st[NH_STREAM_SIDE0] || ctx->aux[1] != st[NH_STREAM_SIDE1])) {
        // another caller stream than last time: whatever the old side streams still hold is waited for
        for(auto a : ctx->aux) HIPCHK(ctx, hipStreamSynchronize(a));
        ctx->pre.valid = false; ctx->regroup_pending = false;
    }
    ctx->aux[0] = st[NH_STREAM_SIDE0]; ctx->aux[1] = st[NH_STREAM_SIDE1]; ctx->aux_main = main;
    return nh_handover_ensure(ctx);
}


// The lane regrouping of the cohesion term (five small launches behind k_cohesion) is for the NEXT tick's launch
// and only has to be spatially coherent: agents move about one world unit per tick and a group's box is compared
// against a 904-unit reach, so a grouping serves many ticks.  It is rebuilt on the two ticks after anything it was
// built for changes (entity / flock / member counts, work range, membership key) and every NH_COH_REGROUP_EVERY-th
// tick otherwise; k_cohesion checks on the device that the grouping it is given fits (else: the identity).
#define NH_COH_REGROUP_EVERY 8
// a jam: the list lengths of the last step, in pinned memory without a wait -- 8 192 workgroup searches and more
static bool step_in_a_jam(navhip_ctx *ctx)
{
    int32_t lists[6];
    return navhip_step_lists_peek(ctx, lists) == NAVHIP_OK && lists[4] >= 8192;
}

static bool coh_regroup_due(navhip_ctx *ctx, const nh_step_params &P)
{
    const int64_t key[4] = {((int64_t)P.n_ents << 32) | (uint32_t)P.n_flocks, (int64_t)P.n_members,
                            ((int64_t)P.work_begin << 32) | (uint32_t)P.work_end, (int64_t)P.members_key};
    if(memcmp(key, ctx->coh_regroup_key, sizeof(key)) != 0) {
        memcpy(ctx->coh_regroup_key, key, sizeof(key));
        ctx->coh_regroup_age = 0;
    }
    const int age = ctx->coh_regroup_age++;
    // In a jam -- the list lengths of the last step, in pinned memory without a wait: 8 192 workgroup searches and more --
    // the regrouping stays on every tick: the crowded world measured 3-4 % SLOWER without its five small launches on the
    // side stream although every kernel takes the same time under the tracer (profiles/archive/r04_ab_regroup_cadence.txt; launch
    // timing against the persistent searches, profiles/HISTORY.md 3.7).  Kept as measured.
    const bool jam = step_in_a_jam(ctx);
    // (a slab step whose caller gave no static_epoch carries a never-repeating key: k_cohesion could not accept a
    // grouping made for it -- the five launches would be wasted)
    if(P.members_key < 0) return false;
    return jam || age < 2 || age % NH_COH_REGROUP_EVERY == 0;
}

int navhip_agent_prefetch_dev_ex(navhip_ctx *ctx, const navhip_world *w, void *stream, uint32_t flags)
{
    if(!ctx) return NAVHIP_ERR_INVALID;
    int rc = step_check_world(ctx, w);
    if(rc) return rc;
    ctx->pre.valid = false;
    if(w->n_ents == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // (a prefetch may be issued on another stream than the step it belongs to -- the exchange stream of a sharded tick --:
    // the side streams stay the ones chosen for the step's stream once there has been a step)
    rc = ensure_side_streams(ctx, ctx->aux[0] ? ctx->aux_main : s);
    if(rc) return rc;
    nh_step_params P;
    rc = step_fill_params(ctx, w, &P);
    if(rc) return rc;
    nh_nbr NB; nh_worklists WL;
    rc = ensure_buf(ctx, ctx->coh, (size_t)w->n_ents * 2 * sizeof(float));
    if(!rc) rc = coh_scratch_ensure(ctx, w->n_flocks, P.n_members, s);
    if(!rc) rc = step_scratch(ctx, w->n_ents, &NB, &WL, s);
    if(rc) return rc;
    // the front of the step (spatial hash -> neighbour walk) is a chain of small launches on the
    // critical path of the tick: NAVHIP_PREFETCH_FRONT_INLINE keeps it on the caller's stream, where it
    // follows the previous step without a cross-stream hand-over (tens of microseconds each)
    hipStream_t front = (flags & NAVHIP_PREFETCH_FRONT_INLINE) ? s : ctx->aux[0];
    // The side streams start behind a word in device memory (stream_set.hip: 2-3 us from the store to the waiting stream's
    // next kernel, against 12 us and a packet on the caller's stream for an event).  NAVHIP_PREFETCH_FOLLOWS_STEP: the
    // last step on this stream stored that word when it ended, and the snapshot was final then -- nothing goes in front
    // of the front at all.  Otherwise a one-lane launch stores it now, in FRONT of the first kernel of the front: the
    // cohesion kernel ends last, so it must not start late (profiles/archive/r03_ab_fork_first.txt).
    nh_handover_mode(ctx, step_in_a_jam(ctx));           // (words, or events: under a serialising profiler, in a jam)
    const bool follows = (flags & NAVHIP_PREFETCH_FOLLOWS_STEP) && ctx->step_end_on == s && !ctx->ho->by_events;
    if(!follows) nh_handover_signal(ctx, NH_HO_START, s);
    ctx->start_flag = follows ? NH_HO_END : NH_HO_START;
    ctx->start_seq = nh_handover_seq(ctx, ctx->start_flag);
    if(front != s) nh_handover_wait(ctx, ctx->start_flag, ctx->aux[0]);
    nh_handover_wait(ctx, ctx->start_flag, ctx->aux[1]);
    // side stream 1: cohesion -- enqueued first: it ends last, and a host that is not ahead of the device (the tick after a
    // synchronisation) would otherwise hold it back by the front's six launches
    const bool regroup = nh_launch_cohesion(P, (int32_t*)ctx->coh_plan.p, (float*)ctx->coh.p, &ctx->coh_parity,
                                            ctx->aux[1]);
    nh_handover_signal(ctx, NH_HO_COH, ctx->aux[1]);
    // the front: spatial hash -> neighbour walk (separation force + ClearPath neighbour lists)
    rc = spatial_build(ctx, w, &P.grid, front, P.work_begin, P.work_end);
    if(rc) return rc;
    nh_launch_agent_nbr(P, NB, front);
    // (an inline front is ordered on the caller's stream by itself: that it is done is stored by the step's own wait
    // for the cohesion term, or by a launch of its own when somebody asks before -- navhip_stream_wait_stage)
    ctx->join0_signalled = false;
    if(front != s) { nh_handover_signal(ctx, NH_HO_NBR, front); ctx->join0_signalled = true; }
    ctx->front_stream = front;
    ctx->snapshot_held = (flags & NAVHIP_PREFETCH_SNAPSHOT_HELD) != 0;
    // (behind the cohesion term's word: the agent step does not wait for next tick's lane grouping; but the
    // caller's stream does, at the end of navhip_agent_step_dev, so that whatever the caller does
    // to the snapshot arrays afterwards is ordered behind the last read of them)
    ctx->regroup_pending = false;
    if(regroup && coh_regroup_due(ctx, P)) {
        nh_launch_cohesion_regroup(P, (int32_t*)ctx->coh_plan.p, &ctx->coh_parity, ctx->aux[1]);
        HIPCHK(ctx, hipEventRecord(ctx->ev_regroup, ctx->aux[1]));
        ctx->regroup_pending = true;
    }
    HIPCHK(ctx, hipGetLastError());
    ctx->pre.valid = true;
    pre_key_fill(ctx, w, P);
    return NAVHIP_OK;
}

int navhip_agent_prefetch_dev(navhip_ctx *ctx, const navhip_world *w, void *stream)
{
    return navhip_agent_prefetch_dev_ex(ctx, w, stream, 0);
}

// behind a step: its list counters on their way to pinned host memory (side stream, after the searches)
static int send_step_lists(navhip_ctx *ctx, int parity_used, hipStream_t fallback, bool on_fallback = false)
{
    if(!ctx->lists_pinned) {
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->lists_pinned, sizeof(int32_t) * NH_WL_LISTS * NH_WL_SUB, hipHostMallocDefault));
        memset(ctx->lists_pinned, 0, sizeof(int32_t) * NH_WL_LISTS * NH_WL_SUB);
    }
    const int32_t *src = (const int32_t*)ctx->wl[0].p + parity_used * NH_WL_COUNTERS;
    HIPCHK(ctx, hipMemcpyAsync(ctx->lists_pinned, src, sizeof(int32_t) * NH_WL_LISTS * NH_WL_SUB, hipMemcpyDeviceToHost,
                               (ctx->aux[0] && !on_fallback) ? ctx->aux[0] : fallback));
    return NAVHIP_OK;
}

int navhip_agent_step_dev(navhip_ctx *ctx, const navhip_world *w, const navhip_step_out *out,
                          void *stream)
{
    if(!ctx || !out || !out->vel_xz) return NAVHIP_ERR_INVALID;
    int rc = step_check_world(ctx, w);
    if(rc) return rc;
    if(w->n_ents == 0) return NAVHIP_OK;
    if(nh_handover_failed(ctx)) return NAVHIP_ERR_DEVICE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    ctx->step_end_on = nullptr; ctx->step_end_signalled = false;

    nh_step_params P;
    rc = step_fill_params(ctx, w, &P);
    if(rc) return rc;
    if(!grid_geometry(w, &P.grid)) {
        ctx->last_error = "agent step: empty spatial-grid bounds";
        return NAVHIP_ERR_INVALID;
    }
    ctx->counters.step_calls++; ctx->counters.agent_steps += (uint64_t)(P.work_end - P.work_begin);
    const bool prof = ctx->profiling;
    const bool joined = ctx->pre.valid && !prof && !ctx->serial_step && pre_key_matches(ctx, w, P, P.grid);
    if(ctx->pre.valid && !joined) {
        // a prefetch for another snapshot is in flight on the side streams: let it drain before
        // its scratch buffers are reused
        if(!ctx->join0_signalled) { nh_handover_signal(ctx, NH_HO_NBR, ctx->front_stream); ctx->join0_signalled = true; }
        nh_handover_wait2(ctx, NH_HO_NBR, NH_HO_COH, s);
        if(ctx->regroup_pending) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_regroup, 0));
        ctx->regroup_pending = false;
    }
    ctx->pre.valid = false;
    nh_step_outs O = {out->vel_xz, out->new_pos_xz, out->vdes_xz, out->vpref_xz, out->status};
    nh_nbr NB; nh_worklists WL;
    rc = step_scratch(ctx, w->n_ents, &NB, &WL, s);
    if(!rc) rc = ensure_side_streams(ctx, s);
    if(rc) return rc;
    if(!joined) nh_handover_mode(ctx, step_in_a_jam(ctx));       // (a joined step keeps the mode its prefetch chose)
    if(joined) {
        // spatial hash + neighbour walk + cohesion were started by navhip_agent_prefetch_dev: join
        P.grid.n = w->n_ents;
        P.grid.cell_start = (int32_t*)ctx->sp[3].p; P.grid.recA = (const float4*)ctx->sp[7].p;
        P.grid.recV = (const float2*)ctx->sp[8].p; P.grid.pool_of = (const int32_t*)ctx->sp[9].p;
        P.grid.active = nullptr; P.grid.n_active = nullptr;       // (the walk, their only reader, ran with the prefetch)
        // ONE launch on this stream waits for the cohesion term -- and for the front, when that ran elsewhere (the step is
        // issued on another stream than the prefetch); behind an inline front it follows the neighbour walk, and says so
        if(ctx->front_stream != s) {
            if(!ctx->join0_signalled) { nh_handover_signal(ctx, NH_HO_NBR, ctx->front_stream); ctx->join0_signalled = true; }
            nh_handover_wait2(ctx, NH_HO_NBR, NH_HO_COH, s);
        }else{
            nh_handover_wait(ctx, NH_HO_COH, s, ctx->join0_signalled ? -1 : NH_HO_NBR);
            ctx->join0_signalled = true;
        }
        if(nh_launch_agent_finish(P, NB, (float*)ctx->coh.p, (nh_mid_rec*)ctx->midrec.p, WL, ctx->wl_parity, O, s,
                                  ctx->aux[0], ctx)) {
            rc = send_step_lists(ctx, ctx->wl_parity, s);
            ctx->wl_parity ^= 1;
            if(rc) return rc;
        }
        if(ctx->regroup_pending && !ctx->snapshot_held) {
            HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_regroup, 0));     // long finished by now
            ctx->regroup_pending = false;
        }
        HIPCHK(ctx, hipGetLastError());
        return NAVHIP_OK;
    }
    if(prof) {
        for(auto &e : ctx->ev) if(!e) HIPCHK(ctx, hipEventCreate(&e));
        HIPCHK(ctx, hipEventRecord(ctx->ev[0], s));
    }
    rc = spatial_build(ctx, w, &P.grid, s, P.work_begin, P.work_end);
    if(rc) return rc;
    if(prof) HIPCHK(ctx, hipEventRecord(ctx->ev[1], s));
    nh_launch_agent_nbr(P, NB, s);
    if(prof) HIPCHK(ctx, hipEventRecord(ctx->ev[2], s));
    rc = ensure_buf(ctx, ctx->coh, (size_t)w->n_ents * 2 * sizeof(float));
    if(!rc) rc = coh_scratch_ensure(ctx, w->n_flocks, P.n_members, s);
    if(rc) return rc;
    const bool regroup = nh_launch_cohesion(P, (int32_t*)ctx->coh_plan.p, (float*)ctx->coh.p, &ctx->coh_parity, s);
    if(prof) HIPCHK(ctx, hipEventRecord(ctx->ev[3], s));
    if(regroup && coh_regroup_due(ctx, P)) nh_launch_cohesion_regroup(P, (int32_t*)ctx->coh_plan.p, &ctx->coh_parity, s);
    if(prof) HIPCHK(ctx, hipEventRecord(ctx->ev[4], s));
    const bool serial = ctx->serial_step;
    if(nh_launch_agent_finish(P, NB, (float*)ctx->coh.p, (nh_mid_rec*)ctx->midrec.p, WL, ctx->wl_parity, O, s,
                              serial ? nullptr : ctx->aux[0], ctx)) {
        rc = send_step_lists(ctx, ctx->wl_parity, s, serial);
        ctx->wl_parity ^= 1;
        if(rc) return rc;
    }
    if(prof) { HIPCHK(ctx, hipEventRecord(ctx->ev[5], s)); ctx->ev_valid = true; }
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_stream_wait_stage(navhip_ctx *ctx, void *stream, int stage)
{
    if(!ctx || !stream || !ctx->aux[0]) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if(stage == NAVHIP_STAGE_NEIGHBOURS) {
        if(!ctx->front_stream) return NAVHIP_ERR_INVALID;
        // (between the prefetch and its step: a launch of its own behind the walk; after the step: the step's wait for
        // the cohesion term has stored it)
        if(!ctx->join0_signalled) { nh_handover_signal(ctx, NH_HO_NBR, ctx->front_stream); ctx->join0_signalled = true; }
        nh_handover_wait(ctx, NH_HO_NBR, (hipStream_t)stream);
    }else if(stage == NAVHIP_STAGE_START) {
        if(!ctx->front_stream) return NAVHIP_ERR_INVALID;
        nh_handover_wait_for(ctx, ctx->start_flag, ctx->start_seq, (hipStream_t)stream);  // (not the end of a step enqueued since)
    }else if(stage == NAVHIP_STAGE_END) {
        if(!ctx->step_end_signalled) return NAVHIP_ERR_INVALID;  // (the last step ran on one stream: its stream is its end)
        nh_handover_wait(ctx, NH_HO_END, (hipStream_t)stream);
    }else if(stage == NAVHIP_STAGE_LISTS) {
        if(ctx->lists_signalled) nh_handover_wait(ctx, NH_HO_MID, (hipStream_t)stream);      // (else: one stream, nothing to wait for)
    }
    else return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_get_counters(navhip_ctx *ctx, navhip_counters *out, int reset)
{
    if(!ctx || !out) return NAVHIP_ERR_INVALID;
    *out = ctx->counters;
    if(reset) memset(&ctx->counters, 0, sizeof(ctx->counters));
    return NAVHIP_OK;
}

int navhip_set_profiling(navhip_ctx *ctx, int on)
{
    if(!ctx) return NAVHIP_ERR_INVALID;
    ctx->profiling = on != 0;
    ctx->ev_valid = false;
    return NAVHIP_OK;
}

int navhip_last_step_ms(navhip_ctx *ctx, float out_ms[NAVHIP_STEP_PHASES])
{
    if(!ctx || !out_ms || !ctx->ev_valid) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[5]));
    for(int i = 0; i < NAVHIP_STEP_PHASES; i++)
        HIPCHK(ctx, hipEventElapsedTime(&out_ms[i], ctx->ev[i], ctx->ev[i + 1]));
    return NAVHIP_OK;
}

// the work-list sizes of the last agent step: {light 1..4, wave, full} (waits for the step)
int navhip_last_step_lists(navhip_ctx *ctx, int32_t out_counts[6])
{
    if(!ctx || !out_counts || !ctx->wl[0].p) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    const int32_t *src = (const int32_t*)ctx->wl[0].p + (ctx->wl_parity ^ 1) * NH_WL_COUNTERS;
    int32_t h[NH_WL_COUNTERS];
    HIPCHK(ctx, hipMemcpy(h, src, sizeof(h), hipMemcpyDeviceToHost));
    // (the wave and the heavy list are reported together: 17-64 neighbours)
    static const int slot_of[NH_WL_LISTS] = {0, 1, 2, 3, 4, 4, 5, -1};     // (the retry list is not reported)
    for(int l = 0; l < 6; l++) out_counts[l] = 0;
    for(int l = 0; l < NH_WL_LISTS; l++)
        for(int sb = 0; sb < NH_WL_SUB; sb++) if(slot_of[l] >= 0) out_counts[slot_of[l]] += h[l * NH_WL_SUB + sb];
    return NAVHIP_OK;
}

int navhip_step_lists_peek(navhip_ctx *ctx, int32_t out_counts[6])
{
    if(!ctx || !out_counts) return NAVHIP_ERR_INVALID;
    static const int slot_of[NH_WL_LISTS] = {0, 1, 2, 3, 4, 4, 5, -1};
    for(int l = 0; l < 6; l++) out_counts[l] = 0;
    if(!ctx->lists_pinned) return NAVHIP_OK;
    const volatile int32_t *h = ctx->lists_pinned;
    for(int l = 0; l < NH_WL_LISTS; l++)
        for(int sb = 0; sb < NH_WL_SUB; sb++) if(slot_of[l] >= 0) out_counts[slot_of[l]] += h[l * NH_WL_SUB + sb];
    return NAVHIP_OK;
}

// copy a host array to a staging buffer; returns device pointer through *dst (NULL stays NULL)
static int stage_in(navhip_ctx *ctx, int slot, const void *host, size_t bytes, const void **dst,
                    hipStream_t s)
{
    nh_async_invalidate_static(ctx);
    *dst = nullptr;
    if(!host) return NAVHIP_OK;
    int rc = ensure_buf(ctx, ctx->stage[slot], bytes);
    if(rc) return rc;
    if(bytes) HIPCHK(ctx, hipMemcpyAsync(ctx->stage[slot].p, host, bytes, hipMemcpyHostToDevice, s));
    *dst = ctx->stage[slot].p;
    return NAVHIP_OK;
}

static int stage_world(navhip_ctx *ctx, const navhip_world *w, navhip_world *d, hipStream_t s)
{
    *d = *w;
    const size_t n = (size_t)w->n_ents, F = (size_t)w->n_flocks;
    size_t nmembers = 0;
    if(F > 0 && w->flock_offsets) nmembers = (size_t)w->flock_offsets[F];
    const size_t nchunks = (size_t)ctx->nchunks;
    int rc = 0;
#define ST(slot, field, bytes) \
    if(!rc) rc = stage_in(ctx, slot, w->field, (bytes), (const void**)&d->field, s)
    ST(0, pos_xz, n * 8);        ST(1, vel_xz, n * 8);       ST(2, radius, n * 4);
    ST(3, max_speed, n * 4);     ST(4, speed, n * 4);        ST(5, flags, n * 4);
    ST(6, state, n);             ST(7, has_dest_los, n);     ST(8, flock, n * 4);
    ST(9, vdes_xz, n * 8);       ST(10, flock_target_xz, F * 8);
    ST(11, flock_offsets, (F + 1) * 4);                      ST(12, flock_members, nmembers * 4);
    if(w->n_field_slots != NAVHIP_POOL_RESIDENT) {
        ST(13, flock_field_slot, F * nchunks * 4);
        ST(14, field_pool, (size_t)w->n_field_slots * NH_CELLS);
    }
    ST(24, form_ready, n);       ST(25, cell_pos_xz, n * 8); ST(26, form_cohesion_xz, n * 8);
    ST(27, form_align_xz, n * 8); ST(28, form_drag_xz, n * 8);
    ST(36, arrival_sink_xz, n * 8); ST(37, arrival_flags, n);
    ST(38, los_pool, (size_t)(w->n_los_slots > 0 ? w->n_los_slots : 0) * NH_CELLS);
    ST(39, flock_los_slot, F * nchunks * 4); ST(40, los_pos_xz, n * 8);
    ST(46, region_row, n * 4);
    if(w->n_field_slots != NAVHIP_POOL_RESIDENT)
        ST(47, region_field_slot, (size_t)(w->n_region_rows > 0 ? w->n_region_rows : 0) * nchunks * 4);
#undef ST
    return rc;
}

int navhip_agent_step(navhip_ctx *ctx, const navhip_world *w, const navhip_step_out *out)
{
    if(!ctx || !w || !out || !out->vel_xz) return NAVHIP_ERR_INVALID;
    if(w->n_ents <= 0) return w->n_ents == 0 ? NAVHIP_OK : NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    navhip_world d;
    int rc = stage_world(ctx, w, &d, s);
    if(rc) return rc;
    const size_t n = (size_t)w->n_ents;
    navhip_step_out dout = {nullptr, nullptr, nullptr, nullptr, nullptr};
    struct { void **dev; void *host; size_t bytes; int slot; } outs[5] = {
        {(void**)&dout.vel_xz,     out->vel_xz,     n * 8, 15},
        {(void**)&dout.new_pos_xz, out->new_pos_xz, n * 8, 16},
        {(void**)&dout.vdes_xz,    out->vdes_xz,    n * 8, 17},
        {(void**)&dout.vpref_xz,   out->vpref_xz,   n * 8, 18},
        {(void**)&dout.status,     out->status,     n,     19}};
    for(auto &o : outs) {
        if(!o.host) continue;
        rc = ensure_buf(ctx, ctx->stage[o.slot], o.bytes);
        if(rc) return rc;
        *o.dev = ctx->stage[o.slot].p;
    }
    rc = navhip_agent_step_dev(ctx, &d, &dout, s);
    if(rc) return rc;
    // only the rows of the stepped slab were written: a caller that issues one call per slab into
    // the same output arrays (move_submit_cpu_work, movement.c:3759-3762) keeps its other slabs
    size_t b = (size_t)w->work_begin, e = (size_t)w->work_end;
    if(b == 0 && e == 0) e = n;
    for(auto &o : outs) {
        if(!o.host || e <= b) continue;
        const size_t row = o.bytes / n;
        HIPCHK(ctx, hipMemcpyAsync((char*)o.host + b * row, (char*)*o.dev + b * row, (e - b) * row,
                                   hipMemcpyDeviceToHost, s));
    }
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

int navhip_state_update_dev(navhip_ctx *ctx, const navhip_world *w, const navhip_state_in *in,
                            uint8_t *out_state, uint8_t *out_flags, void *stream)
{
    if(!ctx || !w || !in || !out_state || !out_flags) return NAVHIP_ERR_INVALID;
    if(w->n_ents < 0) return NAVHIP_ERR_INVALID;
    if(w->n_ents == 0) return NAVHIP_OK;
    if(!w->pos_xz || !w->radius || !w->flags || !w->state || !w->flock || !in->new_pos_xz || !in->vdes_xz
    || (w->n_flocks > 0 && (!w->flock_target_xz || !w->flock_offsets || !w->flock_members || !in->flock_layer
                            || !in->flock_nearest_xz || !in->flock_tiles_off || !in->flock_tiles)))
        return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    nh_step_params P;
    memset(&P, 0, sizeof(P));
    int rc_masks = refresh_derived(ctx, ctx->stream);
    if(rc_masks) return rc_masks;
    fill_map_view(ctx, &P.map);
    P.map_x = w->map_pos_x; P.map_z = w->map_pos_z;
    P.n_ents = w->n_ents; P.n_flocks = w->n_flocks; P.hz = w->hz;
    P.work_begin = w->work_begin; P.work_end = w->work_end;
    if(P.work_begin == 0 && P.work_end == 0) P.work_end = w->n_ents;
    if(P.work_begin < 0 || P.work_end > w->n_ents || P.work_begin > P.work_end) return NAVHIP_ERR_INVALID;
    P.pos_xz = w->pos_xz; P.radius = w->radius; P.flags = w->flags; P.state = w->state; P.flock = w->flock;
    P.flock_target_xz = w->flock_target_xz; P.flock_offsets = w->flock_offsets; P.flock_members = w->flock_members;
    // (scratch of the arrived-flock-mate rule: the ARRIVED members of every flock, compacted)
    const size_t nmem = (size_t)w->n_ents;           // every entity belongs to at most one flock
    int rc = ensure_buf(ctx, ctx->arrived[0], 16 * nmem);
    if(!rc) rc = ensure_buf(ctx, ctx->arrived[1], 4 * (size_t)(w->n_flocks > 0 ? w->n_flocks : 1));
    if(rc) return rc;
    nh_launch_state_update(P, *in, (float4*)ctx->arrived[0].p, (int32_t*)ctx->arrived[1].p, out_state, out_flags,
                           stream ? (hipStream_t)stream : ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_state_update(navhip_ctx *ctx, const navhip_world *w, const navhip_state_in *in,
                        uint8_t *out_state, uint8_t *out_flags)
{
    if(!ctx || !w || !in || !out_state || !out_flags) return NAVHIP_ERR_INVALID;
    if(w->n_ents <= 0) return w->n_ents == 0 ? NAVHIP_OK : NAVHIP_ERR_INVALID;
    if(w->n_flocks > 0 && (!w->flock_offsets || !in->flock_tiles_off)) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)w->n_ents, F = (size_t)w->n_flocks;
    const size_t nmembers = F ? (size_t)w->flock_offsets[F] : 0, ntiles = F ? (size_t)in->flock_tiles_off[F] : 0;
    if(nmembers > n) {                 // (an entity belongs to at most one flock: the scratch of the arrived-mate rule is sized by it)
        ctx->last_error = "navhip_state_update: more flock members than entities";
        return NAVHIP_ERR_INVALID;
    }
    navhip_world d = *w;
    navhip_state_in di = *in;
    int rc = 0;
#define ST(slot, src, dst, bytes) if(!rc) rc = stage_in(ctx, slot, src, (bytes), (const void**)&dst, s)
    ST(0, w->pos_xz, d.pos_xz, n * 8);       ST(2, w->radius, d.radius, n * 4);   ST(5, w->flags, d.flags, n * 4);
    ST(6, w->state, d.state, n);             ST(8, w->flock, d.flock, n * 4);
    ST(10, w->flock_target_xz, d.flock_target_xz, F * 8);
    ST(11, w->flock_offsets, d.flock_offsets, (F + 1) * 4);
    ST(12, w->flock_members, d.flock_members, nmembers * 4);
    ST(16, in->new_pos_xz, di.new_pos_xz, n * 8);   ST(17, in->vdes_xz, di.vdes_xz, n * 8);
    ST(7, in->skip, di.skip, n);
    ST(33, in->flock_layer, di.flock_layer, F);     ST(34, in->flock_nearest_xz, di.flock_nearest_xz, F * 8);
    ST(35, in->flock_tiles_off, di.flock_tiles_off, (F + 1) * 4);
    ST(41, in->flock_tiles, di.flock_tiles, ntiles * 4);
#undef ST
    if(!rc && F > 0 && ntiles == 0) {
        // (no destination has island tiles: the kernel still wants a pointer -- nothing is read from it)
        rc = ensure_buf(ctx, ctx->stage[41], 4);
        di.flock_tiles = (const int16_t*)ctx->stage[41].p;
    }
    if(!rc) rc = ensure_buf(ctx, ctx->stage[32], 2 * n);
    if(rc) return rc;
    nh_async_invalidate_static(ctx);
    uint8_t *d_out = (uint8_t*)ctx->stage[32].p;
    rc = navhip_state_update_dev(ctx, &d, &di, d_out, d_out + n, s);
    if(rc) return rc;
    size_t b = (size_t)w->work_begin, e = (size_t)w->work_end;
    if(b == 0 && e == 0) e = n;
    if(e > b) {
        HIPCHK(ctx, hipMemcpyAsync(out_state + b, d_out + b, e - b, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipMemcpyAsync(out_flags + b, d_out + n + b, e - b, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

int navhip_region_lookup(navhip_ctx *ctx, int nq, const float *pos_xz, const int32_t *rows,
                         const int32_t *region_field_slot, int n_region_rows, const uint8_t *field_pool,
                         int n_field_slots, const int32_t *centre_abs, const int32_t *radius,
                         float map_pos_x, float map_pos_z, uint8_t *out_dir, uint8_t *out_at_slot)
{
    if(!ctx || nq < 0 || (nq > 0 && (!pos_xz || !rows || !out_dir))) return NAVHIP_ERR_INVALID;
    if((out_at_slot != nullptr) && (!centre_abs || !radius)) return NAVHIP_ERR_INVALID;
    if(nq == 0) return NAVHIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    nh_step_params P;
    memset(&P, 0, sizeof(P));
    fill_map_view(ctx, &P.map);
    P.map_x = map_pos_x; P.map_z = map_pos_z;
    const size_t nchunks = (size_t)ctx->nchunks;
    const bool resident = !region_field_slot && !field_pool;
    if(resident) {
        if(!ctx->pool) { ctx->last_error = "navhip_region_lookup: no table given and no resident pool"; return NAVHIP_ERR_INVALID; }
        P.region_field_slot = nh_pool_map(ctx); P.field_pool = nh_pool_fields(ctx);
        n_region_rows = nh_pool_dests(ctx);
    }else{
        if(!region_field_slot || !field_pool || n_region_rows < 1 || n_field_slots < 1) return NAVHIP_ERR_INVALID;
        int rc = stage_in(ctx, 47, region_field_slot, (size_t)n_region_rows * nchunks * 4, (const void**)&P.region_field_slot, s);
        if(!rc) rc = stage_in(ctx, 14, field_pool, (size_t)n_field_slots * NH_CELLS, (const void**)&P.field_pool, s);
        if(rc) return rc;
        nh_async_invalidate_static(ctx);
    }
    for(int q = 0; q < nq; q++)
        if(rows[q] < -1 || rows[q] >= n_region_rows) return NAVHIP_ERR_INVALID;
    const float *d_pos; const int32_t *d_rows, *d_cen = nullptr, *d_rad = nullptr;
    int rc = stage_in(ctx, 20, pos_xz, (size_t)nq * 8, (const void**)&d_pos, s);
    if(!rc) rc = stage_in(ctx, 21, rows, (size_t)nq * 4, (const void**)&d_rows, s);
    if(!rc && out_at_slot) rc = stage_in(ctx, 22, centre_abs, (size_t)nq * 8, (const void**)&d_cen, s);
    if(!rc && out_at_slot) rc = stage_in(ctx, 23, radius, (size_t)nq * 4, (const void**)&d_rad, s);
    if(!rc) rc = ensure_buf(ctx, ctx->stage[32], (size_t)nq * 2);
    if(rc) return rc;
    uint8_t *d_out = (uint8_t*)ctx->stage[32].p;
    nh_launch_region_lookup(P, nq, d_pos, d_rows, d_cen, d_rad, d_out, d_out + nq, s);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out_dir, d_out, (size_t)nq, hipMemcpyDeviceToHost, s));
    if(out_at_slot) HIPCHK(ctx, hipMemcpyAsync(out_at_slot, d_out + nq, (size_t)nq, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

}  // extern "C"

int nh_refresh_derived(navhip_ctx *ctx, hipStream_t s) { return refresh_derived(ctx, s); }
int nh_prepare_step_streams(navhip_ctx *ctx, hipStream_t main) { return ensure_side_streams(ctx, main); }

// bg_ent insert-all + inrange_circle for nq query points with everything on the device: the index over
// dev_w->pos_xz (positions only), ids in the reference's visiting order into d_ids [nq][maxout], counts into d_counts
int nh_spatial_query_dev(navhip_ctx *ctx, const navhip_world *dev_w, const float *d_query, int nq, float range, int maxout,
                         int32_t *d_counts, uint32_t *d_ids, hipStream_t s)
{
    nh_grid g;
    int rc = spatial_build(ctx, dev_w, &g, s, 0, -1, false);
    if(rc) return rc;
    nh_launch_spatial_query(g, d_query, nq, range, maxout, d_counts, d_ids, s);
    HIPCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

extern "C" {

int navhip_spatial_query(navhip_ctx *ctx, const navhip_world *w, const float *query_xz, int nq,
                         float range, int maxout, int32_t *out_counts, uint32_t *out_ids)
{
    if(!ctx || !w || !w->pos_xz || w->n_ents < 0 || nq < 0 || maxout < 1 || !query_xz
    || !out_counts || !out_ids)
        return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    navhip_world d = *w;
    int rc = stage_in(ctx, 0, w->pos_xz, (size_t)w->n_ents * 8, (const void**)&d.pos_xz, s);
    if(rc) return rc;
    const float *dq; 
    rc = stage_in(ctx, 20, query_xz, (size_t)nq * 8, (const void**)&dq, s);
    if(rc) return rc;
    rc = ensure_buf(ctx, ctx->stage[21], (size_t)nq * 4);
    if(rc) return rc;
    rc = ensure_buf(ctx, ctx->stage[22], (size_t)nq * maxout * 4);
    if(rc) return rc;
    rc = nh_spatial_query_dev(ctx, &d, dq, nq, range, maxout, (int32_t*)ctx->stage[21].p, (uint32_t*)ctx->stage[22].p, s);
    if(rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_counts, ctx->stage[21].p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(out_ids, ctx->stage[22].p, (size_t)nq * maxout * 4,
                               hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

static int clearpath_batch(navhip_ctx *ctx, int nq, const float *ent, const float *des_v,
                           const float *dyn, const int32_t *n_dyn, const float *stat,
                           const int32_t *n_stat, float *out, int rows)
{
    if(!ctx || nq < 0 || !ent || !des_v || !dyn || !n_dyn || !stat || !n_stat || !out)
        return NAVHIP_ERR_INVALID;
    if(nq == 0) return NAVHIP_OK;
    for(int i = 0; i < nq; i++) {
        if(n_dyn[i] < 0 || n_dyn[i] > 32 || n_stat[i] < 0 || n_stat[i] > 32) return NAVHIP_ERR_INVALID;
        if(rows == 1 && n_dyn[i] + n_stat[i] > NH_ROW_MAX) return NAVHIP_ERR_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const void *d[6];
    const void *h[6] = {ent, des_v, dyn, n_dyn, stat, n_stat};
    const size_t b[6] = {(size_t)nq * 20, (size_t)nq * 8, (size_t)nq * 640, (size_t)nq * 4,
                         (size_t)nq * 640, (size_t)nq * 4};
    for(int i = 0; i < 6; i++) {
        int rc = stage_in(ctx, i, h[i], b[i], &d[i], s);
        if(rc) return rc;
    }
    int rc = ensure_buf(ctx, ctx->stage[15], (size_t)nq * 8);
    if(rc) return rc;
    nh_launch_clearpath(nq, (const float*)d[0], (const float*)d[1], (const float*)d[2],
                        (const int32_t*)d[3], (const float*)d[4], (const int32_t*)d[5],
                        (float*)ctx->stage[15].p, rows, s);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->stage[15].p, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

int navhip_clearpath(navhip_ctx *ctx, int nq, const float *ent, const float *des_v,
                     const float *dyn, const int32_t *n_dyn, const float *stat,
                     const int32_t *n_stat, float *out)
{
    return clearpath_batch(ctx, nq, ent, des_v, dyn, n_dyn, stat, n_stat, out, 0);
}

int navhip_clearpath_rows(navhip_ctx *ctx, int nq, const float *ent, const float *des_v,
                          const float *dyn, const int32_t *n_dyn, const float *stat,
                          const int32_t *n_stat, float *out)
{
    return clearpath_batch(ctx, nq, ent, des_v, dyn, n_dyn, stat, n_stat, out, 1);
}

int navhip_clearpath_team(navhip_ctx *ctx, int nq, const float *ent, const float *des_v,
                          const float *dyn, const int32_t *n_dyn, const float *stat,
                          const int32_t *n_stat, float *out)
{
    return clearpath_batch(ctx, nq, ent, des_v, dyn, n_dyn, stat, n_stat, out, 2);
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

uint64_t navhip_region_field_id(int kind, int layer, int chunk_r, int chunk_c, uint32_t a, int b, int c)
{
    // N_FlowFieldID, field.c:1976-2003
    const uint64_t head = (((uint64_t)layer) << 60) | (((uint64_t)kind) << 56);
    const uint64_t tail = (((uint64_t)chunk_r) << 8) | ((uint64_t)chunk_c);
    if(kind == NAVHIP_FFID_ENEMIES || kind == NAVHIP_FFID_ENTITY)
        return head | (((uint64_t)a) << 24) | tail;
    if(kind == NAVHIP_FFID_ZONE) {
        const uint32_t cen_chunk_r = a / 64u, cen_tile_r = a % 64u;
        const uint32_t cen_chunk_c = (uint32_t)b / 64u, cen_tile_c = (uint32_t)b % 64u;
        return head | (((uint64_t)(c & 0xff)) << 44) | (((uint64_t)(cen_tile_c & 0x3f)) << 38)
             | (((uint64_t)(cen_tile_r & 0x3f)) << 32) | (((uint64_t)(cen_chunk_c & 0xff)) << 24)
             | (((uint64_t)(cen_chunk_r & 0xff)) << 16) | tail;
    }
    return 0;
}

} // extern "C"
