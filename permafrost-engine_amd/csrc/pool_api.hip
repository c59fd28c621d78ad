// pool_api.hip -- the resident flow-field pool and the asynchronous host-buffer agent step.
//
// What a C host (the reference is C99, it has no device pointers) needs so that the host-buffer entry
// points do not move the whole field cache over PCIe on every call:
//   * a field pool that STAYS in HBM, keyed by the reference's own 64-bit flow-field ids
//     (N_FlowFieldID, field.c:1952), with the same put / contains / dest-mapping operations as the
//     reference's field cache (N_FC_PutFlowField, N_FC_ContainsFlowField, N_FC_PutDestFFMapping,
//     fieldcache.c); batched builds write straight into pool slots;
//   * submit / poll for the per-tick velocity step, staged through pinned memory, so that the nav
//     task can yield between submit and join like the GL path does (movement.c:4212-4233).
#include "navhip_internal.h"
#include <mutex>
#include "agent_internal.h"

#include <algorithm>
#include <cstring>
#include <list>
#include <unordered_map>
#include <vector>

#define HIPCHK(ctx, expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if(_e != hipSuccess) {                                                              \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);          \
            return NAVHIP_ERR_DEVICE;                                                       \
        }                                                                                   \
    } while(0)

struct nh_pool {
    int       n_slots, n_dests, nchunks;
    uint8_t  *d_fields;                 // [n_slots][4096]
    int32_t  *d_map;                    // [n_dests][nchunks] slot of the (dest, chunk) field, -1 = none
    std::vector<int32_t>  h_map;
    std::vector<uint64_t> id_of;        // per slot (valid when used[slot])
    std::vector<uint8_t>  used;
    std::unordered_map<uint64_t, int> slot_of;
    std::list<int> lru;                 // front = most recently used
    std::vector<std::list<int>::iterator> lru_it;
    std::vector<std::vector<int64_t>> refs;   // per slot: map entries that point at it
    // scratch
    void *d_reqs; size_t d_reqs_cap;
    int32_t *d_slots; size_t d_slots_cap;
    int32_t *d_upd; size_t d_upd_cap;   // (index, value) pairs of map updates
    std::vector<int32_t> pending;       // host list of (index, value) map updates not yet on the device
};

__global__ void k_scatter_i32(int32_t *dst, const int32_t *pairs, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i < n) dst[pairs[2 * i]] = pairs[2 * i + 1];
}

__global__ void k_copy_field(uint8_t *fields, const int32_t *src_dst, int n)
{
    // one workgroup of 256 threads per 4 KB field: 16 bytes per thread
    const int i = blockIdx.x;
    if(i >= n) return;
    const uint4 *s = (const uint4*)(fields + ((size_t)src_dst[2 * i] << 12));
    uint4 *d = (uint4*)(fields + ((size_t)src_dst[2 * i + 1] << 12));
    d[threadIdx.x] = s[threadIdx.x];
}

static int grow(navhip_ctx *ctx, void **p, size_t *cap, size_t need)
{
    if(*cap >= need) return NAVHIP_OK;
    if(*p) HIPCHK(ctx, hipFree(*p));
    *p = nullptr; *cap = 0;
    const size_t want = need + need / 2 + 64;
    HIPCHK(ctx, hipMalloc(p, want));
    *cap = want;
    return NAVHIP_OK;
}

static void pool_touch(nh_pool *P, int slot)
{
    P->lru.erase(P->lru_it[slot]);
    P->lru.push_front(slot);
    P->lru_it[slot] = P->lru.begin();
}

// slot of ff_id, taking the least recently used one when it is new (*fresh = the slot holds nothing
// of this id yet).  `pinned`: slots that must not be evicted (used by the batch in flight).
static int pool_slot_for(nh_pool *P, uint64_t id, bool *fresh, const std::vector<uint8_t> *pinned)
{
    auto it = P->slot_of.find(id);
    if(it != P->slot_of.end()) {
        *fresh = false;
        pool_touch(P, it->second);
        return it->second;
    }
    int slot = -1;
    for(auto r = P->lru.rbegin(); r != P->lru.rend(); ++r)
        if(!pinned || !(*pinned)[*r]) { slot = *r; break; }
    if(slot < 0) return -1;
    if(P->used[slot]) {
        // evict: every (dest, chunk) mapping that points at the slot is dropped
        P->slot_of.erase(P->id_of[slot]);
        for(int64_t e : P->refs[slot]) {
            if(P->h_map[(size_t)e] == slot) {
                P->h_map[(size_t)e] = -1;
                P->pending.push_back((int32_t)e); P->pending.push_back(-1);
            }
        }
        P->refs[slot].clear();
    }
    P->used[slot] = 1;
    P->id_of[slot] = id;
    P->slot_of[id] = slot;
    pool_touch(P, slot);
    *fresh = true;
    return slot;
}

// slots a pool build took for ids that were not resident: handed back when the call fails, so that no id
// is ever registered for a slot whose field was not built (the evicted fields are gone either way)
static void pool_rollback(nh_pool *P, const std::vector<int> &fresh_slots)
{
    for(int slot : fresh_slots) {
        if(!P->used[slot]) continue;
        P->slot_of.erase(P->id_of[slot]);
        P->used[slot] = 0;
        P->refs[slot].clear();
        // least recently used again: the next new id takes it first
        P->lru.erase(P->lru_it[slot]);
        P->lru.push_back(slot);
        P->lru_it[slot] = std::prev(P->lru.end());
    }
}

// drop the field of `slot`: every mapping that points at it becomes "no field", the slot is free again
static void pool_drop_slot(nh_pool *P, int slot)
{
    if(!P->used[slot]) return;
    for(int64_t e : P->refs[slot]) {
        if(P->h_map[(size_t)e] == slot) {
            P->h_map[(size_t)e] = -1;
            P->pending.push_back((int32_t)e); P->pending.push_back(-1);
        }
    }
    pool_rollback(P, std::vector<int>{slot});
}

static int pool_flush_map(navhip_ctx *ctx, nh_pool *P, hipStream_t s)
{
    if(P->pending.empty()) return NAVHIP_OK;
    {
        // one update per table entry: the scatter kernel applies the pairs in parallel, so two updates of
        // the same entry in one launch (a chunk re-mapped twice between two flushes) would race -- the
        // LAST one is the one that counts
        std::unordered_map<int32_t, size_t> last;
        for(size_t i = 0; i + 1 < P->pending.size(); i += 2) last[P->pending[i]] = i;
        if(last.size() * 2 != P->pending.size()) {
            std::vector<int32_t> uniq;
            uniq.reserve(last.size() * 2);
            for(size_t i = 0; i + 1 < P->pending.size(); i += 2)
                if(last[P->pending[i]] == i) { uniq.push_back(P->pending[i]); uniq.push_back(P->pending[i + 1]); }
            P->pending.swap(uniq);
        }
    }
    const int n = (int)(P->pending.size() / 2);
    int rc = grow(ctx, (void**)&P->d_upd, &P->d_upd_cap, P->pending.size() * sizeof(int32_t));
    if(rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(P->d_upd, P->pending.data(), P->pending.size() * sizeof(int32_t),
                               hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_scatter_i32, dim3((n + 255) / 256), dim3(256), 0, s, P->d_map, (const int32_t*)P->d_upd, n);
    HIPCHK(ctx, hipStreamSynchronize(s));          // (the host vector is reused)
    P->pending.clear();
    return NAVHIP_OK;
}

// (see nh_is_pinned below)
static struct { const void *p; bool pinned; } s_pin_cache[32];
static int s_pin_next;
static std::mutex s_pin_mu;            // (the cache is the process's: contexts on several threads share it)
static void pin_forget(const void *p) { std::lock_guard<std::mutex> lock(s_pin_mu); for(auto &e : s_pin_cache) if(e.p == p) e.p = nullptr; }

extern "C" {

int navhip_pool_create(navhip_ctx *ctx, int n_slots, int n_dests)
{
    if(!ctx || n_slots < 1 || n_dests < 1) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if((int64_t)n_dests * ctx->nchunks > 0x7fffffffLL) {       // (map updates travel as 32-bit entry indices)
        ctx->last_error = "navhip_pool_create: n_dests x chunks does not fit 31 bits";
        return NAVHIP_ERR_INVALID;
    }
    navhip_pool_destroy(ctx);
    nh_pool *P = new (std::nothrow) nh_pool();
    if(!P) return NAVHIP_ERR_NOMEM;
    P->n_slots = n_slots; P->n_dests = n_dests; P->nchunks = ctx->nchunks;
    P->d_fields = nullptr; P->d_map = nullptr;
    P->d_reqs = nullptr; P->d_reqs_cap = 0; P->d_slots = nullptr; P->d_slots_cap = 0;
    P->d_upd = nullptr; P->d_upd_cap = 0;
    ctx->pool = P;
    HIPCHK(ctx, hipMalloc((void**)&P->d_fields, (size_t)n_slots * NH_CELLS));
    HIPCHK(ctx, hipMalloc((void**)&P->d_map, (size_t)n_dests * P->nchunks * sizeof(int32_t)));
    HIPCHK(ctx, hipMemsetAsync(P->d_fields, 0, (size_t)n_slots * NH_CELLS, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(P->d_map, 0xff, (size_t)n_dests * P->nchunks * sizeof(int32_t), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    P->h_map.assign((size_t)n_dests * P->nchunks, -1);
    P->id_of.assign(n_slots, 0); P->used.assign(n_slots, 0);
    P->refs.assign(n_slots, {});
    P->lru_it.resize(n_slots);
    for(int i = 0; i < n_slots; i++) { P->lru.push_back(i); P->lru_it[i] = std::prev(P->lru.end()); }
    return NAVHIP_OK;
}

void navhip_pool_destroy(navhip_ctx *ctx)
{
    if(!ctx || !ctx->pool) return;
    nh_pool *P = ctx->pool;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipFree(P->d_fields); hipFree(P->d_map); hipFree(P->d_reqs); hipFree(P->d_slots); hipFree(P->d_upd);
    delete P;
    ctx->pool = nullptr;
}

int navhip_pool_clear(navhip_ctx *ctx)
{
    if(!ctx || !ctx->pool) return NAVHIP_ERR_INVALID;
    const int s = ctx->pool->n_slots, d = ctx->pool->n_dests;
    return navhip_pool_create(ctx, s, d);
}

int navhip_pool_contains(navhip_ctx *ctx, uint64_t ff_id)
{
    if(!ctx || !ctx->pool) return 0;
    return ctx->pool->slot_of.count(ff_id) ? 1 : 0;
}

int navhip_pool_invalidate(navhip_ctx *ctx, uint64_t ff_id)
{
    if(!ctx || !ctx->pool) return NAVHIP_ERR_INVALID;
    nh_pool *P = ctx->pool;
    auto it = P->slot_of.find(ff_id);
    if(it == P->slot_of.end()) return NAVHIP_OK;               // (lru_flow_remove of an absent key: no-op)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pool_drop_slot(P, it->second);                             // unregister, least recently used again
    return pool_flush_map(ctx, P, ctx->stream);
}

int navhip_pool_put(navhip_ctx *ctx, uint64_t ff_id, const uint8_t *dirs)
{
    if(!ctx || !ctx->pool || !dirs) return NAVHIP_ERR_INVALID;
    nh_pool *P = ctx->pool;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    bool fresh;
    const int slot = pool_slot_for(P, ff_id, &fresh, nullptr);
    if(slot < 0) return NAVHIP_ERR_NOMEM;
    HIPCHK(ctx, hipMemcpyAsync(P->d_fields + ((size_t)slot << 12), dirs, NH_CELLS, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return pool_flush_map(ctx, P, ctx->stream);
}

int navhip_pool_get(navhip_ctx *ctx, uint64_t ff_id, uint8_t *out_dirs)
{
    if(!ctx || !ctx->pool || !out_dirs) return NAVHIP_ERR_INVALID;
    nh_pool *P = ctx->pool;
    auto it = P->slot_of.find(ff_id);
    if(it == P->slot_of.end()) return NAVHIP_ERR_NOT_UPLOADED;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(out_dirs, P->d_fields + ((size_t)it->second << 12), NH_CELLS, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NAVHIP_OK;
}

int navhip_pool_map(navhip_ctx *ctx, int n, const int32_t *dest, const uint16_t *chunk_r,
                    const uint16_t *chunk_c, const uint64_t *ff_ids)
{
    if(!ctx || !ctx->pool || n < 0 || (n > 0 && (!dest || !chunk_r || !chunk_c || !ff_ids))) return NAVHIP_ERR_INVALID;
    nh_pool *P = ctx->pool;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for(int i = 0; i < n; i++) {
        if(dest[i] < 0 || dest[i] >= P->n_dests || chunk_r[i] >= ctx->h || chunk_c[i] >= ctx->w) return NAVHIP_ERR_INVALID;
        const int64_t e = (int64_t)dest[i] * P->nchunks + (int)chunk_r[i] * ctx->w + chunk_c[i];
        int slot = -1;                                   // an id that is not resident maps to "no field"
        auto it = P->slot_of.find(ff_ids[i]);
        if(it != P->slot_of.end()) slot = it->second;
        if(P->h_map[(size_t)e] != slot) {
            if(slot >= 0) P->refs[slot].push_back(e);     // (once per change: a host that re-puts its
                                                          // mappings every tick must not grow the list)
            P->h_map[(size_t)e] = slot;
            P->pending.push_back((int32_t)e); P->pending.push_back(slot);
        }
    }
    return pool_flush_map(ctx, P, ctx->stream);
}

__global__ void k_zero_fields(uint8_t *fields, const int32_t *slots, int n)
{
    const int i = blockIdx.x;
    if(i >= n) return;
    ((uint4*)(fields + ((size_t)slots[i] << 12)))[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
}

int navhip_pool_build(navhip_ctx *ctx, const navhip_field_req *reqs, const uint64_t *ff_ids,
                      const uint64_t *base_ids, int n, uint8_t *out_dirs)
{
    if(!ctx || !ctx->pool || n < 0 || (n > 0 && (!reqs || !ff_ids))) return NAVHIP_ERR_INVALID;
    if(n == 0) return NAVHIP_OK;
    nh_pool *P = ctx->pool;
    if(n > P->n_slots) { ctx->last_error = "navhip_pool_build: more requests than pool slots"; return NAVHIP_ERR_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc = nh_validate_field_reqs(ctx, reqs, n);
    if(rc) return rc;
    std::vector<navhip_field_req> rq(reqs, reqs + n);
    // ---- plan first, commit nothing: every base must be resident or produced earlier in this call, and
    // the fields the call touches (its ids and the resident bases it reads) must fit the pool together
    // -- they stay pinned for the WHOLE call, so that no sub-batch evicts what another one reads or wrote
    std::vector<uint64_t> base(n, 0);
    {
        std::unordered_map<uint64_t, int> seen;          // ids of earlier requests of this call
        std::unordered_map<uint64_t, int> touched;       // distinct ids + resident bases
        for(int i = 0; i < n; i++) {
            const uint64_t b = (base_ids && (rq[i].flags & NAVHIP_REQ_INOUT)) ? base_ids[i] : 0;
            if(b && b != ff_ids[i]) {
                if(!seen.count(b)) {
                    if(!P->slot_of.count(b)) { ctx->last_error = "navhip_pool_build: base field not resident"; return NAVHIP_ERR_NOT_UPLOADED; }
                    touched[b] = 1;
                }
                base[i] = b;
            }
            seen[ff_ids[i]] = i;
            touched[ff_ids[i]] = 1;
        }
        if((int)touched.size() > P->n_slots) { ctx->last_error = "navhip_pool_build: the call touches more fields than the pool has slots"; return NAVHIP_ERR_NOMEM; }
    }
    rc = grow(ctx, &P->d_reqs, &P->d_reqs_cap, (size_t)n * sizeof(navhip_field_req));
    if(!rc) rc = grow(ctx, (void**)&P->d_slots, &P->d_slots_cap, (size_t)n * 4 * sizeof(int32_t));
    if(rc) return rc;
    std::vector<int32_t> slots(n), copies, zeros;
    std::vector<uint8_t> pinned(P->n_slots, 0);
    std::vector<int> fresh_slots;
    for(int i = 0; i < n; i++) {
        if(!base[i]) continue;
        auto it = P->slot_of.find(base[i]);
        if(it != P->slot_of.end()) pinned[it->second] = 1;
    }
    // A failure undoes the sub-batch it happened in, ALL of it: a fresh slot was never valid, and a resident
    // field that was being rebuilt in place (or had just received its base copy) may be half written -- it is
    // dropped like navhip_pool_invalidate drops it (the host builds it again on the next miss).  The sub-batches
    // before it were built, synchronised and stay registered.
    std::vector<int> batch_slots;
#define POOL_FAIL(code) do { hipStreamSynchronize(s); for(int sl_ : batch_slots) pool_drop_slot(P, sl_); pool_flush_map(ctx, P, s); return (code); } while(0)
#define POOL_HIPCHK(expr) do { hipError_t _e = (expr); if(_e != hipSuccess) { ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e); POOL_FAIL(NAVHIP_ERR_DEVICE); } } while(0)
    // Sub-batches: a request that reads (base) or rewrites the slot of an EARLIER request of the same
    // sub-batch has to wait for it -- the in-place chains of nav.c:1987-2011 -- and so has a request
    // that rewrites a field an earlier request of the sub-batch is COPIED from (the copies of a
    // sub-batch run in one launch in front of its builds).
    int begin = 0;
    while(begin < n) {
        std::unordered_map<uint64_t, int> written, copied_from;
        int end = begin;
        copies.clear(); zeros.clear(); batch_slots.clear();
        for(; end < n; end++) {
            const uint64_t b = base[end];
            if(written.count(ff_ids[end]) || (b && written.count(b)) || copied_from.count(ff_ids[end])) break;
            int base_slot = -1;
            if(b) {
                base_slot = P->slot_of.find(b)->second;      // resident by now: planned above
                pinned[base_slot] = 1;
                copied_from[b] = 1;
            }
            bool fresh;
            const int slot = pool_slot_for(P, ff_ids[end], &fresh, &pinned);
            if(slot < 0) { ctx->last_error = "navhip_pool_build: no evictable slot"; POOL_FAIL(NAVHIP_ERR_NOMEM); }
            pinned[slot] = 1;
            if(fresh) fresh_slots.push_back(slot);
            batch_slots.push_back(slot);
            slots[end] = slot;
            written[ff_ids[end]] = end;
            if(base_slot >= 0) { copies.push_back(base_slot); copies.push_back(slot); }
            if(fresh && base_slot < 0) {
                // A new slot still holds the field of the id it was taken from.  An in-place request
                // without a base starts from N_FlowFieldInit; nothing is cached yet, so "only if changed"
                // does not apply; and a request the kernel may decline (NAVHIP_REQ_LIVE_IIDS: a portal
                // blocked from end to end) must find N_FlowFieldInit's all-FD_NONE field there, not the
                // evicted one.
                if((rq[end].flags & NAVHIP_REQ_INOUT)
                && rq[end].type != NAVHIP_TARGET_NEAREST_PATHABLE && !(rq[end].flags & NAVHIP_REQ_ISLAND_NEAREST))
                    rq[end].flags &= ~NAVHIP_REQ_INOUT;
                rq[end].flags &= ~NAVHIP_REQ_IF_CHANGED;
                if(rq[end].flags & (NAVHIP_REQ_LIVE_IIDS | NAVHIP_REQ_INOUT)) zeros.push_back(slot);
            }else if(fresh) {
                rq[end].flags &= ~NAVHIP_REQ_IF_CHANGED;     // (a copy of its base is not the field asked for)
            }
        }
        const int m = end - begin;
        rc = pool_flush_map(ctx, P, s);
        if(rc) POOL_FAIL(rc);
        POOL_HIPCHK(hipMemcpyAsync((char*)P->d_reqs + (size_t)begin * sizeof(navhip_field_req), &rq[begin],
                                   (size_t)m * sizeof(navhip_field_req), hipMemcpyHostToDevice, s));
        POOL_HIPCHK(hipMemcpyAsync(P->d_slots + begin, &slots[begin], (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s));
        if(!zeros.empty()) {
            int32_t *d_z = P->d_slots + 3 * (size_t)n;
            POOL_HIPCHK(hipMemcpyAsync(d_z, zeros.data(), zeros.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_zero_fields, dim3((unsigned)zeros.size()), dim3(256), 0, s, P->d_fields,
                               (const int32_t*)d_z, (int)zeros.size());
        }
        if(!copies.empty()) {
            int32_t *d_cp = P->d_slots + n;
            POOL_HIPCHK(hipMemcpyAsync(d_cp, copies.data(), copies.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_copy_field, dim3((unsigned)(copies.size() / 2)), dim3(256), 0, s, P->d_fields,
                               (const int32_t*)d_cp, (int)(copies.size() / 2));
        }
        rc = navhip_build_fields_slots_dev(ctx, (const navhip_field_req*)P->d_reqs + begin, m, P->d_fields,
                                           P->d_slots + begin, s);
        if(rc) POOL_FAIL(rc);
        POOL_HIPCHK(hipStreamSynchronize(s));       // (host vectors of the next sub-batch reuse the staging)
        begin = end;
    }
    if(out_dirs) {
        // (the fields are built and registered: a failing read-back does not unregister them)
        for(int i = 0; i < n; i++)
            HIPCHK(ctx, hipMemcpyAsync(out_dirs + ((size_t)i << 12), P->d_fields + ((size_t)slots[i] << 12), NH_CELLS,
                                       hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
    }
#undef POOL_FAIL
#undef POOL_HIPCHK
    return NAVHIP_OK;
}

// -----------------------------------------------------------------------------------------------
// pinned host memory + asynchronous velocity step
// -----------------------------------------------------------------------------------------------
void *navhip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void navhip_host_free(void *p)
{
    if(p) { pin_forget(p); hipHostFree(p); }
}

}  // extern "C"

static bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if(hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

struct nh_async {
    bool        pending;
    bool        empty;           // the submitted world had no entities: nothing is in flight, poll / wait succeed
    hipEvent_t  done;
    // pinned staging: one slab for the inputs, one for the outputs
    char  *h_in;  size_t h_in_cap;
    char  *h_out; size_t h_out_cap;
    struct cp { void *dst; const void *src; size_t bytes; };
    std::vector<cp> finish;          // staging -> caller copies at completion
    // the attribute tables on the device are those of this epoch / entity count / flock count
    uint32_t static_epoch; int32_t static_n, static_f;
    // what the last submitted step left on the device: its snapshot (device addresses) and its outputs -- the state half
    // of the tick reads them in place (navhip_state_pass_resident)
    bool            resident;
    navhip_world    d_world;
    navhip_step_out d_out;
};

static int pinned_grow(navhip_ctx *ctx, char **p, size_t *cap, size_t need)
{
    if(*cap >= need) return NAVHIP_OK;
    if(*p) HIPCHK(ctx, hipHostFree(*p));
    *p = nullptr; *cap = 0;
    const size_t want = need + need / 2 + 4096;
    HIPCHK(ctx, hipHostMalloc((void**)p, want, hipHostMallocDefault));
    *cap = want;
    return NAVHIP_OK;
}

extern "C" {

int navhip_agent_step_submit(navhip_ctx *ctx, const navhip_world *w, const navhip_step_out *out)
{
    if(!ctx || !w || !out || !out->vel_xz) return NAVHIP_ERR_INVALID;
    if(w->n_ents < 0) return NAVHIP_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if(!ctx->async) {
        ctx->async = new (std::nothrow) nh_async();
        if(!ctx->async) return NAVHIP_ERR_NOMEM;
        ctx->async->pending = false; ctx->async->empty = false; ctx->async->h_in = ctx->async->h_out = nullptr;
        ctx->async->h_in_cap = ctx->async->h_out_cap = 0;
        ctx->async->static_epoch = 0; ctx->async->static_n = ctx->async->static_f = 0;
        ctx->async->resident = false;
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->async->done, hipEventDisableTiming));
    }
    nh_async *A = ctx->async;
    if(A->pending) { ctx->last_error = "navhip_agent_step_submit: a step is already in flight"; return NAVHIP_ERR_INVALID; }
    A->resident = false;
    if(w->n_ents == 0) {            // an empty world is a valid tick: submit / poll / wait all succeed
        A->finish.clear();
        A->pending = true; A->empty = true;
        return NAVHIP_OK;
    }
    A->empty = false;
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)w->n_ents, F = (size_t)w->n_flocks;
    size_t nmembers = (F > 0 && w->flock_offsets) ? (size_t)w->flock_offsets[F] : 0;
    const bool resident = w->n_field_slots == NAVHIP_POOL_RESIDENT;
    struct item { const void *host; size_t bytes; int slot; const void **dev; bool attr; bool early; };
    navhip_world d = *w;
    std::vector<item> items = {
        {w->pos_xz, n * 8, 0, (const void**)&d.pos_xz, false, true},       {w->vel_xz, n * 8, 1, (const void**)&d.vel_xz, false, true},
        // (radius, max_speed and flags change without any entity being added, removed or re-flocked --
        // MOVE_CMD_SET_MAX_SPEED movement.c:3226, selection-radius updates, ENTITY_FLAG_GARRISONED toggled
        // by the garrison module --: they travel every tick, only the flock tables are epoch-cached)
        {w->radius, n * 4, 2, (const void**)&d.radius, false, true}, {w->max_speed, n * 4, 3, (const void**)&d.max_speed},
        {w->speed, n * 4, 4, (const void**)&d.speed},         {w->flags, n * 4, 5, (const void**)&d.flags, false, true},
        {w->state, n, 6, (const void**)&d.state, false, true},             {w->has_dest_los, n, 7, (const void**)&d.has_dest_los},
        {w->flock, n * 4, 8, (const void**)&d.flock, true},         {w->vdes_xz, n * 8, 9, (const void**)&d.vdes_xz},
        {w->flock_target_xz, F * 8, 10, (const void**)&d.flock_target_xz, true},
        {w->flock_offsets, (F + 1) * 4, 11, (const void**)&d.flock_offsets, true},
        {w->flock_members, nmembers * 4, 12, (const void**)&d.flock_members, true},
        {w->form_ready, n, 24, (const void**)&d.form_ready},  {w->cell_pos_xz, n * 8, 25, (const void**)&d.cell_pos_xz},
        {w->form_cohesion_xz, n * 8, 26, (const void**)&d.form_cohesion_xz},
        {w->form_align_xz, n * 8, 27, (const void**)&d.form_align_xz},
        {w->form_drag_xz, n * 8, 28, (const void**)&d.form_drag_xz},
        {w->arrival_sink_xz, n * 8, 36, (const void**)&d.arrival_sink_xz, false, true},
        {w->arrival_flags, n, 37, (const void**)&d.arrival_flags, false, true},
        {w->los_pool, (size_t)(w->n_los_slots > 0 ? w->n_los_slots : 0) * NH_CELLS, 38, (const void**)&d.los_pool},
        {w->flock_los_slot, F * (size_t)ctx->nchunks * 4, 39, (const void**)&d.flock_los_slot},
        {w->los_pos_xz, n * 8, 40, (const void**)&d.los_pos_xz},
        {w->region_row, n * 4, 46, (const void**)&d.region_row},
    };
    if(!resident)
        items.push_back({w->region_field_slot, (size_t)(w->n_region_rows > 0 ? w->n_region_rows : 0) * (size_t)ctx->nchunks * 4,
                         47, (const void**)&d.region_field_slot});
    if(!resident) {
        items.push_back({w->flock_field_slot, F * (size_t)ctx->nchunks * 4, 13, (const void**)&d.flock_field_slot});
        items.push_back({w->field_pool, (size_t)(w->n_field_slots > 0 ? w->n_field_slots : 0) * NH_CELLS, 14,
                         (const void**)&d.field_pool});
    }
    // inputs: pageable arrays are packed into the pinned slab (one memcpy each) and cross the bus as ONE
    // transfer into one device slab -- a dozen separate copies cost a dozen hand-overs to the copy
    // engine, more than the bytes --; pinned ones (navhip_host_alloc) are transferred in place
    // The attribute tables keep their own device buffers and stay there while the caller repeats its
    // static_epoch.
    const size_t AL = 256;
    const bool attrs_resident = w->static_epoch != 0 && w->static_epoch == A->static_epoch
                                && A->static_n == w->n_ents && A->static_f == w->n_flocks;
    size_t need = 0;
    for(auto &it : items) if(it.host && !it.attr && !is_pinned(it.host)) need += (it.bytes + AL - 1) & ~(AL - 1);
    int rc = pinned_grow(ctx, &A->h_in, &A->h_in_cap, need);
    if(rc) return rc;
    char *d_slab = nullptr;
    if(need) { rc = navhip_stage_reserve(ctx, 44, need, (void**)&d_slab); if(rc) return rc; }
    // Two passes: what the front of the step reads (positions, velocities, states, the attribute
    // tables) goes first and the front is started on it (navhip_agent_prefetch_dev); the rest is packed
    // and transferred while the spatial hash, the neighbour walk and the cohesion term run.
    size_t off = 0, sent = 0;
    for(auto &it : items) *it.dev = nullptr;
    for(int pass = 0; pass < 2; pass++) {
        for(auto &it : items) {
            if(!it.host || (it.attr || it.early) != (pass == 0)) continue;
            if(it.attr || is_pinned(it.host)) {
                rc = navhip_stage_reserve(ctx, it.slot, it.bytes, (void**)it.dev);
                if(rc) return rc;
                if(it.attr && attrs_resident) continue;
                if(it.bytes) HIPCHK(ctx, hipMemcpyAsync((void*)*it.dev, it.host, it.bytes, hipMemcpyHostToDevice, s));
            }else{
                memcpy(A->h_in + off, it.host, it.bytes);
                *it.dev = d_slab + off;
                off += (it.bytes + AL - 1) & ~(AL - 1);
                // hand what has been packed to the copy engine about every megabyte: it moves that
                // part while the next one is being packed
                if(off - sent >= ((size_t)1 << 20)) {
                    HIPCHK(ctx, hipMemcpyAsync(d_slab + sent, A->h_in + sent, off - sent, hipMemcpyHostToDevice, s));
                    sent = off;
                }
            }
        }
        if(off > sent) {
            HIPCHK(ctx, hipMemcpyAsync(d_slab + sent, A->h_in + sent, off - sent, hipMemcpyHostToDevice, s));
            sent = off;
        }
        if(pass == 0) {
            // (device addresses of the late arrays: fixed before their contents arrive)
            size_t o2 = off;
            for(auto &it : items) {
                if(!it.host || it.attr || it.early) continue;
                if(is_pinned(it.host)) { rc = navhip_stage_reserve(ctx, it.slot, it.bytes, (void**)it.dev); if(rc) return rc; }
                else { *it.dev = d_slab + o2; o2 += (it.bytes + AL - 1) & ~(AL - 1); }
            }
            rc = navhip_agent_prefetch_dev(ctx, &d, s);
            if(rc) return rc;
        }
    }
    A->static_epoch = w->static_epoch; A->static_n = w->n_ents; A->static_f = w->n_flocks;
    // outputs: the same -- one device slab, one transfer, for the pageable ones
    size_t b = (size_t)w->work_begin, e = (size_t)w->work_end;
    if(b == 0 && e == 0) e = n;
    navhip_step_out dout = {nullptr, nullptr, nullptr, nullptr, nullptr};
    struct oitem { void **dev; void *host; size_t row; int slot; size_t off; } outs[5] = {
        {(void**)&dout.vel_xz, out->vel_xz, 8, 15, 0},   {(void**)&dout.new_pos_xz, out->new_pos_xz, 8, 16, 0},
        {(void**)&dout.vdes_xz, out->vdes_xz, 8, 17, 0}, {(void**)&dout.vpref_xz, out->vpref_xz, 8, 18, 0},
        {(void**)&dout.status, out->status, 1, 19, 0}};
    // (a pageable output lives in the slab as its rows [b, e): the kernels index by entity, so the
    // array's device address is the slab position minus b rows)
    size_t oneed = 0;
    for(auto &o : outs) if(o.host && !is_pinned(o.host)) { o.off = oneed; oneed += ((e - b) * o.row + AL - 1) & ~(AL - 1); }
    rc = pinned_grow(ctx, &A->h_out, &A->h_out_cap, oneed);
    if(rc) return rc;
    char *d_oslab = nullptr;
    if(oneed) { rc = navhip_stage_reserve(ctx, 45, oneed, (void**)&d_oslab); if(rc) return rc; }
    for(auto &o : outs) {
        if(!o.host) continue;
        if(is_pinned(o.host)) {
            rc = navhip_stage_reserve(ctx, o.slot, n * o.row, o.dev);
            if(rc) return rc;
        }else{
            *o.dev = d_oslab + o.off - b * o.row;
        }
    }
    rc = navhip_agent_step_dev(ctx, &d, &dout, s);
    if(rc) return rc;
    A->finish.clear();
    for(auto &o : outs) {
        if(!o.host || e <= b) continue;
        const size_t bytes = (e - b) * o.row;
        char *dst = (char*)o.host + b * o.row;
        if(is_pinned(o.host))
            HIPCHK(ctx, hipMemcpyAsync(dst, (char*)*o.dev + b * o.row, bytes, hipMemcpyDeviceToHost, s));
        else
            A->finish.push_back({dst, A->h_out + o.off, bytes});
    }
    if(oneed) HIPCHK(ctx, hipMemcpyAsync(A->h_out, d_oslab, oneed, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipEventRecord(A->done, s));
    A->pending = true;
    A->d_world = d; A->d_out = dout; A->resident = true;
    return NAVHIP_OK;
}

static int async_finish(navhip_ctx *ctx)
{
    nh_async *A = ctx->async;
    for(auto &c : A->finish) memcpy(c.dst, c.src, c.bytes);
    A->finish.clear();
    A->pending = false;
    return NAVHIP_OK;
}

int navhip_agent_step_poll(navhip_ctx *ctx)
{
    if(!ctx || !ctx->async || !ctx->async->pending) return NAVHIP_ERR_INVALID;
    if(ctx->async->empty) return async_finish(ctx);
    hipError_t e = hipEventQuery(ctx->async->done);
    if(e == hipErrorNotReady) return 1;
    if(e != hipSuccess) { ctx->last_error = std::string("navhip_agent_step_poll: ") + hipGetErrorString(e); return NAVHIP_ERR_DEVICE; }
    return async_finish(ctx);
}

int navhip_agent_step_wait(navhip_ctx *ctx)
{
    if(!ctx || !ctx->async || !ctx->async->pending) return NAVHIP_ERR_INVALID;
    if(ctx->async->empty) return async_finish(ctx);
    // a step takes a few hundred microseconds: poll for that long (a blocking wait costs a wake-up of
    // tens of microseconds), then block
    for(int spin = 0; spin < 20000; spin++) {
        const hipError_t e = hipEventQuery(ctx->async->done);
        if(e == hipSuccess) return async_finish(ctx);
        if(e != hipErrorNotReady) break;
    }
    HIPCHK(ctx, hipEventSynchronize(ctx->async->done));
    return async_finish(ctx);
}

}  // extern "C"

void nh_async_invalidate_static(navhip_ctx *ctx) { if(ctx->async) { ctx->async->static_epoch = 0; ctx->async->resident = false; } }

// the device-side snapshot and outputs of the last COMPLETED host-buffer step (false: none, or still in flight)
bool nh_async_resident(navhip_ctx *ctx, navhip_world *w, navhip_step_out *o)
{
    nh_async *A = ctx->async;
    if(!A || !A->resident || A->pending || A->empty) return false;
    *w = A->d_world; *o = A->d_out;
    return true;
}

// its two pinned staging slabs, grown to the sizes asked for (the step is complete: nobody reads them)
int nh_async_slabs(navhip_ctx *ctx, size_t in_bytes, size_t out_bytes, char **h_in, char **h_out)
{
    nh_async *A = ctx->async;
    if(!A || A->pending) return NAVHIP_ERR_INVALID;
    int rc = pinned_grow(ctx, &A->h_in, &A->h_in_cap, in_bytes);
    if(!rc) rc = pinned_grow(ctx, &A->h_out, &A->h_out_cap, out_bytes);
    if(rc) return rc;
    *h_in = A->h_in; *h_out = A->h_out;
    return NAVHIP_OK;
}

// hipPointerGetAttributes costs microseconds; a host passes the same page-locked arrays every tick: the answers for the
// last few pointers are remembered, under a mutex (the cache is shared by every context of the process).  An array
// freed and reallocated pageable at the same address would be answered stale: that only changes the copy path taken
// (a staged copy of pinned memory, or a direct transfer the runtime stages itself), never a result; navhip_host_free
// forgets its pointer.
bool nh_is_pinned(const void *p)
{
    if(!p) return false;
    std::lock_guard<std::mutex> lock(s_pin_mu);
    for(auto &e : s_pin_cache) if(e.p == p) return e.pinned;
    const bool r = is_pinned(p);
    s_pin_cache[s_pin_next] = {p, r};
    s_pin_next = (s_pin_next + 1) % 32;
    return r;
}


void nh_async_destroy(navhip_ctx *ctx)
{
    if(!ctx->async) return;
    if(ctx->async->h_in) hipHostFree(ctx->async->h_in);
    if(ctx->async->h_out) hipHostFree(ctx->async->h_out);
    hipEventDestroy(ctx->async->done);
    delete ctx->async;
    ctx->async = nullptr;
}

const uint8_t *nh_pool_fields(const navhip_ctx *ctx) { return ctx->pool ? ctx->pool->d_fields : nullptr; }
const int32_t *nh_pool_map(const navhip_ctx *ctx) { return ctx->pool ? ctx->pool->d_map : nullptr; }
int nh_pool_dests(const navhip_ctx *ctx) { return ctx->pool ? ctx->pool->n_dests : 0; }
int nh_pool_slots(const navhip_ctx *ctx) { return ctx->pool ? ctx->pool->n_slots : 0; }
