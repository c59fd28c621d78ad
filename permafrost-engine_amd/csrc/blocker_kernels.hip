// blocker_kernels.hip -- dynamic obstacles on the device (gfx950): blocker refcounts, the derived
// passability masks and the per-chunk local-island labels, so that incremental field repair
// (BASELINE.json configs[4]) needs no host round trip.
//
// Reference semantics (permafrost-engine src/):
//   N_BlockersIncref / N_BlockersDecref          navigation/nav.c:4663,4685
//     n_update_blockers_circle_{ground,water,air} nav.c:1051-1133  (1x1: tiles under the circle;
//       3x3/5x5/7x7 layers: the same tiles plus 1/2/3 successive contours, each contour taken of
//       the PREVIOUS contour only)
//     n_update_blockers                           nav.c:1017  (blockers += d, factions[f] += d)
//   M_Tile_AllUnderCircle  map/tile.c:687, M_Tile_Contour map/tile.c:759, M_Tile_Bounds :356,
//   M_Tile_DescForPoint2D :547, C_CircleRectIntersection phys/collision.c:997 (C_PointInsideRect2D
//   :756, C_LineCircleIntersection :960)
//   n_update_local_islands nav.c:967 (+ n_visit_island_local :901): 4-connected components of the
//   passable, unblocked cells of one chunk, ids 1.. in row-major order of each component's first cell
//
// k_blockers_circles   one WAVE per circle.  The tile sets live in a 64x64 bit window centred on
//                      the circle's tile (lane = window row, u64 = window columns): the disc
//                      test runs per tile, a contour is one 8-neighbour dilation minus the set
//                      (3 shifts + 2 DPP row moves), the 1024-entry caps of the reference's
//                      scratch arrays are applied with a wave prefix-popcount in row-major order.
//                      Refcounts are updated with 32-bit atomics on the containing word.
// k_refresh_touched    one wave per chunk: rebuild the passability row masks of chunks whose
//                      blockers were touched, flag the chunk `changed` when any mask differs.
// k_local_islands      one wave per chunk: bit-parallel flood fill per component, labels kept
//                      bit-sliced, expanded to u16 and stored coalesced through LDS.
#include "navhip_internal.h"
#include "wave_bits.h"

struct nh_blk_params {
    int       w, h;
    float     map_x, map_z;
    uint16_t *blockers[NAVHIP_NAV_LAYER_MAX];
    uint8_t  *factions[NAVHIP_NAV_LAYER_MAX];
    uint8_t  *touched[NAVHIP_NAV_LAYER_MAX];
};

#define BLK_EPS 0.0009765625f      /* collision.c:64 EPSILON 1/1024 */

// PFM_Vec2_Len (pf_math.c:82): sqrt in double of a float sum == correctly rounded float sqrt
__device__ __forceinline__ float len2f(float x, float z) { return __builtin_sqrtf(x * x + z * z); }

// C_LineCircleIntersection, collision.c:960 (pow(.,2) and sqrt are double in the reference)
__device__ bool line_circle(float ax, float az, float bx, float bz, float cx, float cz, float radius)
{
    float dx = bx - ax, dz = bz - az;
    float A = (float)((double)dx * (double)dx + (double)dz * (double)dz);
    float B = 2 * (dx * (ax - cx) + dz * (az - cz));
    float fx = ax - cx, fz = az - cz;
    float C = (float)((double)fx * (double)fx + (double)fz * (double)fz - (double)radius * (double)radius);
    float det = (float)((double)B * (double)B - (double)(4 * A * C));
    float t;
    if(det < 0.0f || A < BLK_EPS) {
        return false;
    }else if(det == 0.0f) {
        t = __fdiv_rn(-B, 2 * A);
    }else{
        double sq = __builtin_sqrt((double)det);
        float t1 = (float)(((double)(-B) + sq) / (double)(2 * A));
        float t2 = (float)(((double)(-B) - sq) / (double)(2 * A));
        t = t1 < t2 ? t1 : t2;
    }
    return !(t < 0.0f || t > 1.0f);
}

// C_CircleRectIntersection (collision.c:997) against M_Tile_Bounds (tile.c:356) of the nav tile
// (abs_r, abs_c): box {x, z, 4, 4} with x decreasing to the right
__device__ bool circle_hits_tile(const nh_blk_params &P, float cx, float cz, float radius,
                                 int abs_r, int abs_c)
{
    const int chunk_r = abs_r >> 6, chunk_c = abs_c >> 6, tile_r = abs_r & 63, tile_c = abs_c & 63;
    const float x = (P.map_x - (float)(chunk_c * 256)) - (float)(tile_c * 4);
    const float z = (P.map_z + (float)(chunk_r * 256)) + (float)(tile_r * 4);
    const float wdt = 4.0f, hgt = 4.0f;
    const float qx[4] = {x - wdt, x, x, x - wdt};
    const float qz[4] = {z, z, z + hgt, z + hgt};
    // C_PointInsideRect2D(center, a, b, c, d), collision.c:756
    {
        float apx = cx - qx[0], apz = cz - qz[0];
        float abx = qx[1] - qx[0], abz = qz[1] - qz[0];
        float adx = qx[3] - qx[0], adz = qz[3] - qz[0];
        float ap_ab = apx * abx + apz * abz, ap_ad = apx * adx + apz * adz;
        if((ap_ab >= 0.0f && ap_ab <= abx * abx + abz * abz)
        && (ap_ad >= 0.0f && ap_ad <= adx * adx + adz * adz))
            return true;
    }
    for(int i = 0; i < 4; i++)
        if(len2f(qx[i] - cx, qz[i] - cz) <= radius) return true;
    for(int i = 0; i < 4; i++) {
        int j = (i + 1) & 3;
        if(line_circle(qx[i], qz[i], qx[j], qz[j], cx, cz, radius)) return true;
    }
    return false;
}

// keep the first `cap` set bits of the tile (row-major: lane ascending, then bit ascending)
__device__ __forceinline__ uint64_t cap_rowmajor(uint64_t m, int cap, int lane)
{
    int cnt = __popcll(m), incl = cnt;
#pragma unroll
    for(int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(incl, d);
        if(lane >= d) incl += o;
    }
    const int before = incl - cnt;
    if(before >= cap) return 0;
    int keep = cap - before;
    if(keep >= cnt) return m;
    // lowest `keep` set bits
    uint64_t out = 0;
    for(int k = 0; k < keep; k++) { uint64_t b = m & (~m + 1); out |= b; m ^= b; }
    return out;
}

// M_Tile_Contour (tile.c:759) of a set inside the window: unmarked in-map cells with a marked
// 8-neighbour
__device__ __forceinline__ uint64_t contour_of(uint64_t m, uint64_t inmap)
{
    const u64x s = mk(m);
    const u64x up = from_n(s), dn = from_s(s);
    const u64x h  = s  | from_w(s)  | from_e(s);
    const u64x hu = up | from_w(up) | from_e(up);
    const u64x hd = dn | from_w(dn) | from_e(dn);
    return to64(andn(h | hu | hd, s)) & inmap;
}

__device__ __forceinline__ void add_u16(uint16_t *p, int delta)
{
    unsigned int *word = (unsigned int*)((uintptr_t)p & ~(uintptr_t)3);
    const unsigned sh = ((uintptr_t)p & 2) ? 16u : 0u;
    if(delta >= 0) atomicAdd(word, (unsigned)delta << sh);
    else           atomicSub(word, (unsigned)(-delta) << sh);
}

__device__ __forceinline__ void add_u8(uint8_t *p, int delta)
{
    unsigned int *word = (unsigned int*)((uintptr_t)p & ~(uintptr_t)3);
    const unsigned sh = (unsigned)((uintptr_t)p & 3) * 8u;
    if(delta >= 0) atomicAdd(word, (unsigned)delta << sh);
    else           atomicSub(word, (unsigned)(-delta) << sh);
}

__global__ __launch_bounds__(256) void k_blockers_circles(nh_blk_params P, const navhip_circle *circles,
                                                          int n)
{
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if(wave >= n) return;
    const navhip_circle C = circles[wave];

    // M_Tile_DescForPoint2D (tile.c:547) of the centre; off-map centre: nothing (tile.c:693-695)
    const float width = (float)(P.w * 256), height = (float)(P.h * 256);
    if(C.x > P.map_x || C.x < P.map_x - width) return;
    if(C.z < P.map_z || C.z > P.map_z + height) return;
    int chunk_r = (int)(fabsf(P.map_z - C.z) / 256.0f), chunk_c = (int)(fabsf(P.map_x - C.x) / 256.0f);
    chunk_r = min(max(chunk_r, 0), P.h - 1);
    chunk_c = min(max(chunk_c, 0), P.w - 1);
    const float base_x = P.map_x - (float)(chunk_c * 256), base_z = P.map_z + (float)(chunk_r * 256);
    int tile_r = (int)(fabsf(base_z - C.z) / 4.0f), tile_c = (int)(fabsf(base_x - C.x) / 4.0f);
    tile_r = min(max(tile_r, 0), 63);
    tile_c = min(max(tile_c, 0), 63);
    const int cen_r = chunk_r * 64 + tile_r, cen_c = chunk_c * 64 + tile_c;
    const int ntiles = (int)ceil((double)(C.radius / 4));              // tile.c:700

    // window row of this lane, and the in-map columns of that row
    const int abs_r = cen_r - 32 + lane;
    const bool row_ok = abs_r >= 0 && abs_r < P.h * 64;
    uint64_t inmap = 0;
    if(row_ok) {
        const int lo = max(0, 32 - cen_c), hi = min(63, P.w * 64 - 1 - cen_c + 32);
        if(hi >= lo) inmap = (hi - lo == 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
    }

    // M_Tile_AllUnderCircle, tile.c:687: dr, dc in [-ntiles, ntiles], row-major, cap 1024
    uint64_t tds = 0;
    const int dr = lane - 32;
    if(row_ok && dr >= -ntiles && dr <= ntiles) {
        for(int dc = -ntiles; dc <= ntiles; dc++) {
            const int b = 32 + dc;
            if(b < 0 || b > 63 || !((inmap >> b) & 1)) continue;       // M_Tile_RelativeDesc fails
            if(circle_hits_tile(P, C.x, C.z, C.radius, abs_r, cen_c + dc)) tds |= 1ull << b;
        }
    }
    tds = cap_rowmajor(tds, 1024, lane);
    const uint64_t o3 = cap_rowmajor(contour_of(tds, inmap), 1024, lane);
    const uint64_t o5 = cap_rowmajor(contour_of(o3, inmap), 1024, lane);
    const uint64_t o7 = cap_rowmajor(contour_of(o5, inmap), 1024, lane);

    // N_BlockersIncref/Decref: air entities update the air layers, everything else water AND ground
    const bool air = (C.flags & NAVHIP_ENTITY_FLAG_AIR) != 0;
    const int ngroups = air ? 1 : 2;
    const int bases[2] = {air ? 8 : 4, 0};
    const uint64_t sets[4] = {tds, o3, o5, o7};
    for(int gi = 0; gi < ngroups; gi++) {
        for(int sidx = 0; sidx < 4; sidx++) {
            const int layer = bases[gi] + sidx;
            uint16_t *bl = P.blockers[layer];
            if(!bl) continue;                                           // layer not resident
            uint8_t *fa = P.factions[layer];
            for(int k = 0; k <= sidx; k++) {
                uint64_t m = sets[k];
                while(m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    const int ac = cen_c - 32 + b;
                    const int chunk = (abs_r >> 6) * P.w + (ac >> 6);
                    const size_t cell = ((size_t)chunk << 12) + (abs_r & 63) * 64 + (ac & 63);
                    add_u16(bl + cell, C.delta);
                    if(fa && C.faction_id >= 0 && C.faction_id < NAVHIP_MAX_FACTIONS)
                        add_u8(fa + ((size_t)chunk * NAVHIP_MAX_FACTIONS << 12)
                                  + ((size_t)C.faction_id << 12) + (abs_r & 63) * 64 + (ac & 63), C.delta);
                    P.touched[layer][chunk] = 1;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_refresh_touched(const uint8_t *cost, const uint16_t *blockers,
                                                         uint64_t *passmask, uint64_t *probemask, uint8_t *unit_cost,
                                                         uint8_t *touched, uint8_t *changed, int nchunks)
{
    const int chunk = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if(chunk >= nchunks || !touched[chunk]) return;
    const uint8_t  *cb = cost + ((size_t)chunk << 12);
    const uint16_t *bl = blockers + ((size_t)chunk << 12);
    uint64_t mine = 0, blocked = 0;
    bool nonunit = false;
    for(int r = 0; r < 64; r++) {
        uint32_t cst = cb[r * 64 + lane], blk = bl[r * 64 + lane];
        uint64_t m = __ballot(cst != NAVHIP_COST_IMPASSABLE && blk == 0), mb = __ballot(blk > 0);
        nonunit |= (cst != NAVHIP_COST_IMPASSABLE && cst != 1);
        if(lane == r) { mine = m; blocked = mb; }
    }
    const bool differs = passmask[(size_t)chunk * 64 + lane] != mine;
    passmask[(size_t)chunk * 64 + lane] = mine;
    probemask[((size_t)chunk * 64 + lane) * 2 + 1] = blocked;       // (the cost half does not change here)
    const bool any_nonunit = __any(nonunit), any_diff = __any(differs);
    if(lane == 0) {
        unit_cost[chunk] = any_nonunit ? 0 : 1;
        if(any_diff) changed[chunk] = 1;
        touched[chunk] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// k_local_islands: n_update_local_islands (nav.c:967) for the chunks flagged in `only` (all chunks
// when only == NULL)
#define LI_MAXP 12        /* <= 2048 components in a 64x64 tile */
__global__ __launch_bounds__(256) void k_local_islands(const uint64_t *passmask, uint16_t *local_islands,
                                                       const uint8_t *only, int nchunks)
{
    __shared__ __attribute__((aligned(16))) uint16_t stage[4][NH_CELLS];      // 32 KB
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + wib;
    if(chunk >= nchunks) return;
    if(only && !only[chunk]) return;
    const u64x pass = mk(passmask[(size_t)chunk * 64 + lane]);
    u64x open = pass;
    u64x pl[LI_MAXP];
#pragma unroll
    for(int k = 0; k < LI_MAXP; k++) pl[k] = u64x{0, 0};
    int id = 0;
    for(;;) {
        const uint64_t rows = __ballot(nz(open));
        if(!rows) break;
        // first open cell in row-major order seeds the next component (ids start at 1)
        const int r0 = __builtin_ctzll(rows);
        const uint64_t row0 = __shfl(to64(open), r0);
        const int c0 = __builtin_ctzll(row0);
        id++;
        u64x comp = (lane == r0) ? mk(1ull << c0) : u64x{0, 0};
        for(;;) {                                                   // 4-connected flood fill
            u64x grow = (comp | from_w(comp) | from_e(comp) | from_n(comp) | from_s(comp)) & open;
            // saturate along the row before paying for another cross-lane step
            u64x g2 = (grow | from_w(grow) | from_e(grow)) & open;
            g2 = (g2 | from_w(g2) | from_e(g2)) & open;
            const bool more = nz(andn(g2, comp));
            comp = g2;
            if(!__any(more)) break;
        }
        open = andn(open, comp);
#pragma unroll
        for(int k = 0; k < LI_MAXP; k++)
            if((id >> k) & 1) pl[k] = pl[k] | comp;                 // id is wave-uniform
    }
    // expand: impassable / blocked cells are ISLAND_NONE (0xffff)
    uint16_t *st = stage[wib];
    const int P = 32 - __builtin_clz((unsigned)id | 1u);
    const uint64_t ps = to64(pass);
    for(int c = 0; c < 64; c++) {
        uint32_t v = 0;
#pragma unroll
        for(int k = 0; k < LI_MAXP; k++)
            if(k < P) v |= (uint32_t)((to64(pl[k]) >> c) & 1ull) << k;
        st[lane * 64 + c] = ((ps >> c) & 1ull) ? (uint16_t)v : (uint16_t)NAVHIP_ISLAND_NONE;
    }
    __builtin_amdgcn_wave_barrier();
    uint16_t *out = local_islands + ((size_t)chunk << 12);
#pragma unroll
    for(int j = 0; j < 8; j++)
        *(uint4*)((char*)out + j * 1024 + lane * 16) = *(const uint4*)((const char*)st + j * 1024 + lane * 16);
}

// ---------------------------------------------------------------------------------------------
void nh_launch_blockers_circles(navhip_ctx *ctx, const navhip_circle *d_circles, int n, float map_x,
                                float map_z, hipStream_t s)
{
    nh_blk_params P;
    P.w = ctx->w; P.h = ctx->h; P.map_x = map_x; P.map_z = map_z;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        P.blockers[l] = ctx->layers[l].blockers;
        P.factions[l] = ctx->layers[l].factions;
        P.touched[l]  = ctx->layers[l].touched;
    }
    if(n > 0)
        hipLaunchKernelGGL(k_blockers_circles, dim3((n + 3) / 4), dim3(256), 0, s, P, d_circles, n);
    // derived state of every touched chunk, then the island labels of the chunks that changed
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        navhip_layer &L = ctx->layers[l];
        if(!L.blockers || !L.cost) continue;
        hipLaunchKernelGGL(k_refresh_touched, dim3((ctx->nchunks + 3) / 4), dim3(256), 0, s, L.cost,
                           L.blockers, L.passmask, L.probemask, L.unit_cost, L.touched, L.changed, ctx->nchunks);
        if(L.local_islands)
            hipLaunchKernelGGL(k_local_islands, dim3((ctx->nchunks + 3) / 4), dim3(256), 0, s,
                               L.passmask, L.local_islands, L.changed, ctx->nchunks);
    }
}

void nh_launch_local_islands(navhip_ctx *ctx, int layer, hipStream_t s)
{
    navhip_layer &L = ctx->layers[layer];
    hipLaunchKernelGGL(k_local_islands, dim3((ctx->nchunks + 3) / 4), dim3(256), 0, s, L.passmask,
                       L.local_islands, (const uint8_t*)nullptr, ctx->nchunks);
}
