// agent_kernels.hip -- per-agent movement step for gfx950 (MI355X), hand-written HIP.
//
// Reference semantics (permafrost-engine):
//   src/game/movement.c   move_velocity_work :3395, point_seek_vpref :1870, point_seek_total_force
//                         :1745, arrive_force_point :1546, cohesion_force :1653, separation_force
//                         :1690, nullify_impass_components :1831, find_neighbours :2768,
//                         enemy_seek_vpref :1946, vec2_truncate :643, position accept :2336-2358
//   src/game/clearpath.c  G_ClearPath_NewVelocity :694 and everything under it
//   src/phys/collision.c  C_InfiniteLineIntersection :820, C_RayRayIntersection2D :854
//   src/lib/public/bitmap_grid.h  bg_*_inrange_circle :1376 (candidate order + fixed-point test)
//   src/navigation/nav.c  N_DesiredPointSeekVelocity :3468, n_interpolated_flow_dir :3407,
//                         N_PositionPathable/Blocked :4055/:4070;  src/map/tile.c :356,:391,:547
//
// Arithmetic mirrors the reference's C expression by expression (same types, same order, no FMA
// contraction, IEEE divide / sqrt, double-precision exp) so that results are reproducible against
// the CPU path far inside the 1e-4 parity bound; all order-dependent float sums are evaluated in
// the reference's own order.
//
// Kernels
//   k_sp_*        device spatial hash: fixed-point cell binning, scan, scatter, per-cell ordering
//                 (the layout bg_ent_cleanup produces after inserting uids 0..n-1).
//   k_cohesion    one THREAD per flock member; the O(N*F) exp-weighted centroid, every thread
//                 walking its flock's member list in order (lanes of a wave share the flock, so
//                 member loads are wave-uniform broadcasts).
//   k_agent_step  one WAVE per agent: flow sampling, arrive/separation forces, impassable-component
//                 nullification, neighbour gather (lanes test 64 candidates at a time, ballot +
//                 prefix-popcount compaction keeps the reference's order and caps), ClearPath with
//                 lanes spread over ray pairs and a lexicographic wave arg-min that reproduces the
//                 reference's first-wins tie-break, truncation and the position accept test.
#include "navhip_internal.h"
#include "agent_internal.h"

// ---------------------------------------------------------------------------------------------
// exact-arithmetic helpers (pf_math.c:58-94)
// ---------------------------------------------------------------------------------------------
struct v2 { float x, z; };
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2 mkv(float x, float z) { v2 r; r.x = x; r.z = z; return r; }
__device__ __forceinline__ v2 vadd(v2 a, v2 b) { return mkv(a.x + b.x, a.z + b.z); }
__device__ __forceinline__ v2 vsub(v2 a, v2 b) { return mkv(a.x - b.x, a.z - b.z); }
__device__ __forceinline__ v2 vscale(v2 a, float s) { return mkv(a.x * s, a.z * s); }
__device__ __forceinline__ float vdot(v2 a, v2 b) { return a.x * b.x + a.z * b.z; }
// PFM_Vec2_Len: sqrt in double of a float sum, rounded to float == correctly rounded float sqrt.
// __builtin_sqrtf is IEEE-correct under -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn is
// NOT (it lowers to the native v_sqrt_f32 approximation in this ROCm).
//
// Correctly rounded sqrt for s == 0 or s in the normal range well away from its ends: v_sqrt_f32
// (<= 1 ulp) plus the same one-ulp fix-up the compiler's IEEE expansion uses, without that
// expansion's input scaling / class handling (which only matter for denormal, infinite or NaN s).
__device__ __forceinline__ float sqrt_rn_normal(float s)
{
    float r = __builtin_amdgcn_sqrtf(s);
    const float rm = __int_as_float(__float_as_int(r) - 1), rp = __int_as_float(__float_as_int(r) + 1);
    const float em = __builtin_fmaf(-rm, r, s), ep = __builtin_fmaf(-rp, r, s);
    r = (em <= 0.0f) ? rm : r;
    r = (ep > 0.0f) ? rp : r;
    return r;
}
__device__ __forceinline__ float vlen(v2 a)
{
    const float s = a.x * a.x + a.z * a.z;
    if(!(s >= 0x1p-90f && s <= 0x1p90f) && s != 0.0f) {
        asm volatile("" ::: "memory");      // a real branch: keep the expansion out of the common path
        return __builtin_sqrtf(s);
    }
    return sqrt_rn_normal(s);
}
__device__ __forceinline__ v2 vnormal(v2 a)
{
    float l = vlen(a);
    return mkv(__fdiv_rn(a.x, l), __fdiv_rn(a.z, l));
}
// vec2_truncate, movement.c:643
__device__ __forceinline__ v2 vtrunc(v2 a, float max_len)
{
    if(vlen(a) > max_len) {
        a = vnormal(a);
        a = vscale(a, max_len);
    }
    return a;
}

#define CP_EPS 0.0009765625f   /* 1.0/1024: exactly representable, so float compares == the
                                  reference's float-vs-double compares */

// optional section timing of k_agent_step (scripts/section_prof.py builds a private copy of the
// library with -DNH_SECTION_PROF; the shipped build carries none of it)
#ifdef NH_SECTION_PROF
__device__ unsigned long long nh_sec[1024 * 32];      // [slot = block & 1023][counter]
#define SEC_BEGIN() unsigned long long _sec_t0 = __builtin_amdgcn_s_memtime()
#define SEC_MARK(k) do { unsigned long long _n = __builtin_amdgcn_s_memtime(); \
        if(lane == 0) { unsigned long long *_b = nh_sec + (blockIdx.x & 1023) * 32; \
                        atomicAdd(&_b[k], _n - _sec_t0); atomicAdd(&_b[16 + (k)], 1ull); } \
        _sec_t0 = __builtin_amdgcn_s_memtime(); } while(0)
extern "C" int navhip_debug_sections(unsigned long long *out, int reset)
{
    static unsigned long long h[1024 * 32];
    if(hipMemcpyFromSymbol(h, HIP_SYMBOL(nh_sec), sizeof(h)) != hipSuccess) return 1;
    for(int k = 0; k < 32; k++) { out[k] = 0; for(int b = 0; b < 1024; b++) out[k] += h[b * 32 + k]; }
    if(reset) { for(auto &x : h) x = 0; hipMemcpyToSymbol(HIP_SYMBOL(nh_sec), h, sizeof(h)); }
    return 0;
}
#else
#define SEC_BEGIN()
#define SEC_MARK(k)
#endif
// cost of one section under real contention: scripts/dup_prof.py builds private copies of the library
// that run section NH_DUP twice (same results) and compares tick times
#ifndef NH_DUP
#define NH_DUP 0
#endif
#define DUP_BARRIER() asm volatile("" ::: "memory")

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// tile lookups (tile.c:547 M_Tile_DescForPoint2D, nav.c:4055/4070)
// ---------------------------------------------------------------------------------------------
struct tiledesc { int chunk_r, chunk_c, tile_r, tile_c; };

__device__ __forceinline__ bool tile_for_point(const nh_step_params &P, float x, float z, tiledesc &out)
{
    const float width = (float)(P.map.w * 256), height = (float)(P.map.h * 256);
    if(x > P.map_x || x < P.map_x - width) return false;
    if(z < P.map_z || z > P.map_z + height) return false;
    int chunk_r = (int)(fabsf(P.map_z - z) / 256.0f);      // exact: division by a power of two
    int chunk_c = (int)(fabsf(P.map_x - x) / 256.0f);
    chunk_r = min(max(chunk_r, 0), P.map.h - 1);
    chunk_c = min(max(chunk_c, 0), P.map.w - 1);
    float base_x = P.map_x - (float)(chunk_c * 256);
    float base_z = P.map_z + (float)(chunk_r * 256);
    int tile_r = (int)(fabsf(base_z - z) / 4.0f);
    int tile_c = (int)(fabsf(base_x - x) / 4.0f);
    out.chunk_r = chunk_r; out.chunk_c = chunk_c;
    out.tile_r = min(max(tile_r, 0), 63);
    out.tile_c = min(max(tile_c, 0), 63);
    return true;
}

// Entity_NavLayerWithRadius, entity.c:554
__device__ __forceinline__ int nav_layer_for(uint32_t flags, float radius)
{
    int base = (flags & NAVHIP_ENTITY_FLAG_WATER) ? 4 : (flags & NAVHIP_ENTITY_FLAG_AIR) ? 8 : 0;
    if(radius >= 15.0f) return base + 3;
    if(radius >= 10.0f) return base + 2;
    if(radius >= 5.0f)  return base + 1;
    return base;
}

__device__ __forceinline__ bool pos_pathable(const nh_step_params &P, int layer, float x, float z)
{
    tiledesc t;
    if(!tile_for_point(P, x, z, t)) return false;     // reference asserts; off-map = not pathable
    const uint8_t *cost = P.map.layers[layer].cost;
    return cost[((size_t)(t.chunk_r * P.map.w + t.chunk_c) << 12) + t.tile_r * 64 + t.tile_c]
           != NAVHIP_COST_IMPASSABLE;
}

__device__ __forceinline__ bool pos_blocked(const nh_step_params &P, int layer, float x, float z)
{
    tiledesc t;
    if(!tile_for_point(P, x, z, t)) return false;
    const uint16_t *bl = P.map.layers[layer].blockers;
    if(!bl) return false;
    return bl[((size_t)(t.chunk_r * P.map.w + t.chunk_c) << 12) + t.tile_r * 64 + t.tile_c] > 0;
}

// ---------------------------------------------------------------------------------------------
// flow-field sampling (nav.c:3407 n_interpolated_flow_dir, :3468 N_DesiredPointSeekVelocity)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ v2 flow_dir_vec(int dir)           // N_FlowDir, field.c:2428
{
    const float d = 0.70710678118654757f;                     // (float)(1.0f / sqrt(2.0f))
    switch(dir) {
    case NAVHIP_FD_NW: return mkv( d, -d);
    case NAVHIP_FD_N:  return mkv( 0.0f, -1.0f);
    case NAVHIP_FD_NE: return mkv(-d, -d);
    case NAVHIP_FD_W:  return mkv( 1.0f, 0.0f);
    case NAVHIP_FD_E:  return mkv(-1.0f, 0.0f);
    case NAVHIP_FD_SW: return mkv( d,  d);
    case NAVHIP_FD_S:  return mkv( 0.0f, 1.0f);
    case NAVHIP_FD_SE: return mkv(-d,  d);
    default:           return mkv(0.0f, 0.0f);
    }
}

// One thread per agent (k_agent_pre): the four taps are fetched one after the other; the latency is
// hidden by the other agents of the wave instead of by the other lanes of a wave-per-agent kernel.
__device__ v2 sample_flow(const nh_step_params &P, int flock, v2 pos, uint32_t &status)
{
    tiledesc t;
    if(flock < 0 || !P.flock_field_slot || !P.field_pool || !tile_for_point(P, pos.x, pos.z, t)) {
        status |= NAVHIP_ST_FIELD_MISS;
        return mkv(0.0f, 0.0f);
    }
    const int nchunks = P.map.w * P.map.h;
    const int32_t *slots = P.flock_field_slot + (size_t)flock * nchunks;
    int slot = slots[t.chunk_r * P.map.w + t.chunk_c];
    if(slot < 0) {
        status |= NAVHIP_ST_FIELD_MISS;
        return mkv(0.0f, 0.0f);
    }
    const uint8_t *base_ff = P.field_pool + ((size_t)slot << 12);
    int base_dir = base_ff[t.tile_r * 64 + t.tile_c] & 0xf;
    if(base_dir == NAVHIP_FD_NONE) status |= NAVHIP_ST_FIELD_NONE;

    // M_Tile_Bounds (tile.c:356): two sequential float subtractions / additions
    float bx = (P.map_x - (float)(t.chunk_c * 256)) - (float)(t.tile_c * 4);
    float bz = (P.map_z + (float)(t.chunk_r * 256)) + (float)(t.tile_r * 4);
    float cx = bx - 4.0f / 2.0f, cz = bz + 4.0f / 2.0f;
    float dx = pos.x - cx, dz = pos.z - cz;
    int dc = (dx < 0.0f) ? 1 : -1;
    int dr = (dz > 0.0f) ? 1 : -1;
    float wc = fminf(fabsf(dx) / 4.0f, 1.0f);
    float wr = fminf(fabsf(dz) / 4.0f, 1.0f);
    const int   sdc[4] = {0, dc, 0, dc};
    const int   sdr[4] = {0, 0, dr, dr};
    const float sw[4]  = {(1.0f - wc) * (1.0f - wr), wc * (1.0f - wr), (1.0f - wc) * wr, wc * wr};

    v2 acc = mkv(0.0f, 0.0f);
    float wsum = 0.0f;
#pragma unroll
    for(int i = 0; i < 4; i++) {
        if(sw[i] <= 0.0f) continue;
        // M_Tile_RelativeDesc, tile.c:391
        int abs_r = t.chunk_r * 64 + t.tile_r + sdr[i];
        int abs_c = t.chunk_c * 64 + t.tile_c + sdc[i];
        if(abs_r < 0 || abs_r >= P.map.h * 64 || abs_c < 0 || abs_c >= P.map.w * 64) continue;
        int cr = abs_r >> 6, cc = abs_c >> 6, tr = abs_r & 63, tc = abs_c & 63;
        const uint8_t *ff = base_ff;
        if(cr != t.chunk_r || cc != t.chunk_c) {
            int s2 = slots[cr * P.map.w + cc];
            if(s2 < 0) continue;
            ff = P.field_pool + ((size_t)s2 << 12);
        }
        int dir = ff[tr * 64 + tc] & 0xf;
        if(dir == NAVHIP_FD_NONE) continue;
        v2 scaled = vscale(flow_dir_vec(dir), sw[i]);
        acc = vadd(acc, scaled);
        wsum += sw[i];
    }
    if(wsum < 1e-6f || vlen(acc) < 1e-6f)
        return flow_dir_vec(base_dir);
    return vnormal(acc);
}

// move_work_in.ent_des_v: host supplied, or sampled from the device field pool (vdes_xz == NULL or
// a NaN entry)
__device__ __forceinline__ v2 load_vdes(const nh_step_params &P, int uid, int flock, v2 me,
                                        uint32_t &status)
{
    if(P.vdes_xz) {
        v2 v = mkv(P.vdes_xz[2 * uid], P.vdes_xz[2 * uid + 1]);
        if(v.x == v.x) return v;
    }
    return sample_flow(P, flock, me, status);
}

// ---------------------------------------------------------------------------------------------
// spatial hash (bitmap_grid.h): build
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t bg_scale(float v) { return __float2int_rn(v * 256.0f); }   // BG_SCALE_F

__device__ __forceinline__ int sp_cell_of(const nh_grid &G, int32_t ix, int32_t iy)
{
    int cx = (ix - G.origin_x) >> 12;             // BG_CELL_LOG2_INT = 8 + 4
    int cy = (iy - G.origin_y) >> 12;
    cx = min(max(cx, 0), G.grid_w - 1);
    cy = min(max(cy, 0), G.grid_h - 1);
    return cy * G.grid_w + cx;
}

#define SP_MAX_QUERY_R 30   /* largest query radius of the movement tick (separation, movement.c:1695) */
// Optional slab filter: when a rank steps only the entities [work_begin, work_end), nothing farther
// than the largest query radius of the tick (r = 30) from the bounding box of THOSE entities can be
// returned by any of its queries, and leaving such entities out changes neither the order nor the
// caps of what is returned.  box = {max(-ix), max(ix), max(-iy), max(iy)} over the slab in the
// x256 fixed point the queries compare in; INT_MIN-initialised.
__global__ __launch_bounds__(1024) void k_sp_bbox(const float *pos_xz, int begin, int end, int32_t *box)
{
    // few, large workgroups striding over the slab: four atomics per workgroup, not per wave
    __shared__ int32_t part[16][4];
    int32_t v[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    for(int i = begin + blockIdx.x * 1024 + threadIdx.x; i < end; i += gridDim.x * 1024) {
        const int32_t ix = bg_scale(pos_xz[2 * i]), iy = bg_scale(pos_xz[2 * i + 1]);
        v[0] = max(v[0], -ix); v[1] = max(v[1], ix); v[2] = max(v[2], -iy); v[3] = max(v[3], iy);
    }
#pragma unroll
    for(int q = 0; q < 4; q++) {
#pragma unroll
        for(int d = 32; d >= 1; d >>= 1) v[q] = max(v[q], __shfl_xor(v[q], d));
        if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][q] = v[q];
    }
    __syncthreads();
    if(threadIdx.x < 4) {
        int32_t m = INT32_MIN;
        for(int w = 0; w < 16; w++) m = max(m, part[w][threadIdx.x]);
        if(m != INT32_MIN) atomicMax(&box[threadIdx.x], m);
    }
}

__device__ __forceinline__ bool sp_in_box(const int32_t *box, int32_t ix, int32_t iy)
{
    if(!box) return true;
    const int32_t m = SP_MAX_QUERY_R * 256 + 256;     // BG_SCALE_F(largest radius) + 1 wu of slack
    // (int64: the INT_MIN box of an empty slab must reject everything without overflowing)
    return (int64_t)ix >= -(int64_t)box[0] - m && (int64_t)ix <= (int64_t)box[1] + m
        && (int64_t)iy >= -(int64_t)box[2] - m && (int64_t)iy <= (int64_t)box[3] + m;
}

// Every inserted entity also gets a packed 32-byte RECORD -- {pos.x, pos.z, radius, flags} |
// {vel.x, vel.z, state, -} -- so that the neighbour gathers of the agent step (separation, neighbour
// classification, garrison filter) fetch one or two 16-byte halves of one line per neighbour instead
// of 4-8 bytes from each of three to five structure-of-arrays lines.
__global__ __launch_bounds__(256) void k_sp_count(nh_grid G, const float *pos_xz, int n,
                                                  int32_t *ent_ix, int32_t *ent_iy,
                                                  int32_t *ent_cell, int32_t *cell_count,
                                                  const int32_t *box, nh_pack_src src, float4 *rec)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= n) return;
    const float px = pos_xz[2 * i], pz = pos_xz[2 * i + 1];
    int32_t ix = bg_scale(px), iy = bg_scale(pz);
    ent_ix[i] = ix; ent_iy[i] = iy;
    if(!sp_in_box(box, ix, iy)) { ent_cell[i] = -1; return; }
    int c = sp_cell_of(G, ix, iy);
    ent_cell[i] = c;
    atomicAdd(&cell_count[c], 1);
    if(rec) {
        rec[2 * i]     = make_float4(px, pz, src.radius[i], __uint_as_float(src.flags[i]));
        rec[2 * i + 1] = make_float4(src.vel_xz[2 * i], src.vel_xz[2 * i + 1],
                                     __uint_as_float((uint32_t)src.state[i]), 0.0f);
    }
}

// exclusive scan of cell_count[0..ncells) -> cell_start[0..ncells], two passes over 1024-cell
// blocks: (1) block-local exclusive scan + block totals, (2) add the sum of the preceding totals.
__global__ __launch_bounds__(1024) void k_sp_scan_local(const int32_t *cell_count, int32_t *cell_start,
                                                        int32_t *block_sum, int ncells)
{
    __shared__ int32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i = blockIdx.x * 1024 + t;
    int32_t v = (i < ncells) ? cell_count[i] : 0;
    int32_t incl = v;
#pragma unroll
    for(int d = 1; d < 64; d <<= 1) {
        int32_t o = __shfl_up(incl, d);
        if(lane >= d) incl += o;
    }
    if(lane == 63) wsum[w] = incl;
    __syncthreads();
    int32_t woff = 0, tot = 0;
#pragma unroll
    for(int k = 0; k < 16; k++) {
        int32_t x = wsum[k];
        if(k < w) woff += x;
        tot += x;
    }
    if(i < ncells) cell_start[i] = woff + incl - v;
    if(t == 0) block_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_sp_scan_add(int32_t *cell_start, const int32_t *block_sum,
                                                      int ncells, int nblocks)
{
    __shared__ int32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // sum of the totals of the blocks before this one (and, in the last block, of all blocks)
    int32_t part = 0, all = 0;
    for(int k = t; k < nblocks; k += 1024) {
        int32_t x = block_sum[k];
        all += x;
        if(k < (int)blockIdx.x) part += x;
    }
    const bool last = (int)blockIdx.x == nblocks - 1;
    int32_t red = last ? all : part;          // the last block needs both: two reductions
#pragma unroll
    for(int d = 32; d >= 1; d >>= 1) { red += __shfl_xor(red, d); part += __shfl_xor(part, d); }
    if(lane == 0) wsum[w] = red;
    __syncthreads();
    int32_t tot = 0;
#pragma unroll
    for(int k = 0; k < 16; k++) tot += wsum[k];
    __syncthreads();
    if(lane == 0) wsum[w] = part;
    __syncthreads();
    int32_t off = 0;
#pragma unroll
    for(int k = 0; k < 16; k++) off += wsum[k];
    const int i = blockIdx.x * 1024 + t;
    if(i < ncells) cell_start[i] += off;
    if(last && t == 0) cell_start[ncells] = tot;
}

__global__ __launch_bounds__(256) void k_sp_scatter(const int32_t *ent_cell, int n,
                                                    const int32_t *cell_start, int32_t *cell_fill,
                                                    int32_t *sorted_id)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= n) return;
    int c = ent_cell[i];
    if(c < 0) return;                            // outside the slab filter
    int slot = cell_start[c] + atomicAdd(&cell_fill[c], 1);
    sorted_id[slot] = i;
}

// Per-cell order: bg_ent_insert pushes at the head of the cell's overflow chain and
// bg_ent_cleanup copies the chain head-first (bitmap_grid.h:1102-1121,1515-1521), so after
// inserting uids 0..n-1 each cell holds its elements in DESCENDING uid order.
__global__ __launch_bounds__(256) void k_sp_order(const int32_t *cell_start, int ncells,
                                                  int32_t *sorted_id, const int32_t *ent_ix,
                                                  const int32_t *ent_iy, int32_t *sx, int32_t *sy)
{
    int c = blockIdx.x * 256 + threadIdx.x;
    if(c >= ncells) return;
    int b = cell_start[c], e = cell_start[c + 1];
    for(int i = b + 1; i < e; i++) {            // insertion sort, descending; cells are tiny
        int32_t v = sorted_id[i];
        int j = i - 1;
        while(j >= b && sorted_id[j] < v) { sorted_id[j + 1] = sorted_id[j]; j--; }
        sorted_id[j + 1] = v;
    }
    for(int i = b; i < e; i++) {
        int32_t id = sorted_id[i];
        sx[i] = ent_ix[id];
        sy[i] = ent_iy[id];
    }
}

// ---------------------------------------------------------------------------------------------
// spatial hash: query.  All 64 lanes cooperate on ONE query; returns the number written (wave
// uniform).  Visiting order == bg_*_inrange_circle (bitmap_grid.h:1408-1466): coarse 8x8 blocks
// row-major, inside a block fine rows top to bottom, cells left to right, packed elements in
// order.  Cells of one fine row are contiguous in the cell-sorted pool, so a (block,row) pair is
// one contiguous range that the lanes test 64 elements at a time; ballot + prefix popcount
// appends hits in order and enforces `maxout` exactly where the reference stops.
// ---------------------------------------------------------------------------------------------
// Extent of a query in fine cells + whether it takes the reference's wide-query path
// (bitmap_grid.h:1389-1397); returns false when the query box misses the grid.
// inclusive prefix sum over the 64 lanes: four row_shr steps inside each row of 16, then the two
// row broadcasts (DPP; zero fill outside the row)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

struct sp_extent { int cx_lo, cx_hi, cy_lo, cy_hi; bool wide; };

__device__ __forceinline__ bool sp_query_extent(const nh_grid &G, int32_t icx, int32_t icy, int32_t ir,
                                                sp_extent &E)
{
    const int32_t imnx = icx - ir, imxx = icx + ir, imny = icy - ir, imxy = icy + ir;
    // _bg_cell_extent, bitmap_grid.h:1236
    if(imxx < G.origin_x || imxy < G.origin_y) return false;
    const int32_t span_x = (int32_t)((uint32_t)G.grid_w << 12), span_y = (int32_t)((uint32_t)G.grid_h << 12);
    if(imnx >= G.origin_x + span_x || imny >= G.origin_y + span_y) return false;
    E.cx_lo = max((imnx - G.origin_x) >> 12, 0);
    E.cy_lo = max((imny - G.origin_y) >> 12, 0);
    E.cx_hi = min((imxx - G.origin_x) >> 12, G.grid_w - 1);
    E.cy_hi = min((imxy - G.origin_y) >> 12, G.grid_h - 1);
    E.wide = (int64_t)(E.cx_hi - E.cx_lo + 1) * (E.cy_hi - E.cy_lo + 1) * 4 >= (int64_t)G.grid_w * G.grid_h * 3;
    return true;
}

// out_d2 (optional, [maxout]): squared fixed-point distance of every hit (fits int32 for the
// ranges the movement tick uses), so that a narrower query around the same point can be derived
// from this one without touching memory again.
__device__ int sp_query_wave(const nh_grid &G, float x, float z, float range, int maxout,
                             uint32_t *out_ids, int lane, int32_t *out_d2 = nullptr)
{
    if(maxout <= 0 || range < 0.0f) return 0;
    const int32_t icx = bg_scale(x), icy = bg_scale(z), ir = bg_scale(range);
    const int64_t ir2 = (int64_t)ir * (int64_t)ir;
    sp_extent E;
    if(!sp_query_extent(G, icx, icy, ir, E)) return 0;

    int written = 0;
    if(E.wide) {
        // wide-query fast path (bitmap_grid.h:1389-1397): the clean pool is scanned linearly
        const int npool = G.cell_start[G.grid_w * G.grid_h];     // == G.n unless a slab filter is on
        for(int base = 0; base < npool; base += 64) {
            int k = base + lane;
            bool hit = false;
            int64_t d2 = 0;
            if(k < npool) {
                int64_t dx = (int64_t)G.sx[k] - icx, dy = (int64_t)G.sy[k] - icy;
                d2 = dx * dx + dy * dy;
                hit = d2 <= ir2;
            }
            uint64_t m = __ballot(hit);
            int p = written + __popcll(m & ((1ull << lane) - 1ull));
            if(hit && p < maxout) {
                out_ids[p] = (uint32_t)G.sorted_id[k];
                if(out_d2) out_d2[p] = (int32_t)d2;
            }
            written += __popcll(m);
            if(written >= maxout) return maxout;
        }
        return written;
    }

    // Visiting order (bitmap_grid.h:1408-1466): coarse 8x8 blocks row-major; inside a block fine
    // rows top to bottom, cells left to right, packed elements in order.  The cells of one fine
    // row inside one coarse block are contiguous in the cell-sorted pool: a SEGMENT.  One pass
    // resolves 64 segments in visiting order -- lane = ((coarse row, block column) << 3) | fine row,
    // with CB (a power of two) block columns and 8 / CB coarse rows per pass, so the r = 30 and
    // r = 10 boxes of the movement tick (at most 2 x 2 coarse blocks) take a single pass: two
    // cell_start loads per lane, a DPP prefix sum, then the candidates 64 at a time, each lane
    // finding its segment by binary search over the prefix.
    const int cxc_lo = E.cx_lo >> 3, cxc_hi = E.cx_hi >> 3, ncx = cxc_hi - cxc_lo + 1;
    const int cyc_lo = E.cy_lo >> 3, cyc_hi = E.cy_hi >> 3;
    const int lgcb = ncx <= 1 ? 0 : ncx <= 2 ? 1 : ncx <= 4 ? 2 : 3;
    const int CB = 1 << lgcb, CR = 8 >> lgcb;
    // squared distances fit 32 bits when the box is small (r = 30: |d| <= 7680 + 4095)
    const bool small = ir <= 16000;
    for(int cyc0 = cyc_lo; cyc0 <= cyc_hi; cyc0 += CR) {
        for(int cbase = 0; cbase < ncx; cbase += CB) {
            const int sg = lane >> 3, fyi = lane & 7;
            const int cyc = cyc0 + (sg >> lgcb), cxi = cbase + (sg & (CB - 1));
            int bb = 0, len = 0;
            if(cyc <= cyc_hi && cxi < ncx) {
                const int fy = cyc * 8 + fyi;
                if(fy >= E.cy_lo && fy <= E.cy_hi) {
                    const int cxc = cxc_lo + cxi;
                    const int fx0 = max(cxc * 8, E.cx_lo), fx1 = min(cxc * 8 + 8, E.cx_hi + 1);
                    bb = G.cell_start[fy * G.grid_w + fx0];
                    len = G.cell_start[fy * G.grid_w + fx1] - bb;
                }
            }
            const int incl = wave_incl_scan(len);
            const int total = __shfl(incl, 63);
            if(total == 0) continue;
            const int bo = bb - (incl - len);                         // pool index of candidate q: bo + q
            for(int base = 0; base < total; base += 64) {
                const int q = base + lane;
                // first lane whose inclusive prefix exceeds q = the segment of candidate q (empty
                // segments are stepped over because the prefix does not move); every lane runs
                // the shuffles
                int lo = 0;
#pragma unroll
                for(int st = 32; st >= 1; st >>= 1) {
                    const int pv = __shfl(incl, lo + st - 1);
                    if(pv <= q) lo += st;
                }
                const int kk = __shfl(bo, lo & 63) + q;
                const int k = (q < total) ? kk : -1;
                bool hit = false;
                int32_t d2s = 0;
                if(k >= 0) {
                    if(small) {
                        // (elements clamped into a border cell may be far away: range-check before
                        // squaring in 32 bits)
                        const int32_t dx = G.sx[k] - icx, dy = G.sy[k] - icy;
                        const bool near = (uint32_t)(dx + 32767) < 65535u && (uint32_t)(dy + 32767) < 65535u;
                        d2s = near ? dx * dx + dy * dy : 0x7fffffff;
                        hit = d2s <= (int32_t)ir2;
                    }else{
                        const int64_t dx = (int64_t)G.sx[k] - icx, dy = (int64_t)G.sy[k] - icy;
                        const int64_t d2 = dx * dx + dy * dy;
                        hit = d2 <= ir2;
                        d2s = (int32_t)d2;
                    }
                }
                uint64_t m = __ballot(hit);
                int p = written + __popcll(m & ((1ull << lane) - 1ull));
                if(hit && p < maxout) {
                    out_ids[p] = (uint32_t)G.sorted_id[k];
                    if(out_d2) out_d2[p] = d2s;
                }
                written += __popcll(m);
                if(written >= maxout) return maxout;
            }
        }
    }
    return written;
}

// filter_garrisoned, position.c:100-119: walk backwards, overwrite with the current last
__device__ __forceinline__ uint32_t rec_flags(const float4 *rec, uint32_t id)
{
    return __float_as_uint(rec[2 * id].w);
}

__device__ int filter_garrisoned_wave(const float4 *rec, uint32_t *ids, int count, int lane)
{
    bool any = false;
    for(int base = 0; base < count; base += 64) {
        int k = base + lane;
        any |= (k < count) && (rec_flags(rec, ids[k]) & NAVHIP_ENTITY_FLAG_GARRISONED);
    }
    if(!__any(any)) return count;
    int ret = count;
    if(lane == 0) {
        for(int i = count - 1; i >= 0; i--) {
            if(rec_flags(rec, ids[i]) & NAVHIP_ENTITY_FLAG_GARRISONED) {
                ids[i] = ids[ret - 1];
                ret--;
            }
        }
    }
    wave_sync();
    return __shfl(ret, 0);
}

// ---------------------------------------------------------------------------------------------
// ClearPath (clearpath.c), one wave per problem
// ---------------------------------------------------------------------------------------------
struct cpent { v2 pos, vel; float radius; };
struct ray   { v2 point, dir; };

// slope of a line as C_InfiniteLineIntersection takes it (collision.c:823-831): NaN = vertical
__device__ __forceinline__ float line_slope(v2 dir)
{
    return fabsf(dir.x) < CP_EPS ? __builtin_nanf("") : __fdiv_rn(dir.z, dir.x);
}

// C_InfiniteLineIntersection, collision.c:820 (including the l2.point term of the vertical-l2
// branch, :840), with the two slopes s1/s2 = line_slope(dir) supplied by the caller (they only
// depend on the line, and every line meets many others)
__device__ __forceinline__ bool line_isect(v2 p1, float s1, v2 p2, float s2, v2 &out)
{
    bool n1 = s1 != s1, n2 = s2 != s2;
    if(n1 && n2) return false;
    if(fabsf(s1 - s2) < CP_EPS) return false;
    if(n1 && !n2) {
        out.x = p1.x;
        out.z = (p1.x - p2.x) * s2 + p2.z;
    }else if(!n1 && n2) {
        out.x = p2.x;
        out.z = (p2.x - p1.x) * s1 + p2.z;
    }else{
        out.x = __fdiv_rn((s1 * p1.x - s2 * p2.x + p2.z - p1.z), (s1 - s2));
        out.z = s2 * (out.x - p2.x) + p2.z;
    }
    return true;
}

// `a / b < 0.0f` of C_RayRayIntersection2D (collision.c:862-871) without the division when the sign
// rule is safe: for finite a, b with a == 0 or |a| >= 2^-100 and |b| <= 2^20 the quotient cannot
// underflow to -0, so it is negative exactly when a != 0 and the signs differ (b = +-0 included:
// a/+-0 = +-inf).  ok = false -> the caller divides.
__device__ __forceinline__ bool quot_neg_fast(float a, float b, bool &ok)
{
    const float aa = fabsf(a);
    ok = ok && (aa >= 0x1p-100f || a == 0.0f) && aa < __builtin_inff() && fabsf(b) <= 0x1p20f;
    return a != 0.0f && ((__float_as_int(a) ^ __float_as_int(b)) < 0);
}

// C_RayRayIntersection2D, collision.c:854
__device__ __forceinline__ bool ray_isect(v2 p1, v2 d1, float s1, v2 p2, v2 d2, float s2, v2 &out)
{
    v2 p;
    if(!line_isect(p1, s1, p2, s2, p)) return false;
    bool ok = true;
    const float a1 = p.x - p1.x, a2 = p.z - p1.z, a3 = p.x - p2.x, a4 = p.z - p2.z;
    bool neg = quot_neg_fast(a1, d1.x, ok);
    neg |= quot_neg_fast(a2, d1.z, ok);
    neg |= quot_neg_fast(a3, d2.x, ok);
    neg |= quot_neg_fast(a4, d2.z, ok);
    if(!ok) {
        neg = __fdiv_rn(a1, d1.x) < 0.0f || __fdiv_rn(a2, d1.z) < 0.0f
           || __fdiv_rn(a3, d2.x) < 0.0f || __fdiv_rn(a4, d2.z) < 0.0f;
    }
    if(neg) return false;
    out = p;
    return true;
}

// compute_vo_edges, clearpath.c:130
__device__ __forceinline__ void vo_edges(const cpent &ent, const cpent &nb, v2 &out_right, v2 &out_left)
{
    v2 e2n = vnormal(vsub(nb.pos, ent.pos));
    v2 right = mkv(-e2n.z, e2n.x);
    right = vscale(right, nb.radius + ent.radius + 0.0f);      // CLEARPATH_BUFFER_RADIUS
    v2 right_tangent = vadd(nb.pos, right);
    v2 left_tangent = vsub(nb.pos, right);
    out_right = vnormal(vsub(right_tangent, ent.pos));
    out_left = vnormal(vsub(left_tangent, ent.pos));
}

// compute_vo :153 / compute_hrvo :180 -> (apex, left, right) + the slopes of the two sides
__device__ __forceinline__ void make_cone(const cpent &ent, const cpent &nb, bool hrvo, v2 &apex,
                                          v2 &left, v2 &right, float &sl, float &sr)
{
    vo_edges(ent, nb, right, left);
    sl = line_slope(left); sr = line_slope(right);
    const v2 vo_apex = vadd(ent.pos, nb.vel);
    apex = vo_apex;
    if(hrvo) {
        v2 apex_off = vscale(vadd(ent.vel, nb.vel), 0.5f);
        v2 rvo_apex = vadd(ent.pos, apex_off);
        v2 centerline = vadd(left, right);
        float det = (centerline.x * ent.vel.z) - (centerline.z * ent.vel.x);
        apex = rvo_apex;
        if(det > CP_EPS || det < -CP_EPS) {
            // :196-212: (rvo_apex, left) x (vo_apex, right) or the mirrored pair
            const bool pos = det > CP_EPS;
            v2 p = rvo_apex;
            line_isect(rvo_apex, pos ? sl : sr, vo_apex, pos ? sr : sl, p);
            apex = p;
        }
    }
}

// inside_pcr, clearpath.c:249.  The combined obstacle lives in LDS as two float4 per cone:
//   cones[2c]   = {apex.x, apex.z, slope(left), slope(right)}
//   cones[2c+1] = {left.x, left.z, right.x, right.z}
// (both rays of a cone start at its apex, rays_repr :291; ray 2c is the left side, 2c+1 the right).
//
// One cone, evaluated exactly as the reference does (normalisation with IEEE sqrt/divide; the
// reference normalises test - apex once per ray, with identical operands both times):
// true when `test` is strictly inside the cone.
__device__ __forceinline__ bool cone_contains_exact(float4 A, float4 B, v2 test)
{
    v2 ptt = mkv(test.x - A.x, test.z - A.y);
    if(vlen(ptt) < CP_EPS) return false;
    ptt = vnormal(ptt);
    float left_det = (ptt.z * B.x) - (ptt.x * B.y);
    if(left_det < CP_EPS) return false;
    float right_det = (ptt.z * B.z) - (ptt.x * B.w);
    if(right_det > -CP_EPS) return false;
    return true;
}

// The same verdict from cheap arithmetic (one v_rsq_f32 instead of a correctly rounded sqrt and two
// IEEE divides) whenever every comparison is decided with a safety margin; 2 = too close to a
// threshold, the caller falls back to the exact evaluation.  The exact determinant differs from
// (p.z*d.x - p.x*d.z)/|p| by < 4e-7 (six roundings of magnitudes <= 1) and the cheap one by
// < 1.5e-6, so a margin of 2e-5 around the +-1/1024 thresholds leaves an order of magnitude of
// slack; the |p| < 1/1024 test gets a relative margin of 1e-4.  The decisions -- hence the result
// of inside_pcr -- are identical to the exact evaluation by construction.
__device__ __forceinline__ int cone_contains_fast(float4 A, float4 B, v2 test)
{
    const float MARG = 2e-5f;
    const float px = test.x - A.x, pz = test.z - A.y;
    const float s = px * px + pz * pz;
    const float inv = __builtin_amdgcn_rsqf(s);
    const float len = s * inv;
    if(!(s > 0.0f) || !(s < 1e30f) || fabsf(len - CP_EPS) <= CP_EPS * 1e-4f) return 2;
    if(len < CP_EPS) return 0;
    const float detl = (pz * B.x - px * B.y) * inv;
    if(fabsf(detl - CP_EPS) <= MARG) return 2;
    if(detl < CP_EPS) return 0;
    const float detr = (pz * B.z - px * B.w) * inv;
    if(fabsf(detr + CP_EPS) <= MARG) return 2;
    if(detr > -CP_EPS) return 0;
    return 1;
}

__device__ __forceinline__ bool cone_contains(float4 A, float4 B, v2 test)
{
    int v = cone_contains_fast(A, B, test);
    if(v == 2) v = cone_contains_exact(A, B, test) ? 1 : 0;
    return v == 1;
}

__device__ __forceinline__ bool inside_pcr(const float4 *cones, int n_cones, v2 test)
{
    for(int c = 0; c < n_cones; c++)
        if(cone_contains(cones[2 * c], cones[2 * c + 1], test)) return true;
    return false;
}

// lexicographic (key, idx) wave arg-min; key = +inf means "no candidate"
__device__ __forceinline__ void wave_argmin(float &key, int &idx)
{
#pragma unroll
    for(int d = 32; d >= 1; d >>= 1) {
        float ok = __shfl_xor(key, d);
        int   oi = __shfl_xor(idx, d);
        bool take = (ok < key) || (ok == key && oi < idx);
        if(take) { key = ok; idx = oi; }
    }
}

// LDS scratch of one ClearPath problem: the cones (2 float4 each, <= 64 cones) and a queue of
// candidate points (<= 128 pending)
struct cp_scratch {
    float4  *cones;      // [128]
    float   *qx, *qz;    // [128]
    int32_t *qi;         // [128]
};

// compute_vnew :368 keeps the first strictly-smaller distance in candidate order, i.e. the minimum
// of (distance, order index): candidates can be examined in any order.
struct cp_best { float len; int idx; v2 pt; bool any; };

// one candidate per lane: drop it when it lies inside the combined obstacle, else rank it
__device__ __forceinline__ void cp_rank(const cpent &ent, v2 des_v, const float4 *cones, int n_cones,
                                        bool have, v2 pt, int order, cp_best &B)
{
    if(have && !inside_pcr(cones, n_cones, pt)) {
        B.any = true;
        const v2 curr = vsub(pt, ent.pos);
        const float len = vlen(vsub(des_v, curr));
        if(len < B.len || (len == B.len && order < B.idx)) { B.len = len; B.idx = order; B.pt = curr; }
    }
}

// G_ClearPath_NewVelocity (clearpath.c:694) for one agent on one wave.
// dyn/stat: LDS arrays of 5 floats per neighbour (order matters).
//
// Candidate points (ray-pair intersections, :321, then the projections of des_v on every ray, :344)
// are produced 64 at a time, one ordered pair per lane; the pairs whose rays do meet are compacted
// into the queue and the expensive part -- is the point inside any cone? -- always runs on 64 real
// candidates per pass.
__device__ v2 clearpath_wave(const cpent &ent, v2 des_v, float *dyn, int n_dyn, float *stat,
                             int n_stat, const cp_scratch &S, int lane)
{
    if(n_dyn + n_stat == 0) return des_v;          // no obstacle: inside_pcr of nothing is false
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // at most 64 neighbours can be removed; the bound only guards against a NaN-poisoned input
    for(int guard = 0; guard < 66; guard++) {
        // ---- HRVOs for dynamic, VOs for static neighbours -> rays (rays_repr :291) -----------
        // lanes 0..31 dynamic, 32..63 static; same_position neighbours are skipped (:216-246)
        bool isdyn = lane < 32;
        int  k = isdyn ? lane : lane - 32;
        bool have = isdyn ? (k < n_dyn) : (k < n_stat);
        cpent nb; nb.pos = mkv(0, 0); nb.vel = mkv(0, 0); nb.radius = 0;
        if(have) {
            const float *src = (isdyn ? dyn : stat) + 5 * k;
            nb.pos = mkv(src[0], src[1]); nb.vel = mkv(src[2], src[3]); nb.radius = src[4];
        }
        bool use = have && !(vlen(vsub(nb.pos, ent.pos)) < CP_EPS);
        v2 apex = mkv(0, 0), left = mkv(0, 0), right = mkv(0, 0);
        float sl = 0.0f, sr = 0.0f;
        if(use) make_cone(ent, nb, isdyn, apex, left, right, sl, sr);
        uint64_t m = __ballot(use);
        int slot = __popcll(m & lt_mask);                      // hrvos first, then vos, in order
        const int n_cones = __popcll(m);
        const int n_rays = 2 * n_cones;
        wave_sync();
        if(use) {
            S.cones[2 * slot]     = make_float4(apex.x, apex.z, sl, sr);
            S.cones[2 * slot + 1] = make_float4(left.x, left.z, right.x, right.z);
        }
        wave_sync();

        // des_v admissible as it is?  lane = cone
        const v2 des_ws = vadd(ent.pos, des_v);
        bool in = false;
        if(lane < n_cones) in = cone_contains(S.cones[2 * lane], S.cones[2 * lane + 1], des_ws);
        if(!__any(in))
            return des_v;

        cp_best B; B.len = __builtin_inff(); B.idx = 0x7fffffff; B.pt = mkv(0, 0); B.any = false;
        int qn = 0;                                            // pending candidates (wave uniform)
        const int npairs = n_rays * n_rays;
        const float inv_nr = 1.0f / (float)n_rays;
        for(int p0 = 0; p0 < npairs + n_rays; p0 += 64) {
            const int p = p0 + lane;
            bool ok = false;
            v2 pt = mkv(0, 0);
            if(p < npairs) {
                // (i, j) = divmod(p, n_rays): float estimate + one correction step (p < 2^14)
                int i = (int)((float)p * inv_nr);
                int j = p - i * n_rays;
                if(j < 0) { i--; j += n_rays; }
                if(j >= n_rays) { i++; j -= n_rays; }
                if(i != j) {
                    const float4 Ai = S.cones[i & ~1], Bi = S.cones[i | 1];
                    const float4 Aj = S.cones[j & ~1], Bj = S.cones[j | 1];
                    const bool ri = i & 1, rj = j & 1;
                    ok = ray_isect(mkv(Ai.x, Ai.y), ri ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), ri ? Ai.w : Ai.z,
                                   mkv(Aj.x, Aj.y), rj ? mkv(Bj.z, Bj.w) : mkv(Bj.x, Bj.y), rj ? Aj.w : Aj.z,
                                   pt);
                }
            }else if(p < npairs + n_rays) {
                const int i = p - npairs;
                const float4 Ai = S.cones[i & ~1], Bi = S.cones[i | 1];
                const v2 dir = (i & 1) ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), point = mkv(Ai.x, Ai.y);
                const float len = vdot(dir, des_v);
                pt = vadd(point, vscale(dir, len));
                ok = true;
            }
            const uint64_t mk = __ballot(ok);
            if(ok) {
                const int at = qn + __popcll(mk & lt_mask);
                S.qx[at] = pt.x; S.qz[at] = pt.z; S.qi[at] = p;
            }
            qn += __popcll(mk);
            wave_sync();
            if(qn >= 64) {
                qn -= 64;
                cp_rank(ent, des_v, S.cones, n_cones, true, mkv(S.qx[qn + lane], S.qz[qn + lane]),
                        S.qi[qn + lane], B);
                wave_sync();
            }
        }
        if(qn > 0) {
            const bool mine = lane < qn;
            cp_rank(ent, des_v, S.cones, n_cones, mine, mine ? mkv(S.qx[lane], S.qz[lane]) : mkv(0, 0),
                    mine ? S.qi[lane] : 0, B);
        }
        if(__any(B.any)) {
            float key = B.len; int idx = B.idx;
            wave_argmin(key, idx);
            if(!(key < __builtin_inff())) return mkv(0.0f, 0.0f);   // only NaN distances: ret stays 0
            int owner = __ffsll((unsigned long long)__ballot(B.idx == idx && B.len == key)) - 1;
            return mkv(__shfl(B.pt.x, owner), __shfl(B.pt.z, owner));
        }

        // ---- no admissible point: remove_furthest (:390) and retry while both lists non-empty
        float dist = -__builtin_inff();
        int   ord = 0x7fffffff;                 // dyn entries precede stat entries in the scan
        if(have) {
            dist = vlen(vsub(ent.pos, nb.pos));
            ord = isdyn ? k : 32 + k;
        }
        // first strict maximum in scan order == min over (-dist, ord)
        float nk = -dist; int ni = ord;
        if(!have || !(dist == dist)) { nk = __builtin_inff(); }   // NaN never passes `len > max_dist`
        wave_argmin(nk, ni);
        wave_sync();
        if(nk < __builtin_inff() && lane == 0) {
            if(ni < 32) { n_dyn--;  for(int q = 0; q < 5; q++) dyn[5 * ni + q] = dyn[5 * n_dyn + q]; }
            else        { int s = ni - 32; n_stat--; for(int q = 0; q < 5; q++) stat[5 * s + q] = stat[5 * n_stat + q]; }
        }
        wave_sync();
        n_dyn = __shfl(n_dyn, 0);
        n_stat = __shfl(n_stat, 0);
        if(!(n_dyn > 0 && n_stat > 0))
            return mkv(0.0f, 0.0f);
    }
    return mkv(0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// cohesion_force (movement.c:1653): one thread per flock member
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool state_is_still(int s)
{
    return s == NAVHIP_STATE_ARRIVED || s == NAVHIP_STATE_WAITING;     // ent_still, movement.c:652
}
__device__ __forceinline__ bool state_uses_point_seek(int s)
{
    return s == NAVHIP_STATE_MOVING || s == NAVHIP_STATE_SURROUND_ENTITY
        || s == NAVHIP_STATE_ENTER_ENTITY_RANGE;
}

// 2^(j/64), j = 0..63, correctly rounded doubles
__constant__ double c_exp2_64[64] = {
    0x1p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92dep+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cdp+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e5p+0,
    0x1.9c49182a3f09p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e454p+0, 0x1.fa7c1819e90d8p+0};

// (float)exp((double)a) for a in (-inf, ~88]: the reference evaluates libm's double exp on a float
// argument and rounds to float (movement.c:1671,1731).  Table-driven double evaluation,
// exp(a) = 2^(k/64) * exp(r), |r| <= ln2/128, degree-5 polynomial: < 2 ulp in double, so the float
// rounding agrees with a correctly rounded exp except with probability ~1e-8 per call.  tab =
// 64-entry table in LDS.  k = rint(x * 64/ln2) falls out of the low
// mantissa bits of x * 64/ln2 + 1.5 * 2^52 (one FMA), and no final select is needed -- the clamped
// argument -104 gives 6.8e-46, which the f64 -> f32 conversion rounds to +0 like every value below
// half the smallest denormal (the true cut-off is a = -103.972).  The final scaling by 2^(k>>6) is
// an integer add on the exponent field (the result stays a normal double for every argument in
// [-104, 89]).  Checked against glibc's exp on 3e8 random arguments in [-110, 6], 3e8 in [-21, 89]
// and on every float in [-104.5, -102]: no mismatch.
__device__ __forceinline__ float exp_f32_magic(float a, const double *tab)
{
    const double x = (double)fmaxf(a, -104.0f);
    const double z = __builtin_fma(x, 0x1.71547652b82fep+6, 0x1.8p52);   // 64/ln2
    const double kd = z - 0x1.8p52;
    const int k = (int)__double_as_longlong(z);
    double r = __builtin_fma(-kd, 0x1.62e42fefa0000p-7, x);              // ln2/64, high part
    r = __builtin_fma(-kd, 0x1.cf79abc9e3b3ap-46, r);                    //         low part
    double p = __builtin_fma(r, 1.0 / 120, 1.0 / 24);
    p = __builtin_fma(p, r, 1.0 / 6);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double v = tab[k & 63] * p;
    const long long bits = __double_as_longlong(v) + ((long long)(k >> 6) << 52);
    return (float)__longlong_as_double(bits);
}

// float t = (len - 50.0f*0.75) / 50.0f of movement.c:1668 (the reference evaluates it in double and
// rounds to float).  For len >= 16 the f32 subtraction is exact and the division by 50 as
// reciprocal multiply + one FMA correction (Markstein) reproduces the double-then-float result for
// EVERY float in [16, 8192) (checked exhaustively on the CPU); beyond that the weight is 0 anyway.
__device__ __forceinline__ float cohesion_t_f32(float len)
{
    const float r50f = 1.0f / 50.0f;
    const float x = len - 37.5f;
    const float q0 = x * r50f;
    return __builtin_fmaf(__builtin_fmaf(-q0, 50.0f, x), r50f, q0);
}

// the same through double, for len < 16 where the f32 subtraction may round
__device__ __forceinline__ float cohesion_t_f64(float len)
{
    const double r50 = 1.0 / 50.0;
    const double x = (double)len - (double)50.0f * 0.75;
    const double q0 = x * r50;
    return (float)__builtin_fma(__builtin_fma(-q0, 50.0, x), r50, q0);
}

#define COH_BINS 257       /* 256 Morton blocks + 1 bin for members that take no cohesion force */
// k_coh_plan: wave_off[f] = number of 16-member (COH_APW) waves of the flocks before f (exclusive
// scan); one workgroup, chunked.  Two forms:
//  * bin_start != nullptr (fresh grouping, runs after the bin scan): ceil(active members / 16), the
//    active members being the bins before the flock's last one;
//  * bin_start == nullptr (grouping of the previous tick): ceil(flock size / 16) -- every member gets
//    a lane, whatever its state was when the grouping was made -- and *perm_valid = the grouping
//    was built for exactly these flock offsets (saved_offs).
__global__ __launch_bounds__(256) void k_coh_plan(const int32_t *bin_start, const int32_t *flock_offsets,
                                                  const int32_t *saved_offs, int n_flocks,
                                                  int32_t *wave_off, int32_t *perm_valid)
{
    __shared__ int32_t wsum[4];
    __shared__ int32_t carry;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if(t == 0) carry = 0;
    __syncthreads();
    bool same = true;
    for(int base = 0; base < n_flocks; base += 256) {
        const int f = base + t;
        int32_t v = 0;
        if(f < n_flocks) {
            if(bin_start) {
                v = (bin_start[f * COH_BINS + 256] - bin_start[f * COH_BINS] + 15) >> 4;   // COH_APW
            }else{
                const int32_t b = flock_offsets[f], e = flock_offsets[f + 1];
                v = (e - b + 15) >> 4;
                same = same && saved_offs[f] == b && saved_offs[f + 1] == e;
            }
        }
        int32_t incl = v;
#pragma unroll
        for(int d = 1; d < 64; d <<= 1) {
            int32_t o = __shfl_up(incl, d);
            if(lane >= d) incl += o;
        }
        if(lane == 63) wsum[w] = incl;
        __syncthreads();
        int32_t woff = 0;
        for(int k = 0; k < w; k++) woff += wsum[k];
        const int32_t excl = carry + woff + incl - v;
        if(f < n_flocks) wave_off[f] = excl;
        __syncthreads();
        if(t == 255) carry = excl + v;
        __syncthreads();
    }
    if(t == 0) wave_off[n_flocks] = carry;
    if(perm_valid) {
        const int ok = __syncthreads_and(same);
        if(t == 0) *perm_valid = ok;
    }
}

// k_coh_bin / k_coh_scatter: per-tick lane assignment of the cohesion launch.  Which thread handles
// which member is free (each member's sum only depends on the flock's member ORDER, which the walk
// keeps), so the members of a flock are regrouped by 256-wu map blocks (16x16) in Morton order:
// the 64 members of a wave are then close together and the wave can skip, exactly, every flock
// mate that is too far from ALL of them to carry a non-zero weight (see k_cohesion).
//   bin = flock * COH_BINS + morton(block);   perm[] = CSR entries ordered by bin (counting sort;
//   the order inside a bin is whatever the atomics give -- it only moves members between lanes).
// Members that take no cohesion force this tick (outside the work range, not point seeking, combat
// hold) go to the flock's last bin: the lanes in use are packed at the front and the trailing waves
// of a flock exit at once.
__device__ __forceinline__ int coh_bin_of(const nh_step_params &P, int g, int *flock_out)
{
    int lo = 0, hi = P.n_flocks;
    while(hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if(P.flock_offsets[mid] <= g) lo = mid; else hi = mid;
    }
    *flock_out = lo;
    const int m = P.flock_members[g];
    const bool act = m >= P.work_begin && m < P.work_end && state_uses_point_seek(P.state[m])
                  && !(P.flags[m] & NAVHIP_ENTITY_FLAG_COMBAT_HELD);
    if(!act) return lo * COH_BINS + 256;
    const float ox = (float)P.grid.origin_x * (1.0f / 256.0f), oz = (float)P.grid.origin_y * (1.0f / 256.0f);
    // 256-wu blocks, 16 x 16 of them before the pattern repeats (a flock spread over more than
    // 4096 wu merely shares bins: the grouping is a locality heuristic, never a correctness matter)
    const int bx = (int)floorf((P.pos_xz[2 * m] - ox) * (1.0f / 256.0f)) & 15;
    const int bz = (int)floorf((P.pos_xz[2 * m + 1] - oz) * (1.0f / 256.0f)) & 15;
    int mo = 0;
#pragma unroll
    for(int k = 0; k < 4; k++) mo |= (((bx >> k) & 1) << (2 * k)) | (((bz >> k) & 1) << (2 * k + 1));
    return lo * COH_BINS + mo;
}

// (The idle members of a flock all share one bin: on a rank that steps one slab of a large job that
// is most members, and one atomic per member on the same address serialises -- 110 us for 800 k
// members.  Lanes of a wave that hit the same idle bin are counted by ONE atomic of the first of
// them; active members are spread over 256 bins and use plain atomics.)
__device__ __forceinline__ int coh_grouped_add(int32_t *counter, int bin, bool idle)
{
    const int lane = threadIdx.x & 63;
    int slot = 0;
    unsigned long long rem = __ballot(idle);
    while(rem) {
        const int leader = __ffsll((long long)rem) - 1;
        const int b0 = __shfl(bin, leader);
        const unsigned long long m = __ballot(idle && bin == b0);
        int base = 0;
        if(lane == leader) base = atomicAdd(&counter[b0], __popcll(m));
        base = __shfl(base, leader);
        if(idle && bin == b0) slot = base + __popcll(m & ((1ull << lane) - 1ull));
        rem &= ~m;
    }
    if(!idle) slot = atomicAdd(&counter[bin], 1);
    return slot;
}

__global__ __launch_bounds__(256) void k_coh_bin(nh_step_params P, int32_t *bin_of, int32_t *bin_count,
                                                 int32_t *saved_offs)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if(saved_offs)
        for(int f = g; f <= P.n_flocks; f += gridDim.x * 256) saved_offs[f] = P.flock_offsets[f];
    if(g >= P.flock_offsets[P.n_flocks]) return;
    int f;
    const int bin = coh_bin_of(P, g, &f);
    bin_of[g] = bin;
    coh_grouped_add(bin_count, bin, bin - f * COH_BINS == 256);
}

__global__ __launch_bounds__(256) void k_coh_scatter(nh_step_params P, const int32_t *bin_of,
                                                     const int32_t *bin_start, int32_t *bin_fill,
                                                     int32_t *perm)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if(g >= P.flock_offsets[P.n_flocks]) return;
    const int bin = bin_of[g];
    perm[bin_start[bin] + coh_grouped_add(bin_fill, bin, bin % COH_BINS == 256)] = g;
}

// exp(-6 t) rounds to +0 in float once the distance exceeds 904 wu (t >= 17.33); COH_FAR leaves a
// margin for the roundings of the box test
#define COH_FAR 906.0f

// k_cohesion: one WAVE (= one 64-thread workgroup) per COH_APW = 16 members of ONE flock (perm[]
// order); FOUR lanes share a member (lane = member << 2 | sub).  A wave never straddles two flocks.
//
// The flock's member positions are staged through LDS 256 at a time (coalesced gather); a
// lane-parallel pre-pass drops the staged members that lie more than COH_FAR from the bounding box
// of the wave's own members (their weight is exactly 0 for every lane, and adding +-0 leaves the
// never-negative-zero running sums unchanged) and queues the survivors IN MEMBER ORDER.  Queue entry
// k belongs to sub-lane k & 3: each lane evaluates the expensive part (distance -> t -> exp, ~36
// instructions per entry) for a quarter of the entries only, eight at a time, two per packed f32
// instruction.  The float sums are order dependent, so the products are then added strictly in
// entry order: the sub-lane that owns entry k broadcasts its product to the quad (DPP quad_perm as
// an operand of the add) and all four lanes keep the same running sum.
//
// Why four lanes per member: with one lane per member the launch was 1 600 long waves on 1 024 SIMDs
// (one or two per SIMD, 25 000 instructions each, nothing to hide a wave's own scalar/LDS/
// transcendental issue slots behind); 6 500 shorter waves fill every SIMD six deep for about the
// same instruction total, and a 16-member box is tighter than a 64-member one.
#define COH_APW 16
#define COH_QS  72            /* queue slots per sub-lane: (256 staged + 31 carried) / 4 */
#ifndef COH_NP
#define COH_NP  2             /* entry pairs per lane and batch: a batch is COH_G = 8 * COH_NP entries
                                 (2 measured 1.3 % faster per tick than 4: fewer VGPRs, more waves) */
#endif
#define COH_G   (8 * COH_NP)

__device__ __forceinline__ float quad_bcast(float v, int sub)
{
    // quad_perm:[sub,sub,sub,sub]
    switch(sub) {
    case 0:  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00, 0xf, 0xf, true));
    case 1:  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55, 0xf, 0xf, true));
    case 2:  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xaa, 0xf, 0xf, true));
    default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xff, 0xf, 0xf, true));
    }
}

// COH_G queue entries starting at entry jj (a multiple of COH_G): this lane's 2 * COH_NP are local
// slots jj/4 .. of its own quarter.  TAIL: entries >= n_valid are padding (weight 0).
template <bool TAIL>
__device__ __forceinline__ void coh_batch(const float *qx, const float *qz, const double *tab, int sub,
                                          int jj, int n_valid, int self_k, v2 me, float &comx, float &comz)
{
    const f2 mex = {me.x, me.x}, mez = {me.z, me.z};
    const int lo = sub * COH_QS + (jj >> 2);
    constexpr int NP = COH_NP;
    f2 X[NP], Z[NP], ss[NP], ln[NP], tt[NP], W[NP];
    bool close = false, odd = false;
#pragma unroll
    for(int v = 0; v < NP / 2; v++) {
        const f4 xa = *(const f4*)&qx[lo + 4 * v], za = *(const f4*)&qz[lo + 4 * v];
        X[2 * v] = f2{xa.x, xa.y}; X[2 * v + 1] = f2{xa.z, xa.w};
        Z[2 * v] = f2{za.x, za.y}; Z[2 * v + 1] = f2{za.z, za.w};
    }
#pragma unroll
    for(int u = 0; u < NP; u++) {
        const f2 dx = X[u] - mex, dz = Z[u] - mez;
        ss[u] = dx * dx + dz * dz;
        // sqrt_rn_normal on both halves
        f2 r = {__builtin_amdgcn_sqrtf(ss[u].x), __builtin_amdgcn_sqrtf(ss[u].y)};
        const f2 rm = {__int_as_float(__float_as_int(r.x) - 1), __int_as_float(__float_as_int(r.y) - 1)};
        const f2 rp = {__int_as_float(__float_as_int(r.x) + 1), __int_as_float(__float_as_int(r.y) + 1)};
        const f2 em = __builtin_elementwise_fma(-rm, r, ss[u]);
        const f2 ep = __builtin_elementwise_fma(-rp, r, ss[u]);
        r.x = (em.x <= 0.0f) ? rm.x : r.x;  r.y = (em.y <= 0.0f) ? rm.y : r.y;
        r.x = (ep.x > 0.0f) ? rp.x : r.x;   r.y = (ep.y > 0.0f) ? rp.y : r.y;
        ln[u] = r;
        // outside [2^-90, 2^90] (or NaN) and not exactly 0: leave it to the general IEEE expansion
        odd |= !(ss[u].x >= 0x1p-90f && ss[u].x <= 0x1p90f) && ss[u].x != 0.0f;
        odd |= !(ss[u].y >= 0x1p-90f && ss[u].y <= 0x1p90f) && ss[u].y != 0.0f;
    }
    if(__any(odd)) {
        asm volatile("" ::: "memory");        // keep the expansion out of the common path
#pragma unroll
        for(int u = 0; u < NP; u++) ln[u] = f2{__builtin_sqrtf(ss[u].x), __builtin_sqrtf(ss[u].y)};
    }
#pragma unroll
    for(int u = 0; u < NP; u++) {
        // cohesion_t_f32 on both halves
        const f2 x = ln[u] - 37.5f;
        const f2 q0 = x * (1.0f / 50.0f);
        const f2 inner = __builtin_elementwise_fma(-q0, f2{50.0f, 50.0f}, x);
        tt[u] = __builtin_elementwise_fma(inner, f2{1.0f / 50.0f, 1.0f / 50.0f}, q0);
        close |= ln[u].x < 16.0f || ln[u].y < 16.0f;
    }
    if(__any(close)) {                // rare unless the flock is one dense cluster
        asm volatile("" ::: "memory");
#pragma unroll
        for(int u = 0; u < NP; u++) {
            if(ln[u].x < 16.0f) tt[u].x = cohesion_t_f64(ln[u].x);
            if(ln[u].y < 16.0f) tt[u].y = cohesion_t_f64(ln[u].y);
        }
    }
    const int k0 = jj + sub;                  // entry number of this lane's first entry; then +4 each
#pragma unroll
    for(int u = 0; u < NP; u++) {
        const f2 a = tt[u] * -6.0f;
        float w0 = exp_f32_magic(a.x, tab), w1 = exp_f32_magic(a.y, tab);
        // curr == uid is skipped by the reference: a zero weight adds +-0, which leaves the (never
        // negative-zero) running sum unchanged; so does the padding of the last batch
        const int ka = k0 + 8 * u, kb = ka + 4;
        if(ka == self_k || (TAIL && ka >= n_valid)) w0 = 0.0f;
        if(kb == self_k || (TAIL && kb >= n_valid)) w1 = 0.0f;
        W[u] = f2{w0, w1};
    }
    // products, then the ordered sums: entry jj + 4*i + s is held by sub-lane s as element i
#pragma unroll
    for(int u = 0; u < NP; u++) {
        const f2 px = X[u] * W[u], pz = Z[u] * W[u];
#pragma unroll
        for(int h = 0; h < 2; h++) {
            const float tx = h ? px.y : px.x, tz = h ? pz.y : pz.x;
#pragma unroll
            for(int sb = 0; sb < 4; sb++) {
                comx = comx + quad_bcast(tx, sb);
                comz = comz + quad_bcast(tz, sb);
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_cohesion(nh_step_params P, const int32_t *wave_off,
                                                 const int32_t *perm, const int32_t *perm_valid,
                                                 float *coh_xz)
{
    __shared__ double tab[64];
    // the members of the current tile that survive the box test, in member order (+ carry-over):
    // entry k lives in slot (k & 3) * COH_QS + (k >> 2)
    __shared__ __attribute__((aligned(16))) float qx[4 * COH_QS];
    __shared__ __attribute__((aligned(16))) float qz[4 * COH_QS];
    const int t = threadIdx.x, sub = t & 3;
    const int wv = blockIdx.x;
    if(wv >= wave_off[P.n_flocks]) return;
    tab[t] = c_exp2_64[t];
    // flock of this wave: binary search over the wave prefix (uniform)
    int f = 0;
    {
        int lo = 0, hi = P.n_flocks;
        while(hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if(wave_off[mid] <= wv) lo = mid; else hi = mid;
        }
        f = lo;
    }
    const float scaled_max_force = (float)((double)(0.75f / (float)P.hz) * 20.0);
    const int b = P.flock_offsets[f], e = P.flock_offsets[f + 1];
    const int gp = b + (wv - wave_off[f]) * COH_APW + (t >> 2);
    const bool mine = gp < e;
    // CSR entry of this quad's member (any permutation of the flock's entries serves; a grouping
    // made for other flock offsets is ignored)
    const bool use_perm = !perm_valid || *perm_valid != 0;
    const int g = mine ? (use_perm ? perm[gp] : gp) : -1;
    const int uid = mine ? P.flock_members[g] : -1;
    bool act = mine && uid >= P.work_begin && uid < P.work_end;
    if(act) act = state_uses_point_seek(P.state[uid]) && !(P.flags[uid] & NAVHIP_ENTITY_FLAG_COMBAT_HELD);
    if(!__syncthreads_or(act)) return;
    const v2 me = act ? mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]) : mkv(0.0f, 0.0f);
    // bounding box of the wave's active members
    float bx0 = act ? me.x : INFINITY, bx1 = act ? me.x : -INFINITY;
    float bz0 = act ? me.z : INFINITY, bz1 = act ? me.z : -INFINITY;
#pragma unroll
    for(int d = 4; d < 64; d <<= 1) {
        bx0 = fminf(bx0, __shfl_xor(bx0, d)); bx1 = fmaxf(bx1, __shfl_xor(bx1, d));
        bz0 = fminf(bz0, __shfl_xor(bz0, d)); bz1 = fmaxf(bz1, __shfl_xor(bz1, d));
    }
    const unsigned long long lt_mask = (1ull << t) - 1ull;
    float comx = 0.0f, comz = 0.0f;
    int self_k = -1;                                      // queue entry of this quad's own member
    int pend = 0;                                         // entries carried over from the last tile
    for(int jb = b; jb < e; jb += 256) {
        // ---- stage the tile behind the carry-over [0, pend)
        int ncnt = pend;
        const int gl = act ? g - jb : -1;                 // own slot in the unfiltered tile, if any
#pragma unroll
        for(int q = 0; q < 4; q++) {
            const int j = jb + q * 64 + t;
            bool keep = false;
            float2 c2 = make_float2(0.0f, 0.0f);
            if(j < e) {
                const int m = P.flock_members[j];
                c2 = make_float2(P.pos_xz[2 * m], P.pos_xz[2 * m + 1]);
                const float dx = fmaxf(fmaxf(bx0 - c2.x, c2.x - bx1), 0.0f);
                const float dz = fmaxf(fmaxf(bz0 - c2.y, c2.y - bz1), 0.0f);
                keep = !(dx * dx + dz * dz > COH_FAR * COH_FAR);       // NaN stays in
            }
            const unsigned long long mk = __ballot(keep);
            const int at = ncnt + __popcll(mk & lt_mask);
            if(keep) { const int sl = (at & 3) * COH_QS + (at >> 2); qx[sl] = c2.x; qz[sl] = c2.y; }
            // (an active member lies inside the wave's box, so it is always kept)
            const int at_self = __shfl(at, gl & 63);
            if((gl >> 6) == q) self_k = at_self;          // gl < 0 or >= 256 never matches q = 0..3
            ncnt += __popcll(mk);
        }
        const bool last = jb + 256 >= e;
        int cnt32 = ncnt & ~(COH_G - 1);                  // whole batches
        if(last && cnt32 < ncnt) {
            // pad the final batch with finite dummies (their weight is forced to 0)
            const int k = ncnt + t;
            if(k < cnt32 + COH_G) { const int sl = (k & 3) * COH_QS + (k >> 2); qx[sl] = 0.0f; qz[sl] = 0.0f; }
        }
        __syncthreads();
        if(act) {
            for(int jj = 0; jj < cnt32; jj += COH_G)
                coh_batch<false>(qx, qz, tab, sub, jj, ncnt, self_k, me, comx, comz);
            if(last && cnt32 < ncnt)
                coh_batch<true>(qx, qz, tab, sub, cnt32, ncnt, self_k, me, comx, comz);
        }
        if(!last) {
            // carry the last (< COH_G) entries over: entry k -> k - cnt32 keeps its sub-lane
            pend = ncnt - cnt32;
            float cx = 0.0f, cz = 0.0f;
            const int k = cnt32 + t;
            if(t < pend) { const int sl = (k & 3) * COH_QS + (k >> 2); cx = qx[sl]; cz = qz[sl]; }
            __syncthreads();
            if(t < pend) { const int sl = (t & 3) * COH_QS + (t >> 2); qx[sl] = cx; qz[sl] = cz; }
            self_k = (self_k >= cnt32) ? self_k - cnt32 : -1;
            __syncthreads();
        }
    }
    if(act && sub == 0) {
        const int count = (e - b) - 1;
        v2 ret = mkv(0.0f, 0.0f);
        if(count > 0) {
            const v2 cm = vscale(mkv(comx, comz), 1.0f / (float)count);
            ret = vtrunc(vsub(cm, me), scaled_max_force);
        }
        coh_xz[2 * uid] = ret.x;
        coh_xz[2 * uid + 1] = ret.z;
    }
}

// ---------------------------------------------------------------------------------------------
// k_agent_step: one wave per entity
// ---------------------------------------------------------------------------------------------
// waves (= agents) per workgroup of k_agent_step; 2 measured best (1: 0.490, 2: 0.483, 4: 0.494,
// 8: 0.521 ms/tick in one session, scripts/ab_lib.py)
#ifndef AG_WAVES
#define AG_WAVES 2
#endif
struct wave_lds {
    uint32_t ids30[128];                       // separation query result (cap 128, :1695)
    union {
        uint32_t ids10[512];                   // ClearPath neighbour query result (cap 512, :2779)
        float4   rays[128];                    // later: the combined obstacle, two float4 per cone
        float    sep[256];                     // earlier: separation terms (x, z)[128]
    } u;
    float dyn[32 * 5];
    float stat[32 * 5];
    int32_t  d2_30[128];                       // squared fixed-point distances of the r=30 hits
    uint32_t ids10d[128];                      // r=10 list derived from the r=30 list
};

// separation_force, movement.c:1690.  ids30/n30 already gathered; wave-uniform result.
__device__ v2 separation_wave(const nh_step_params &P, int uid, v2 me, float my_radius,
                              uint32_t my_flags, const uint32_t *ids30, int n30, float *sep,
                              float scaled_max_force, int lane, const double *exp_tab)
{
    if(n30 == 0) return mkv(0.0f, 0.0f);
    for(int base = 0; base < n30; base += 64) {
        int k = base + lane;
        if(k < n30) {
            uint32_t curr = ids30[k];
            const float4 ra = P.grid.rec[2 * curr];                         // {pos, radius, flags}
            uint32_t fl = __float_as_uint(ra.w);
            v2 term = mkv(0.0f, 0.0f);
            bool skip = (curr == (uint32_t)uid) || !(fl & NAVHIP_ENTITY_FLAG_MOVABLE)
                     || ((my_flags & NAVHIP_ENTITY_FLAG_AIR) != (fl & NAVHIP_ENTITY_FLAG_AIR));
            if(!skip) {
                v2 cp = mkv(ra.x, ra.y);
                float radius = my_radius + ra.z + 0.0f;                     // SEPARATION_BUFFER_DIST
                v2 diff = vsub(cp, me);
                float len = vlen(diff);
                if(!(len < CP_EPS)) {
                    float t = __fdiv_rn(len - radius * 0.85f, len);
                    float scale = exp_f32_magic(fminf(-20.0f * t, 40.0f), exp_tab);
                    term = vscale(diff, scale);
                }
            }
            ((f2*)sep)[k] = f2{term.x, term.z};
        }
    }
    wave_sync();
    // ret += diff, strictly in candidate order (every lane evaluates the same chain; x and z ride
    // in one packed add)
    f2 acc = {0.0f, 0.0f};
    for(int k = 0; k < n30; k++)
        acc = acc + ((const f2*)sep)[k];
    wave_sync();
    const v2 ret = vscale(mkv(acc.x, acc.y), -1.0f);
    return vtrunc(ret, scaled_max_force);
}

// arrive_force_point, movement.c:1546
__device__ __forceinline__ v2 arrive_force(v2 me, v2 vel, v2 target, v2 vdes, bool los,
                                           float max_speed, int hz, float scaled_max_force)
{
    v2 desired;
    if(los) {
        desired = vsub(target, me);
        float distance = vlen(desired);
        desired = vnormal(desired);
        desired = vscale(desired, max_speed / (float)hz);
        if(distance < 10.0f)
            desired = vscale(desired, distance / 10.0f);
    }else{
        desired = vscale(vdes, max_speed / (float)hz);
    }
    return vtrunc(vsub(desired, vel), scaled_max_force);
}

// nullify_impass_components, movement.c:1831
__device__ __forceinline__ v2 nullify_impass(const nh_step_params &P, int layer, v2 pos, v2 f)
{
    bool on_blocked = pos_blocked(P, layer, pos.x, pos.z);
    if(f.x > 0 && (!pos_pathable(P, layer, pos.x + 4.0f, pos.z)
               || (!on_blocked && pos_blocked(P, layer, pos.x + 4.0f, pos.z)))) f.x = 0.0f;
    if(f.x < 0 && (!pos_pathable(P, layer, pos.x - 4.0f, pos.z)
               || (!on_blocked && pos_blocked(P, layer, pos.x - 4.0f, pos.z)))) f.x = 0.0f;
    if(f.z > 0 && (!pos_pathable(P, layer, pos.x, pos.z + 4.0f)
               || (!on_blocked && pos_blocked(P, layer, pos.x, pos.z + 4.0f)))) f.z = 0.0f;
    if(f.z < 0 && (!pos_pathable(P, layer, pos.x, pos.z - 4.0f)
               || (!on_blocked && pos_blocked(P, layer, pos.x, pos.z - 4.0f)))) f.z = 0.0f;
    return f;
}

// The five tile probes of nullify_impass_components (own tile, +-4 wu in x and z), issued together
// at the start of the step instead of one dependent load after another.
struct tile_probes { bool path[5], blk[5]; };     // 0 self, 1 x+4, 2 x-4, 3 z+4, 4 z-4

// One thread per agent (k_agent_pre): ten booleans packed as bits 0-4 pathable, 5-9 blocked.
__device__ __forceinline__ uint32_t probe_tiles_bits(const nh_step_params &P, int layer, v2 pos)
{
    const float px[5] = {pos.x, pos.x + 4.0f, pos.x - 4.0f, pos.x, pos.x};
    const float pz[5] = {pos.z, pos.z, pos.z, pos.z + 4.0f, pos.z - 4.0f};
    const uint8_t *cost = P.map.layers[layer].cost;
    const uint16_t *bl = P.map.layers[layer].blockers;
    uint32_t bits = 0;
#pragma unroll
    for(int i = 0; i < 5; i++) {
        tiledesc t;
        if(!tile_for_point(P, px[i], pz[i], t)) continue;
        const size_t idx = ((size_t)(t.chunk_r * P.map.w + t.chunk_c) << 12) + t.tile_r * 64 + t.tile_c;
        if(cost[idx] != NAVHIP_COST_IMPASSABLE) bits |= 1u << i;
        if(bl && bl[idx] > 0) bits |= 32u << i;
    }
    return bits;
}

__device__ __forceinline__ tile_probes unpack_probes(uint32_t bits)
{
    tile_probes T;
#pragma unroll
    for(int i = 0; i < 5; i++) { T.path[i] = (bits >> i) & 1; T.blk[i] = (bits >> (5 + i)) & 1; }
    return T;
}

__device__ __forceinline__ v2 nullify_impass_pre(const tile_probes &T, v2 f)
{
    const bool on_blocked = T.blk[0];
    if(f.x > 0 && (!T.path[1] || (!on_blocked && T.blk[1]))) f.x = 0.0f;
    if(f.x < 0 && (!T.path[2] || (!on_blocked && T.blk[2]))) f.x = 0.0f;
    if(f.z > 0 && (!T.path[3] || (!on_blocked && T.blk[3]))) f.z = 0.0f;
    if(f.z < 0 && (!T.path[4] || (!on_blocked && T.blk[4]))) f.z = 0.0f;
    return f;
}

// find_neighbours, movement.c:2768: classify the r=10 query result into dynamic / static lists
__device__ void classify_neighbours(const nh_step_params &P, int uid, uint32_t my_flags,
                                    const uint32_t *ids10, int n10, float *dyn, int &n_dyn,
                                    float *stat, int &n_stat, int lane)
{
    n_dyn = 0; n_stat = 0;
    for(int base = 0; base < n10; base += 64) {
        int k = base + lane;
        int cls = 0;                              // 0 skip, 1 dynamic, 2 static
        float rec[5] = {0, 0, 0, 0, 0};
        if(k < n10) {
            uint32_t curr = ids10[k];
            const float4 ra = P.grid.rec[2 * curr];                         // {pos, radius, flags}
            uint32_t fl = __float_as_uint(ra.w);
            float rad = ra.z;
            bool skip = (curr == (uint32_t)uid) || !(fl & NAVHIP_ENTITY_FLAG_MOVABLE) || (rad == 0.0f)
                     || ((my_flags & NAVHIP_ENTITY_FLAG_AIR) != (fl & NAVHIP_ENTITY_FLAG_AIR));
            if(!skip) {
                const float4 rb = P.grid.rec[2 * curr + 1];                 // {vel, state, -}
                v2 vel = mkv(rb.x, rb.y);
                rec[0] = ra.x; rec[1] = ra.y;
                rec[4] = rad;
                if(state_is_still((int)__float_as_uint(rb.z)) || vlen(vel) < 0.3f) {   // CLEARPATH_STILL_SPEED
                    cls = 2;                       // static: velocity forced to zero (:2817)
                }else{
                    cls = 1;
                    rec[2] = vel.x; rec[3] = vel.z;
                }
            }
        }
        uint64_t md = __ballot(cls == 1), ms = __ballot(cls == 2);
        uint64_t lt = (1ull << lane) - 1ull;
        if(cls == 1) {
            int p = n_dyn + __popcll(md & lt);
            if(p < 32) { for(int q = 0; q < 5; q++) dyn[5 * p + q] = rec[q]; }     // MAX_NEIGHBOURS
        }else if(cls == 2) {
            int p = n_stat + __popcll(ms & lt);
            if(p < 32) { for(int q = 0; q < 5; q++) stat[5 * p + q] = rec[q]; }
        }
        n_dyn = min(32, n_dyn + __popcll(md));
        n_stat = min(32, n_stat + __popcll(ms));
    }
    wave_sync();
}

// The r = 10 neighbour query (find_neighbours, movement.c:2779) as a filter of the r = 30 query of
// the same agent (separation_force, :1695).  Valid when the r = 30 list is COMPLETE (its cap of 128
// did not bind) and both queries scan in the same mode: every entity within 10 is then in the list,
// and two entities keep their relative order in any query that returns both (the visiting key --
// coarse block, fine row, cell, slot -- does not depend on the query).  Returns the derived count,
// or -1 when the r = 10 query has to run on its own.  Must see the list BEFORE filter_garrisoned
// permutes it.
__device__ int derive_r10(const nh_grid &G, v2 me, const uint32_t *ids30, const int32_t *d2_30,
                          int n30raw, uint32_t *out, int lane)
{
    if(n30raw >= 128) return -1;
    // an r=30 box covers at most 5x5 cells: on a grid of more than 33 cells neither query can take
    // the wide path (25 * 4 < 34 * 3) and both scan in block order
    if((int64_t)G.grid_w * G.grid_h <= 33) {
        const int32_t icx = bg_scale(me.x), icy = bg_scale(me.z);
        sp_extent E30, E10;
        const bool ok30 = sp_query_extent(G, icx, icy, bg_scale(30.0f), E30);
        const bool ok10 = sp_query_extent(G, icx, icy, bg_scale(10.0f), E10);
        if(!ok30 || !ok10 || E30.wide != E10.wide) return -1;
    }
    const int32_t ir10 = bg_scale(10.0f);
    const int32_t lim = ir10 * ir10;
    int written = 0;
    for(int base = 0; base < n30raw; base += 64) {
        const int k = base + lane;
        const bool hit = k < n30raw && d2_30[k] <= lim;
        const uint64_t m = __ballot(hit);
        const int p = written + __popcll(m & ((1ull << lane) - 1ull));
        if(hit) out[p] = ids30[k];
        written += __popcll(m);
    }
    wave_sync();
    return written;
}

// What the scalar pre-pass hands to the wave-per-agent kernel (24 bytes per entity)
enum { AM_IDLE = 0,        // still or combat held: velocity 0, no neighbour work
       AM_ZERO_VPREF,      // turning / formation assignment not ready: vpref = 0, ClearPath still runs
       AM_POINT_SEEK, AM_ENEMY_SEEK, AM_FORM_CELL, AM_FORM_POINT,
       AM_UNSUPPORTED };   // formation state without formation inputs
struct nh_pre_rec {
    float    vdes[2];
    float    arrive[2];    // the arrive term of the state's steering force, already truncated
    uint16_t probes;       // probe_tiles_bits
    uint8_t  status;
    uint8_t  mode;
    float    vpref_cap;    // speed / hz      (movement.c:1880 and friends)
    float    vel_cap;      // max_speed / hz  (movement.c:3464)
    uint32_t pad;
};
static_assert(sizeof(nh_pre_rec) == 32, "nh_pre_rec");

// k_agent_pre: one THREAD per entity.  Everything of move_velocity_work that needs no neighbour
// list and is the same for all 64 lanes of a wave-per-agent kernel -- desired direction (flow-field
// sampling), the arrive force of the state's steering behaviour, the five tile probes of
// nullify_impass_components -- is evaluated here with all lanes busy.
__global__ __launch_bounds__(256) void k_agent_pre(nh_step_params P, nh_pre_rec *pre, nh_step_outs O)
{
    const int uid = P.work_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if(uid >= P.work_end) return;
    const int state = P.state[uid];
    const uint32_t my_flags = P.flags[uid];
    nh_pre_rec R;
    R.vdes[0] = R.vdes[1] = R.arrive[0] = R.arrive[1] = 0.0f;
    R.probes = 0; R.status = 0; R.mode = AM_IDLE; R.pad = 0;
    R.vpref_cap = P.speed[uid] / (float)P.hz;
    R.vel_cap = P.max_speed[uid] / (float)P.hz;
    if(!state_is_still(state) && !(my_flags & NAVHIP_ENTITY_FLAG_COMBAT_HELD)) {
        const v2 me = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
        const v2 vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
        const float my_radius = P.radius[uid], max_speed = P.max_speed[uid];
        const int flock = P.flock[uid], hz = P.hz;
        const float scaled_max_force = (float)((double)(0.75f / (float)hz) * 20.0);   // SCALED_MAX_FORCE
        const int layer = nav_layer_for(my_flags, my_radius);
        uint32_t status = 0;
        v2 vdes = mkv(0.0f, 0.0f), arrive = mkv(0.0f, 0.0f);
        const bool form = state == NAVHIP_STATE_MOVING_IN_FORMATION || state == NAVHIP_STATE_ARRIVING_TO_CELL;
        if(state == NAVHIP_STATE_TURNING) {
            R.mode = AM_ZERO_VPREF;
        }else if(state == NAVHIP_STATE_SEEK_ENEMIES || state_uses_point_seek(state)) {
            vdes = load_vdes(P, uid, flock, me, status);
            if(state_uses_point_seek(state)) {
                const bool los = P.has_dest_los[uid] != 0;
                const v2 target = (flock >= 0) ? mkv(P.flock_target_xz[2 * flock], P.flock_target_xz[2 * flock + 1]) : me;
                arrive = arrive_force(me, vel, target, vdes, los, max_speed, hz, scaled_max_force);
                R.mode = AM_POINT_SEEK;
            }else{
                // arrive_force_enemies, movement.c:1593
                v2 desired = vscale(vdes, max_speed / (float)hz);
                arrive = vtrunc(vsub(desired, vel), scaled_max_force);
                R.mode = AM_ENEMY_SEEK;
            }
        }else if(P.form_ready && form) {
            if(!P.form_ready[uid]) {
                R.mode = AM_ZERO_VPREF;
            }else{
                vdes = load_vdes(P, uid, flock, me, status);
                if(state == NAVHIP_STATE_ARRIVING_TO_CELL) {
                    // arrive_force_cell :1574 (no velocity term, no truncation)
                    const v2 cell = mkv(P.cell_pos_xz[2 * uid], P.cell_pos_xz[2 * uid + 1]);
                    v2 desired = vsub(cell, me);
                    float distance = vlen(desired);
                    if(distance < 10.0f) desired = vscale(desired, distance / 10.0f);
                    else                 desired = vscale(vdes, max_speed / (float)hz);
                    arrive = desired;
                    R.mode = AM_FORM_CELL;
                }else{
                    const bool los = P.has_dest_los[uid] != 0;
                    const v2 target = (flock >= 0) ? mkv(P.flock_target_xz[2 * flock], P.flock_target_xz[2 * flock + 1]) : me;
                    arrive = arrive_force(me, vel, target, vdes, los, max_speed, hz, scaled_max_force);
                    R.mode = AM_FORM_POINT;
                }
            }
        }else{
            R.mode = AM_UNSUPPORTED;
            status |= NAVHIP_ST_UNSUPPORTED;
        }
        if(R.mode >= AM_POINT_SEEK && R.mode <= AM_FORM_POINT)
            R.probes = (uint16_t)probe_tiles_bits(P, layer, me);
        R.vdes[0] = vdes.x; R.vdes[1] = vdes.z; R.arrive[0] = arrive.x; R.arrive[1] = arrive.z;
        R.status = (uint8_t)status;
    }
    pre[uid] = R;
    if(O.vdes_xz) { O.vdes_xz[2 * uid] = R.vdes[0]; O.vdes_xz[2 * uid + 1] = R.vdes[1]; }
}

// k_agent_step: one WAVE per entity -- the neighbour-dependent part: r = 30 query + separation,
// the priority ladder of the steering force, r = 10 neighbours, ClearPath, truncation.
__global__ __launch_bounds__(AG_WAVES * 64) void k_agent_step(nh_step_params P, const float *coh_xz,
                                                    const nh_pre_rec *pre, nh_step_outs O,
                                                    float scaled_max_force, double force_thresh)
{
    __shared__ wave_lds lds[AG_WAVES];
    __shared__ double exp_tab[64];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if(threadIdx.x < 64) exp_tab[threadIdx.x] = c_exp2_64[threadIdx.x];
    __syncthreads();
    const int uid = P.work_begin + blockIdx.x * AG_WAVES + wib;
    if(uid >= P.work_end) return;
    wave_lds &W = lds[wib];

    SEC_BEGIN();
    const nh_pre_rec R = pre[uid];
    v2 out_vel = mkv(0.0f, 0.0f), vpref = mkv(0.0f, 0.0f);
    if(R.mode != AM_IDLE && R.mode != AM_UNSUPPORTED) {
        const uint32_t my_flags = P.flags[uid];
        const v2 me = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
        const v2 vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
        const float my_radius = P.radius[uid];
        const int flock = P.flock[uid];
        const int hz = P.hz;
        int n30raw = -1;                 // size of the unfiltered r=30 list (-1: no such query)

        if(R.mode != AM_ZERO_VPREF) {
            const tile_probes probes = unpack_probes(R.probes);
            const v2 arrive = mkv(R.arrive[0], R.arrive[1]);
            // separation (movement.c:1690): r = 30 query, cap 128
            SEC_MARK(0);
#if NH_DUP == 1
            sp_query_wave(P.grid, me.x, me.z, 30.0f, 128, W.ids30, lane, W.d2_30);
            wave_sync(); DUP_BARRIER();
#endif
            int n30 = sp_query_wave(P.grid, me.x, me.z, 30.0f, 128, W.ids30, lane, W.d2_30);
            wave_sync();
            SEC_MARK(1);
            n30raw = n30;
#if NH_DUP == 2
            derive_r10(P.grid, me, W.ids30, W.d2_30, n30raw, W.ids10d, lane);
            filter_garrisoned_wave(P.grid.rec, W.ids30, n30, lane);
            wave_sync(); DUP_BARRIER();
#endif
            const int n10d = derive_r10(P.grid, me, W.ids30, W.d2_30, n30raw, W.ids10d, lane);
            n30 = filter_garrisoned_wave(P.grid.rec, W.ids30, n30, lane);
            if(n10d < 0) n30raw = -1; else n30raw = n10d;
            SEC_MARK(2);
#if NH_DUP == 3
            {
                const v2 sdup = separation_wave(P, uid, me, my_radius, my_flags, W.ids30, n30,
                                                W.u.sep, scaled_max_force, lane, exp_tab);
                if(sdup.x == 12345.678f) W.d2_30[0] = 1;        // keep it alive
                wave_sync(); DUP_BARRIER();
            }
#endif
            const v2 separation = separation_wave(P, uid, me, my_radius, my_flags, W.ids30, n30,
                                                  W.u.sep, scaled_max_force, lane, exp_tab);
            SEC_MARK(3);
            v2 steer;
            if(R.mode == AM_ENEMY_SEEK) {
                // enemy_seek_vpref :1946 (no priorities, no nullify)
                v2 a = vscale(arrive, 0.5f), s = vscale(separation, 0.6f);
                v2 ret = mkv(0.0f, 0.0f);
                ret = vadd(ret, a); ret = vadd(ret, s);
                steer = vtrunc(ret, scaled_max_force);
            }else{
                // point_seek_vpref :1870 / cell_arrival_seek_vpref :1908 / formation_seek_vpref :1985
                const bool to_cell = R.mode == AM_FORM_CELL, form = R.mode != AM_POINT_SEEK;
                v2 cohesion, align = mkv(0.0f, 0.0f), cell = mkv(0.0f, 0.0f);
                if(form) {
                    cohesion = mkv(P.form_cohesion_xz[2 * uid], P.form_cohesion_xz[2 * uid + 1]);
                    align = mkv(P.form_align_xz[2 * uid], P.form_align_xz[2 * uid + 1]);
                    cell = mkv(P.cell_pos_xz[2 * uid], P.cell_pos_xz[2 * uid + 1]);
                }else{
                    cohesion = (flock >= 0) ? mkv(coh_xz[2 * uid], coh_xz[2 * uid + 1]) : mkv(0.0f, 0.0f);
                }
                for(int prio = 0; prio < 3; prio++) {
                    if(prio == 0) {
                        v2 a = vscale(arrive, 0.5f), s = vscale(separation, 0.6f);
                        v2 c = vscale(cohesion, 0.15f), al = vscale(align, 0.15f);
                        v2 ret = mkv(0.0f, 0.0f);
                        ret = vadd(ret, a); ret = vadd(ret, s);
                        if(to_cell) {
                            if(vlen(vsub(cell, me)) > 30.0f) {       // CELL_ARRIVAL_RADIUS
                                ret = vadd(ret, c); ret = vadd(ret, al);
                            }
                        }else{
                            ret = vadd(ret, c);
                        }
                        steer = vtrunc(ret, scaled_max_force);
                    }else if(prio == 1) {
                        steer = separation;
                    }else{
                        steer = arrive;
                    }
                    steer = nullify_impass_pre(probes, steer);
                    if((double)vlen(steer) > force_thresh) break;
                }
            }
            v2 accel = vscale(steer, 1.0f / 1.0f);
            vpref = vtrunc(vadd(vel, accel), R.vpref_cap);
            if(R.mode == AM_FORM_CELL || R.mode == AM_FORM_POINT) {
                const v2 f_drag = mkv(P.form_drag_xz[2 * uid], P.form_drag_xz[2 * uid + 1]);
                if(vlen(f_drag) > CP_EPS)                            // :1935 / :2018
                    vpref = vtrunc(vpref, (float)(((double)P.speed[uid] * 0.75) / (double)hz));
            }
        }

        // find_neighbours :2768: r = 10 query, cap 512 -- taken from the r = 30 list when that list
        // is complete (n30raw now holds the derived count, -1 = not derivable)
        SEC_MARK(4);
        uint32_t *ids10 = W.u.ids10;
        int n10;
        if(n30raw >= 0) {
            ids10 = W.ids10d;
            n10 = n30raw;
        }else{
            n10 = sp_query_wave(P.grid, me.x, me.z, 10.0f, 512, W.u.ids10, lane);
            wave_sync();
        }
        n10 = filter_garrisoned_wave(P.grid.rec, ids10, n10, lane);
        int n_dyn, n_stat;
#if NH_DUP == 4
        filter_garrisoned_wave(P.grid.rec, ids10, n10, lane);
        classify_neighbours(P, uid, my_flags, ids10, n10, W.dyn, n_dyn, W.stat, n_stat, lane);
        wave_sync(); DUP_BARRIER();
#endif
        classify_neighbours(P, uid, my_flags, ids10, n10, W.dyn, n_dyn, W.stat, n_stat, lane);
        SEC_MARK(5);
        cpent ent; ent.pos = me; ent.vel = vel; ent.radius = my_radius;
        // the neighbour lists are dead from here on: their LDS becomes the candidate queue
        const cp_scratch cps = {W.u.rays, (float*)W.d2_30, (float*)W.ids10d, (int32_t*)W.ids30};
#if NH_DUP == 5
        {
            v2 ndup = clearpath_wave(ent, vpref, W.dyn, n_dyn, W.stat, n_stat, cps, lane);
            if(ndup.x == 12345.678f) W.dyn[0] = 1.0f;
            wave_sync(); DUP_BARRIER();
        }
#endif
        v2 nv = clearpath_wave(ent, vpref, W.dyn, n_dyn, W.stat, n_stat, cps, lane);
        SEC_MARK(6);
        out_vel = nv;                     // k_agent_post truncates it to max_speed / hz (:3464)
    }
    if(lane == 0) {
        O.vel_xz[2 * uid] = out_vel.x; O.vel_xz[2 * uid + 1] = out_vel.z;
        if(O.vpref_xz) { O.vpref_xz[2 * uid] = vpref.x; O.vpref_xz[2 * uid + 1] = vpref.z; }
    }
}

// k_agent_post: one THREAD per entity -- the position accept test of entity_compute_update
// (movement.c:2336-2358; the heading gate stays on the host) and the status byte.
__global__ __launch_bounds__(256) void k_agent_post(nh_step_params P, const nh_pre_rec *pre, nh_step_outs O)
{
    const int uid = P.work_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if(uid >= P.work_end) return;
    const v2 me = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
    const nh_pre_rec R = pre[uid];
    uint32_t status = R.status;
    v2 new_pos = me;
    // vec2_truncate(new velocity, max_speed / hz), movement.c:3464 (k_agent_step leaves it raw:
    // there one agent is one wave, here 64 agents share the instructions)
    const v2 out_vel = vtrunc(mkv(O.vel_xz[2 * uid], O.vel_xz[2 * uid + 1]), R.vel_cap);
    O.vel_xz[2 * uid] = out_vel.x; O.vel_xz[2 * uid + 1] = out_vel.z;
    if(!state_is_still(P.state[uid])) {
        const uint32_t my_flags = P.flags[uid];
        const int layer = nav_layer_for(my_flags, P.radius[uid]);
        v2 cand = vadd(me, out_vel);
        const bool on_blocked = pos_blocked(P, layer, me.x, me.z);
        bool cand_path = false, cand_blk = false;
        {
            tiledesc t;
            if(tile_for_point(P, cand.x, cand.z, t)) {               // one lookup, two planes
                const size_t idx = ((size_t)(t.chunk_r * P.map.w + t.chunk_c) << 12) + t.tile_r * 64 + t.tile_c;
                const uint16_t *bl = P.map.layers[layer].blockers;
                cand_path = P.map.layers[layer].cost[idx] != NAVHIP_COST_IMPASSABLE;
                cand_blk = bl && bl[idx] > 0;
            }
        }
        if(vlen(out_vel) > 0 && cand_path && (on_blocked || !cand_blk)) {
            new_pos = cand;
            status |= NAVHIP_ST_MOVED;
        }
    }
    if(O.new_pos_xz) { O.new_pos_xz[2 * uid] = new_pos.x; O.new_pos_xz[2 * uid + 1] = new_pos.z; }
    if(O.status) O.status[uid] = (uint8_t)status;
}

// ---------------------------------------------------------------------------------------------
// test / utility kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_spatial_query(nh_grid G, const float *query_xz, int nq,
                                                       float range, int maxout, int32_t *out_counts,
                                                       uint32_t *out_ids)
{
    const int q = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if(q >= nq) return;
    int n = sp_query_wave(G, query_xz[2 * q], query_xz[2 * q + 1], range, maxout,
                          out_ids + (size_t)q * maxout, lane);
    if(lane == 0) out_counts[q] = n;
}

__global__ __launch_bounds__(AG_WAVES * 64) void k_clearpath(int nq, const float *ent, const float *des_v,
                                                   const float *dyn, const int32_t *n_dyn,
                                                   const float *stat, const int32_t *n_stat,
                                                   float *out)
{
    __shared__ wave_lds lds[AG_WAVES];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * AG_WAVES + wib;
    if(q >= nq) return;
    wave_lds &W = lds[wib];
    for(int i = lane; i < 160; i += 64) {
        W.dyn[i] = dyn[(size_t)q * 160 + i];
        W.stat[i] = stat[(size_t)q * 160 + i];
    }
    wave_sync();
    cpent e; e.pos = mkv(ent[5 * q], ent[5 * q + 1]); e.vel = mkv(ent[5 * q + 2], ent[5 * q + 3]);
    e.radius = ent[5 * q + 4];
    const cp_scratch cps = {W.u.rays, (float*)W.d2_30, (float*)W.ids10d, (int32_t*)W.ids30};
    v2 r = clearpath_wave(e, mkv(des_v[2 * q], des_v[2 * q + 1]), W.dyn, n_dyn[q], W.stat, n_stat[q],
                          cps, lane);
    if(lane == 0) { out[2 * q] = r.x; out[2 * q + 1] = r.z; }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
void nh_launch_spatial_build(const nh_grid &G, const float *d_pos_xz, nh_spatial_scratch &S,
                             int slab_begin, int slab_end, hipStream_t s)
{
    const int n = G.n, ncells = G.grid_w * G.grid_h;
    hipMemsetAsync(S.cell_count, 0, sizeof(int32_t) * (size_t)ncells, s);
    hipMemsetAsync(S.cell_fill, 0, sizeof(int32_t) * (size_t)ncells, s);
    // a strict sub-range of the entities is stepped: hash only what its queries can reach
    const int32_t *box = nullptr;
    if(S.box && (slab_begin > 0 || slab_end < n)) {
        hipMemsetD32Async((hipDeviceptr_t)S.box, (int)0x80000000, 4, s);
        if(slab_end > slab_begin)
            hipLaunchKernelGGL(k_sp_bbox, dim3(min(64, (slab_end - slab_begin + 1023) / 1024)), dim3(1024), 0, s,
                               d_pos_xz, slab_begin, slab_end, S.box);
        box = S.box;
    }
    if(n > 0)
        hipLaunchKernelGGL(k_sp_count, dim3((n + 255) / 256), dim3(256), 0, s, G, d_pos_xz, n,
                           S.ent_ix, S.ent_iy, S.ent_cell, S.cell_count, box, S.src, S.src.flags ? S.rec : nullptr);
    const int nblocks = (ncells + 1023) / 1024;
    hipLaunchKernelGGL(k_sp_scan_local, dim3(nblocks), dim3(1024), 0, s, S.cell_count, S.cell_start,
                       S.block_sum, ncells);
    hipLaunchKernelGGL(k_sp_scan_add, dim3(nblocks), dim3(1024), 0, s, S.cell_start, S.block_sum, ncells,
                       nblocks);
    if(n > 0)
        hipLaunchKernelGGL(k_sp_scatter, dim3((n + 255) / 256), dim3(256), 0, s, S.ent_cell, n,
                           S.cell_start, S.cell_fill, S.sorted_id);
    hipLaunchKernelGGL(k_sp_order, dim3((ncells + 255) / 256), dim3(256), 0, s, S.cell_start, ncells,
                       S.sorted_id, S.ent_ix, S.ent_iy, S.sx, S.sy);
}

// scratch of the cohesion launch: wave prefix | bin counts | bin fills | bin starts | scan block sums
// | bin of each CSR entry | perm x2 | flock offsets the two perms were built for x2 | perm-valid flag.
// (Tried: scan + plan + re-zeroing fused into ONE single-workgroup kernel to shorten the chain of
// dependent launches -- 0.465 vs 0.450 ms/tick in one session: the serial chunks of a single
// workgroup take longer than three small parallel kernels.)
struct coh_scratch {
    int32_t *wave_off, *bin_count, *bin_fill, *bin_start, *block_sum, *bin_of, *perm[2], *saved[2], *valid;
    int nb, nblocks;
};
static coh_scratch coh_layout(int32_t *scratch, int n_flocks, int n_members)
{
    coh_scratch C;
    C.nb = n_flocks * COH_BINS; C.nblocks = (C.nb + 1023) / 1024;
    C.wave_off = scratch;
    C.bin_count = C.wave_off + n_flocks + 1;
    C.bin_fill = C.bin_count + C.nb;
    C.bin_start = C.bin_fill + C.nb;                      // [nb + 1]
    C.block_sum = C.bin_start + C.nb + 1;
    C.bin_of = C.block_sum + C.nblocks;
    C.perm[0] = C.bin_of + n_members;
    C.perm[1] = C.perm[0] + n_members;
    C.saved[0] = C.perm[1] + n_members;
    C.saved[1] = C.saved[0] + n_flocks + 1;
    C.valid = C.saved[1] + n_flocks + 1;
    return C;
}
size_t nh_cohesion_scratch_bytes(int n_flocks, int n_members)
{
    const size_t nb = (size_t)n_flocks * COH_BINS;
    return sizeof(int32_t) * (3 * ((size_t)n_flocks + 1) + 3 * nb + 1 + (nb + 1023) / 1024
                              + 3 * (size_t)n_members + 1);
}

// after (re)allocation: no grouping has been built for any flock layout yet
void nh_cohesion_scratch_reset(int32_t *scratch, int n_flocks, int n_members, hipStream_t s)
{
    const coh_scratch C = coh_layout(scratch, n_flocks, n_members);
    hipMemsetAsync(C.saved[0], 0xff, sizeof(int32_t) * 2 * ((size_t)n_flocks + 1), s);
}

// the counting sort that regroups the lanes of every flock (k_coh_bin .. k_coh_scatter) into perm[which]
static void coh_regroup(const nh_step_params &P, const coh_scratch &C, int which, bool plan_from_bins,
                        hipStream_t s)
{
    hipMemsetAsync(C.bin_count, 0, sizeof(int32_t) * 2 * (size_t)C.nb, s);
    const int gm = (P.n_members + 255) / 256;
    hipLaunchKernelGGL(k_coh_bin, dim3(gm), dim3(256), 0, s, P, C.bin_of, C.bin_count, C.saved[which]);
    hipLaunchKernelGGL(k_sp_scan_local, dim3(C.nblocks), dim3(1024), 0, s, C.bin_count, C.bin_start,
                       C.block_sum, C.nb);
    hipLaunchKernelGGL(k_sp_scan_add, dim3(C.nblocks), dim3(1024), 0, s, C.bin_start, C.block_sum, C.nb,
                       C.nblocks);
    if(plan_from_bins)
        hipLaunchKernelGGL(k_coh_plan, dim3(1), dim3(256), 0, s, (const int32_t*)C.bin_start,
                           P.flock_offsets, (const int32_t*)nullptr, P.n_flocks, C.wave_off, (int32_t*)nullptr);
    hipLaunchKernelGGL(k_coh_scatter, dim3(gm), dim3(256), 0, s, P, C.bin_of, C.bin_start, C.bin_fill,
                       C.perm[which]);
}

// The cohesion term of one tick.  *parity (in/out, kept by the context) = which of the two perm
// buffers the NEXT regrouping writes.
//  * A rank that steps only a slab regroups first (members outside the slab must not occupy lanes,
//    and only a fresh count says how many waves that takes), then runs k_cohesion.
//  * When every entity is stepped, k_cohesion starts at once on the grouping the PREVIOUS tick left
//    behind (any permutation of a flock's entries is valid; the grouping only has to be spatially
//    coherent, and agents move ~1 wu per tick; k_coh_plan checks that it was built for these flock
//    offsets, else the identity is used), and the regrouping for the next tick follows it --
//    nh_launch_cohesion_regroup, which the caller launches AFTER recording its "cohesion done"
//    event: five dependent small launches leave the tick's critical path.
bool nh_launch_cohesion(const nh_step_params &P, int32_t *scratch, float *d_coh, int *parity, hipStream_t s)
{
    if(!(P.n_ents > 0 && P.n_flocks > 0 && P.n_members > 0)) return false;
    const coh_scratch C = coh_layout(scratch, P.n_flocks, P.n_members);
    // upper bound of the number of 16-member (COH_APW) waves; surplus waves exit at once
    const int nwaves = (P.n_members + 15) / 16 + P.n_flocks;
    const bool whole = P.work_begin == 0 && P.work_end == P.n_ents;
    if(!whole) {
        coh_regroup(P, C, *parity, true, s);
        hipLaunchKernelGGL(k_cohesion, dim3(nwaves), dim3(64), 0, s, P, (const int32_t*)C.wave_off,
                           (const int32_t*)C.perm[*parity], (const int32_t*)nullptr, d_coh);
        *parity ^= 1;
        return false;
    }
    const int prev = *parity ^ 1;
    hipLaunchKernelGGL(k_coh_plan, dim3(1), dim3(256), 0, s, (const int32_t*)nullptr, P.flock_offsets,
                       (const int32_t*)C.saved[prev], P.n_flocks, C.wave_off, C.valid);
    hipLaunchKernelGGL(k_cohesion, dim3(nwaves), dim3(64), 0, s, P, (const int32_t*)C.wave_off,
                       (const int32_t*)C.perm[prev], (const int32_t*)C.valid, d_coh);
    return true;                                  // caller: record the event, then ..._regroup
}

void nh_launch_cohesion_regroup(const nh_step_params &P, int32_t *scratch, int *parity, hipStream_t s)
{
    const coh_scratch C = coh_layout(scratch, P.n_flocks, P.n_members);
    coh_regroup(P, C, *parity, false, s);
    *parity ^= 1;
}

size_t nh_pre_rec_bytes() { return sizeof(nh_pre_rec); }

// the scalar pre-pass needs only the snapshot and the field pool: it may run before the spatial
// hash and the cohesion term are ready
void nh_launch_agent_pre(const nh_step_params &P, void *d_pre, const nh_step_outs &O, hipStream_t s)
{
    const int nwork = P.work_end - P.work_begin;
    if(P.n_ents > 0 && nwork > 0)
        hipLaunchKernelGGL(k_agent_pre, dim3((nwork + 255) / 256), dim3(256), 0, s, P, (nh_pre_rec*)d_pre, O);
}

void nh_launch_agent_step(const nh_step_params &P, float *d_coh, const void *d_pre, const nh_step_outs &O,
                          hipStream_t s)
{
    const int nwork = P.work_end - P.work_begin;
    if(P.n_ents > 0 && nwork > 0) {
        // SCALED_MAX_FORCE and the 1 % force threshold of movement.c:1870-1905 (same for every agent)
        const float smf = (float)((double)(0.75f / (float)P.hz) * 20.0);
        const double thresh = ((double)(0.75f / (float)P.hz) * 20.0) * 0.01;
        hipLaunchKernelGGL(k_agent_step, dim3((nwork + AG_WAVES - 1) / AG_WAVES), dim3(AG_WAVES * 64), 0, s, P,
                           d_coh, (const nh_pre_rec*)d_pre, O, smf, thresh);
        hipLaunchKernelGGL(k_agent_post, dim3((nwork + 255) / 256), dim3(256), 0, s, P,
                           (const nh_pre_rec*)d_pre, O);
    }
}

void nh_launch_spatial_query(const nh_grid &G, const float *d_query, int nq, float range, int maxout,
                             int32_t *d_counts, uint32_t *d_ids, hipStream_t s)
{
    if(nq > 0)
        hipLaunchKernelGGL(k_spatial_query, dim3((nq + 3) / 4), dim3(256), 0, s, G, d_query, nq, range,
                           maxout, d_counts, d_ids);
}

void nh_launch_clearpath(int nq, const float *ent, const float *des_v, const float *dyn,
                         const int32_t *n_dyn, const float *stat, const int32_t *n_stat, float *out,
                         hipStream_t s)
{
    if(nq > 0)
        hipLaunchKernelGGL(k_clearpath, dim3((nq + AG_WAVES - 1) / AG_WAVES), dim3(AG_WAVES * 64), 0, s, nq, ent,
                           des_v, dyn, n_dyn, stat, n_stat, out);
}
