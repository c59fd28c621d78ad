// agent_math.h -- the exact-arithmetic building blocks of the movement step, one THREAD's worth each.
//
// Arithmetic mirrors the reference's C expression by expression (same types, same order, no FMA
// contraction, IEEE divide / sqrt, double-precision exp):
//   src/pf_math.c:58-94, src/game/movement.c (vec2_truncate :643, arrive_force_point :1546,
//   nullify_impass_components :1831), src/game/clearpath.c (compute_vo_edges :130, compute_vo :153,
//   compute_hrvo :180, inside_pcr :249), src/phys/collision.c (C_InfiniteLineIntersection :820,
//   C_RayRayIntersection2D :854), src/navigation/nav.c (n_interpolated_flow_dir :3407,
//   N_PositionPathable/Blocked :4055/:4070), src/map/tile.c (:356,:391,:547).
//
// Everything here is free of cross-lane operations, so the same source also compiles with g++
// (-DNH_HOSTSIM) for the host-side unit tests of the per-thread logic under tests/hostsim -- test
// infrastructure only; libnavhip.so has no CPU path.
#pragma once
#include "agent_types.h"
#include <math.h>

#ifdef NH_HOSTSIM
#include <string.h>
#define NH_FN static inline
static inline float    nh_sqrt_native(float s) { return sqrtf(s); }
static inline float    nh_rsq_native(float s)  { return 1.0f / sqrtf(s); }
static inline float    nh_sqrt_ieee(float s)   { return sqrtf(s); }
static inline float    nh_fdiv(float a, float b) { return a / b; }
static inline int32_t  nh_f2i_rn(float v)      { return (int32_t)lrintf(v); }
static inline float    nh_fmaf(float a, float b, float c) { return fmaf(a, b, c); }
static inline double   nh_fma(double a, double b, double c) { return fma(a, b, c); }
static inline float    nh_i2f(int32_t i)  { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t  nh_f2i(float f)    { int32_t i; memcpy(&i, &f, 4); return i; }
static inline uint32_t nh_f2u(float f)    { uint32_t i; memcpy(&i, &f, 4); return i; }
static inline float    nh_u2f(uint32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline long long nh_d2ll(double d) { long long i; memcpy(&i, &d, 8); return i; }
static inline double   nh_ll2d(long long i) { double d; memcpy(&d, &i, 8); return d; }
#define NH_COLD_PATH() do { } while(0)
#else
#define NH_FN __device__ __forceinline__
NH_FN float    nh_sqrt_native(float s) { return __builtin_amdgcn_sqrtf(s); }
NH_FN float    nh_rsq_native(float s)  { return __builtin_amdgcn_rsqf(s); }
// __builtin_sqrtf is IEEE-correct under -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn is NOT
// (it lowers to the native v_sqrt_f32 approximation in this ROCm)
NH_FN float    nh_sqrt_ieee(float s)   { return __builtin_sqrtf(s); }
NH_FN float    nh_fdiv(float a, float b) { return __fdiv_rn(a, b); }
NH_FN int32_t  nh_f2i_rn(float v)      { return __float2int_rn(v); }
NH_FN float    nh_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
NH_FN double   nh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
NH_FN float    nh_i2f(int32_t i)  { return __int_as_float(i); }
NH_FN int32_t  nh_f2i(float f)    { return __float_as_int(f); }
NH_FN uint32_t nh_f2u(float f)    { return __float_as_uint(f); }
NH_FN float    nh_u2f(uint32_t i) { return __uint_as_float(i); }
NH_FN long long nh_d2ll(double d) { return __double_as_longlong(d); }
NH_FN double   nh_ll2d(long long i) { return __longlong_as_double(i); }
// a real branch: keeps a rarely taken expansion out of the common path
#define NH_COLD_PATH() asm volatile("" ::: "memory")
#endif

NH_FN uint32_t nh_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
NH_FN uint32_t nh_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------
// vec2 (pf_math.c:58-94)
// ---------------------------------------------------------------------------------------------
struct v2 { float x, z; };

NH_FN v2 mkv(float x, float z) { v2 r; r.x = x; r.z = z; return r; }
NH_FN v2 vadd(v2 a, v2 b) { return mkv(a.x + b.x, a.z + b.z); }
NH_FN v2 vsub(v2 a, v2 b) { return mkv(a.x - b.x, a.z - b.z); }
NH_FN v2 vscale(v2 a, float s) { return mkv(a.x * s, a.z * s); }
NH_FN float vdot(v2 a, v2 b) { return a.x * b.x + a.z * b.z; }

// PFM_Vec2_Len: sqrt in double of a float sum, rounded to float == correctly rounded float sqrt.
// Correctly rounded sqrt for s == 0 or s in the normal range well away from its ends: v_sqrt_f32
// (<= 1 ulp) plus the same one-ulp fix-up the compiler's IEEE expansion uses, without that
// expansion's input scaling / class handling (which only matter for denormal, infinite or NaN s).
NH_FN float sqrt_rn_normal(float s)
{
    float r = nh_sqrt_native(s);
    const float rm = nh_i2f(nh_f2i(r) - 1), rp = nh_i2f(nh_f2i(r) + 1);
    const float em = nh_fmaf(-rm, r, s), ep = nh_fmaf(-rp, r, s);
    r = (em <= 0.0f) ? rm : r;
    r = (ep > 0.0f) ? rp : r;
    return r;
}
NH_FN float vlen(v2 a)
{
    const float s = a.x * a.x + a.z * a.z;
    if(!(s >= 0x1p-90f && s <= 0x1p90f) && s != 0.0f) {
        NH_COLD_PATH();
        return nh_sqrt_ieee(s);
    }
    return sqrt_rn_normal(s);
}
NH_FN v2 vnormal(v2 a)
{
    float l = vlen(a);
    return mkv(nh_fdiv(a.x, l), nh_fdiv(a.z, l));
}
// vec2_truncate, movement.c:643
NH_FN v2 vtrunc(v2 a, float max_len)
{
    if(vlen(a) > max_len) {
        a = vnormal(a);
        a = vscale(a, max_len);
    }
    return a;
}

#define CP_EPS 0.0009765625f   /* 1.0/1024: exactly representable, so float compares == the
                                  reference's float-vs-double compares */

// ---------------------------------------------------------------------------------------------
// tile lookups (tile.c:547 M_Tile_DescForPoint2D, nav.c:4055/4070)
// ---------------------------------------------------------------------------------------------
struct tiledesc { int chunk_r, chunk_c, tile_r, tile_c; };

NH_FN bool tile_for_point(const nh_step_params &P, float x, float z, tiledesc &out)
{
    const float width = (float)(P.map.w * 256), height = (float)(P.map.h * 256);
    if(x > P.map_x || x < P.map_x - width) return false;
    if(z < P.map_z || z > P.map_z + height) return false;
    int chunk_r = (int)(fabsf(P.map_z - z) / 256.0f);      // exact: division by a power of two
    int chunk_c = (int)(fabsf(P.map_x - x) / 256.0f);
    chunk_r = chunk_r < 0 ? 0 : (chunk_r > P.map.h - 1 ? P.map.h - 1 : chunk_r);
    chunk_c = chunk_c < 0 ? 0 : (chunk_c > P.map.w - 1 ? P.map.w - 1 : chunk_c);
    float base_x = P.map_x - (float)(chunk_c * 256);
    float base_z = P.map_z + (float)(chunk_r * 256);
    int tile_r = (int)(fabsf(base_z - z) / 4.0f);
    int tile_c = (int)(fabsf(base_x - x) / 4.0f);
    out.chunk_r = chunk_r; out.chunk_c = chunk_c;
    out.tile_r = tile_r < 0 ? 0 : (tile_r > 63 ? 63 : tile_r);
    out.tile_c = tile_c < 0 ? 0 : (tile_c > 63 ? 63 : tile_c);
    return true;
}

NH_FN size_t tile_index(const nh_step_params &P, const tiledesc &t)
{
    return ((size_t)(t.chunk_r * P.map.w + t.chunk_c) << 12) + t.tile_r * 64 + t.tile_c;
}

// Entity_NavLayerWithRadius, entity.c:554
NH_FN int nav_layer_for(uint32_t flags, float radius)
{
    int base = (flags & NAVHIP_ENTITY_FLAG_WATER) ? 4 : (flags & NAVHIP_ENTITY_FLAG_AIR) ? 8 : 0;
    if(radius >= 15.0f) return base + 3;
    if(radius >= 10.0f) return base + 2;
    if(radius >= 5.0f)  return base + 1;
    return base;
}

// bit 0: pathable (cost_base != COST_IMPASSABLE, nav.c:4055), bit 1: blocked (blockers > 0, nav.c:4070) of
// one tile, from the derived row masks: 16 contiguous bytes per tile row -- the five probes of an agent fall
// into one or two 64-byte sectors instead of six (a byte plane and a 16-bit plane, three rows each)
NH_FN uint32_t tile_probe(const nh_step_params &P, int layer, const tiledesc &t)
{
    const nh_layer_view &L = P.map.layers[layer];
#if defined(NH_HOSTSIM) && !defined(NH_HOSTSIM_PROBEMASK)
    // (the host-side unit tests of the per-thread logic build their map views by hand, without derived planes; the
    // whole-library emulator build defines NH_HOSTSIM_PROBEMASK and reads what the device reads: the masks k_derive made)
    const size_t idx = tile_index(P, t);
    return (L.cost[idx] != NAVHIP_COST_IMPASSABLE ? 1u : 0u) | ((L.blockers && L.blockers[idx] > 0) ? 2u : 0u);
#else
    const uint64_t *row = L.probemask + ((((size_t)(t.chunk_r * P.map.w + t.chunk_c) << 6) + t.tile_r) << 1);
#ifdef NH_HOSTSIM
    const uint64_t mx = row[0], my = row[1];
#else
    const ulonglong2 m = *(const ulonglong2*)row;
    const uint64_t mx = m.x, my = m.y;
#endif
    return (uint32_t)((mx >> t.tile_c) & 1ull) | ((uint32_t)((my >> t.tile_c) & 1ull) << 1);
#endif
}

NH_FN bool pos_pathable(const nh_step_params &P, int layer, float x, float z)
{
    tiledesc t;
    if(!tile_for_point(P, x, z, t)) return false;     // reference asserts; off-map = not pathable
    return (tile_probe(P, layer, t) & 1u) != 0;
}

NH_FN bool pos_blocked(const nh_step_params &P, int layer, float x, float z)
{
    tiledesc t;
    if(!tile_for_point(P, x, z, t)) return false;
    return (tile_probe(P, layer, t) & 2u) != 0;
}

// The five tile probes of nullify_impass_components (own tile, +-4 wu in x and z), issued together:
// ten booleans packed as bits 0-4 pathable, 5-9 blocked (0 self, 1 x+4, 2 x-4, 3 z+4, 4 z-4).
NH_FN uint32_t probe_tiles_bits(const nh_step_params &P, int layer, v2 pos)
{
    const float px[5] = {pos.x, pos.x + 4.0f, pos.x - 4.0f, pos.x, pos.x};
    const float pz[5] = {pos.z, pos.z, pos.z, pos.z + 4.0f, pos.z - 4.0f};
    uint32_t bits = 0;
#pragma unroll
    for(int i = 0; i < 5; i++) {
        tiledesc t;
        if(!tile_for_point(P, px[i], pz[i], t)) continue;
        const uint32_t pb = tile_probe(P, layer, t);
        bits |= (pb & 1u) << i;
        bits |= ((pb >> 1) & 1u) << (5 + i);
    }
    return bits;
}

// nullify_impass_components, movement.c:1831, on the packed probes
NH_FN v2 nullify_impass_bits(uint32_t bits, v2 f)
{
    const bool on_blocked = (bits >> 5) & 1;
#define NH_PATH(i) ((bits >> (i)) & 1u)
#define NH_BLK(i)  ((bits >> (5 + (i))) & 1u)
    if(f.x > 0 && (!NH_PATH(1) || (!on_blocked && NH_BLK(1)))) f.x = 0.0f;
    if(f.x < 0 && (!NH_PATH(2) || (!on_blocked && NH_BLK(2)))) f.x = 0.0f;
    if(f.z > 0 && (!NH_PATH(3) || (!on_blocked && NH_BLK(3)))) f.z = 0.0f;
    if(f.z < 0 && (!NH_PATH(4) || (!on_blocked && NH_BLK(4)))) f.z = 0.0f;
#undef NH_PATH
#undef NH_BLK
    return f;
}

// ---------------------------------------------------------------------------------------------
// flow-field sampling (nav.c:3407 n_interpolated_flow_dir, :3468 N_DesiredPointSeekVelocity)
// ---------------------------------------------------------------------------------------------
NH_FN v2 flow_dir_vec(int dir)           // N_FlowDir, field.c:2428
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

// The cache-hit path of N_DesiredPointSeekVelocity: the four taps are fetched one after the other;
// the latency is hidden by the other agents of the wave.  A missing field / FD_NONE under the agent
// is reported in `status` (the fallbacks of nav.c:3483-3554 are driven by the host: see
// navhip_collect_misses).
NH_FN v2 sample_flow(const nh_step_params &P, int flock, v2 pos, uint32_t &status)
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

    // The four taps in three dependent steps instead of up to nine: every tap's slot entry (its own chunk's
    // again when the tap stays inside the chunk: a cache hit), then every tap's direction byte, then the
    // blend in tap order with the conditions of nav.c:3435-3458 (a tap that does not count has loaded
    // the base tile for nothing).
    bool  use[4];
    int   tslot[4], toff[4];
#pragma unroll
    for(int i = 0; i < 4; i++) {
        // M_Tile_RelativeDesc, tile.c:391
        const int abs_r = t.chunk_r * 64 + t.tile_r + sdr[i];
        const int abs_c = t.chunk_c * 64 + t.tile_c + sdc[i];
        use[i] = sw[i] > 0.0f && !(abs_r < 0 || abs_r >= P.map.h * 64 || abs_c < 0 || abs_c >= P.map.w * 64);
        const int cr = use[i] ? abs_r >> 6 : t.chunk_r, cc = use[i] ? abs_c >> 6 : t.chunk_c;
        toff[i] = use[i] ? (abs_r & 63) * 64 + (abs_c & 63) : t.tile_r * 64 + t.tile_c;
        tslot[i] = slots[cr * P.map.w + cc];
    }
    int tdir[4];
#pragma unroll
    for(int i = 0; i < 4; i++) {
        use[i] = use[i] && tslot[i] >= 0;
        tdir[i] = P.field_pool[((size_t)(use[i] ? tslot[i] : slot) << 12) + toff[i]] & 0xf;
    }
    v2 acc = mkv(0.0f, 0.0f);
    float wsum = 0.0f;
#pragma unroll
    for(int i = 0; i < 4; i++) {
        if(!use[i] || tdir[i] == NAVHIP_FD_NONE) continue;
        v2 scaled = vscale(flow_dir_vec(tdir[i]), sw[i]);
        acc = vadd(acc, scaled);
        wsum += sw[i];
    }
    if(wsum < 1e-6f || vlen(acc) < 1e-6f)
        return flow_dir_vec(base_dir);
    return vnormal(acc);
}

// N_DesiredEnemySeekVelocity (nav.c:3603) / N_DesiredSurroundVelocity (nav.c:3687), cache-hit path:
// N_FlowDir of the tile under the entity in the chunk field of its mapping row -- no blend.  A missing
// field and FD_NONE under the entity (the repair builds of :3652-3683, :3733-3760) are reported.
NH_FN v2 sample_region(const nh_step_params &P, int row, v2 pos, uint32_t &status)
{
    tiledesc t;
    if(!P.region_field_slot || !P.field_pool || !tile_for_point(P, pos.x, pos.z, t)) {
        status |= NAVHIP_ST_FIELD_MISS;
        return mkv(0.0f, 0.0f);
    }
    const int slot = P.region_field_slot[(size_t)row * (P.map.w * P.map.h) + t.chunk_r * P.map.w + t.chunk_c];
    if(slot < 0) {
        status |= NAVHIP_ST_FIELD_MISS;
        return mkv(0.0f, 0.0f);
    }
    const int dir = P.field_pool[((size_t)slot << 12) + t.tile_r * 64 + t.tile_c] & 0xf;
    if(dir == NAVHIP_FD_NONE) status |= NAVHIP_ST_FIELD_NONE;
    return flow_dir_vec(dir);
}

// move_work_in.ent_des_v: host supplied, or sampled from the device field pool (vdes_xz == NULL or
// a NaN entry)
NH_FN v2 load_vdes(const nh_step_params &P, int uid, int flock, v2 me, uint32_t &status)
{
    if(P.vdes_xz) {
        v2 v = mkv(P.vdes_xz[2 * uid], P.vdes_xz[2 * uid + 1]);
        if(v.x == v.x) return v;
    }
    if(P.region_row) {
        const int row = P.region_row[uid];
        if(row >= 0) return sample_region(P, row, me, status);
    }
    return sample_flow(P, flock, me, status);
}

// ---------------------------------------------------------------------------------------------
// spatial hash geometry (bitmap_grid.h)
// ---------------------------------------------------------------------------------------------
NH_FN int32_t bg_scale(float v) { return nh_f2i_rn(v * 256.0f); }   // BG_SCALE_F

NH_FN int sp_cell_of(const nh_grid &G, int32_t ix, int32_t iy)
{
    int cx = (ix - G.origin_x) >> 12;             // BG_CELL_LOG2_INT = 8 + 4
    int cy = (iy - G.origin_y) >> 12;
    cx = cx < 0 ? 0 : (cx > G.grid_w - 1 ? G.grid_w - 1 : cx);
    cy = cy < 0 ? 0 : (cy > G.grid_h - 1 ? G.grid_h - 1 : cy);
    return cy * G.grid_w + cx;
}

// Extent of a query in fine cells + whether it takes the reference's wide-query path
// (bitmap_grid.h:1389-1397); returns false when the query box misses the grid.
struct sp_extent { int cx_lo, cx_hi, cy_lo, cy_hi; bool wide; };

NH_FN bool sp_query_extent(const nh_grid &G, int32_t icx, int32_t icy, int32_t ir, sp_extent &E)
{
    const int32_t imnx = icx - ir, imxx = icx + ir, imny = icy - ir, imxy = icy + ir;
    // _bg_cell_extent, bitmap_grid.h:1236
    if(imxx < G.origin_x || imxy < G.origin_y) return false;
    const int32_t span_x = (int32_t)((uint32_t)G.grid_w << 12), span_y = (int32_t)((uint32_t)G.grid_h << 12);
    if(imnx >= G.origin_x + span_x || imny >= G.origin_y + span_y) return false;
    int v;
    v = (imnx - G.origin_x) >> 12; E.cx_lo = v > 0 ? v : 0;
    v = (imny - G.origin_y) >> 12; E.cy_lo = v > 0 ? v : 0;
    v = (imxx - G.origin_x) >> 12; E.cx_hi = v < G.grid_w - 1 ? v : G.grid_w - 1;
    v = (imxy - G.origin_y) >> 12; E.cy_hi = v < G.grid_h - 1 ? v : G.grid_h - 1;
    E.wide = (int64_t)(E.cx_hi - E.cx_lo + 1) * (E.cy_hi - E.cy_lo + 1) * 4 >= (int64_t)G.grid_w * G.grid_h * 3;
    return true;
}

// ---------------------------------------------------------------------------------------------
// (float)exp((double)a)
// ---------------------------------------------------------------------------------------------
// 2^(j/64), j = 0..63, correctly rounded doubles (device: __constant__ c_exp2_64, staged in LDS)
#define NH_EXP2_64_TABLE \
    0x1p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0, \
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92dep+0, \
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0, \
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0, \
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0, \
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cdp+0, \
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0, \
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0, \
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0, \
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0, \
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e5p+0, \
    0x1.9c49182a3f09p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0, \
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0, \
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0, \
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0, \
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e454p+0, 0x1.fa7c1819e90d8p+0

// (float)exp((double)a) for a in (-inf, ~88]: the reference evaluates libm's double exp on a float
// argument and rounds to float (movement.c:1671,1731).  Table-driven double evaluation,
// exp(a) = 2^(k/64) * exp(r), |r| <= ln2/128, degree-5 polynomial: < 2 ulp in double, so the float
// rounding agrees with a correctly rounded exp except with probability ~1e-8 per call.  tab =
// 64-entry table (LDS on the device).  k = rint(x * 64/ln2) falls out of the low mantissa bits of
// x * 64/ln2 + 1.5 * 2^52 (one FMA), and no final select is needed -- the clamped argument -104
// gives 6.8e-46, which the f64 -> f32 conversion rounds to +0 like every value below half the
// smallest denormal (the true cut-off is a = -103.972).  The final scaling by 2^(k>>6) is an
// integer add on the exponent field (the result stays a normal double for every argument in
// [-104, 89]).  Checked against glibc's exp on 3e8 random arguments in [-110, 6], 3e8 in [-21, 89]
// and on every float in [-104.5, -102]: no mismatch.
NH_FN float exp_f32_magic(float a, const double *tab)
{
    const double x = (double)fmaxf(a, -104.0f);
    const double z = nh_fma(x, 0x1.71547652b82fep+6, 0x1.8p52);   // 64/ln2
    const double kd = z - 0x1.8p52;
    const int k = (int)nh_d2ll(z);
    double r = nh_fma(-kd, 0x1.62e42fefa0000p-7, x);              // ln2/64, high part
    r = nh_fma(-kd, 0x1.cf79abc9e3b3ap-46, r);                    //         low part
    double p = nh_fma(r, 1.0 / 120, 1.0 / 24);
    p = nh_fma(p, r, 1.0 / 6);
    p = nh_fma(p, r, 0.5);
    p = nh_fma(p, r, 1.0);
    p = nh_fma(p, r, 1.0);
    const double v = tab[k & 63] * p;
    const long long bits = nh_d2ll(v) + ((long long)(k >> 6) << 52);
    return (float)nh_ll2d(bits);
}

// ---------------------------------------------------------------------------------------------
// cohesion_force (movement.c:1653): the weight's argument
// ---------------------------------------------------------------------------------------------
// float t = (len - 50.0f*0.75) / 50.0f of movement.c:1668 (the reference evaluates it in double and
// rounds to float).  For len >= 16 the f32 subtraction is exact and the division by 50 as
// reciprocal multiply + one FMA correction (Markstein) reproduces the double-then-float result for
// EVERY float in [16, 8192) (swept on the device: tests/test_mathsweep_gpu.py, MS_COH_T_F32); beyond
// that the weight is 0 anyway.
NH_FN float cohesion_t_f32(float len)
{
    const float r50f = 1.0f / 50.0f;
    const float x = len - 37.5f;
    const float q0 = x * r50f;
    return nh_fmaf(nh_fmaf(-q0, 50.0f, x), r50f, q0);
}

// the same through double, for len < 16 where the f32 subtraction may round (every float in [0, 16):
// MS_COH_T_F64)
NH_FN float cohesion_t_f64(float len)
{
    const double r50 = 1.0 / 50.0;
    const double x = (double)len - (double)50.0f * 0.75;
    const double q0 = x * r50;
    return (float)nh_fma(nh_fma(-q0, 50.0, x), r50, q0);
}

// ---------------------------------------------------------------------------------------------
// movement states / steering terms
// ---------------------------------------------------------------------------------------------
NH_FN bool state_is_still(int s)
{
    return s == NAVHIP_STATE_ARRIVED || s == NAVHIP_STATE_WAITING;     // ent_still, movement.c:652
}
NH_FN bool state_uses_point_seek(int s)
{
    return s == NAVHIP_STATE_MOVING || s == NAVHIP_STATE_SURROUND_ENTITY
        || s == NAVHIP_STATE_ENTER_ENTITY_RANGE;
}

// arrive_force_point, movement.c:1546
NH_FN v2 arrive_force(v2 me, v2 vel, v2 target, v2 vdes, bool los, float max_speed, int hz,
                      float scaled_max_force)
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

// one term of separation_force (movement.c:1713-1734): diff * exp(min(-20 t, 40)); false = skipped
NH_FN bool separation_term(v2 me, float my_radius, v2 cp, float cp_radius, const double *exp_tab, v2 &term)
{
    float radius = my_radius + cp_radius + 0.0f;                     // SEPARATION_BUFFER_DIST
    v2 diff = vsub(cp, me);
    float len = vlen(diff);
    if(len < CP_EPS) return false;
    float t = nh_fdiv(len - radius * 0.85f, len);
    float scale = exp_f32_magic(fminf(-20.0f * t, 40.0f), exp_tab);
    term = vscale(diff, scale);
    return true;
}

// ---------------------------------------------------------------------------------------------
// ClearPath primitives (clearpath.c, collision.c)
// ---------------------------------------------------------------------------------------------
struct cpent { v2 pos, vel; float radius; };

// slope of a line as C_InfiniteLineIntersection takes it (collision.c:823-831): NaN = vertical
NH_FN float line_slope(v2 dir)
{
    return fabsf(dir.x) < CP_EPS ? nh_u2f(0x7fc00000u) : nh_fdiv(dir.z, dir.x);
}

// C_InfiniteLineIntersection, collision.c:820 (including the l2.point term of the vertical-l2
// branch, :840), with the two slopes s1/s2 = line_slope(dir) supplied by the caller (they only
// depend on the line, and every line meets many others)
NH_FN bool line_isect(v2 p1, float s1, v2 p2, float s2, v2 &out)
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
        out.x = nh_fdiv((s1 * p1.x - s2 * p2.x + p2.z - p1.z), (s1 - s2));
        out.z = s2 * (out.x - p2.x) + p2.z;
    }
    return true;
}

// `a / b < 0.0f` of C_RayRayIntersection2D (collision.c:862-871) without the division when the sign
// rule is safe: for finite a, b with a == 0 or |a| >= 2^-100 and |b| <= 2^20 the quotient cannot
// underflow to -0, so it is negative exactly when a != 0 and the signs differ (b = +-0 included:
// a/+-0 = +-inf).  ok = false -> the caller divides.
NH_FN bool quot_neg_fast(float a, float b, bool &ok)
{
    const float aa = fabsf(a);
    ok = ok && (aa >= 0x1p-100f || a == 0.0f) && aa < INFINITY && fabsf(b) <= 0x1p20f;
    return a != 0.0f && ((nh_f2i(a) ^ nh_f2i(b)) < 0);
}

// C_RayRayIntersection2D, collision.c:854
NH_FN bool ray_isect(v2 p1, v2 d1, float s1, v2 p2, v2 d2, float s2, v2 &out)
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
        neg = nh_fdiv(a1, d1.x) < 0.0f || nh_fdiv(a2, d1.z) < 0.0f
           || nh_fdiv(a3, d2.x) < 0.0f || nh_fdiv(a4, d2.z) < 0.0f;
    }
    if(neg) return false;
    out = p;
    return true;
}

// C_RayRayIntersection2D + the distance of the point to des_v, without a branch in the common case: the three
// shapes of C_InfiniteLineIntersection are all evaluated and selected (the division runs on whatever operands
// there are; a result that is not selected is never looked at), the sign tests are bit tests, and the two rare
// expansions -- a quotient whose sign needs the division, a squared distance outside the range of the short
// square root -- are reported in `slow` for the caller to branch on ONCE per wave (ray_isect / vlen then redo
// that lane).  Same arithmetic, same operand order as line_isect / ray_isect / vlen above.
NH_FN bool ray_isect_bf(v2 p1, v2 d1, float s1, v2 p2, v2 d2, float s2, v2 des_local, v2 ent_pos, v2 &out, float &len,
                        bool &slow)
{
    const bool n1 = s1 != s1, n2 = s2 != s2;
    const float ds = s1 - s2;
    bool ok = !(n1 & n2) & !(fabsf(ds) < CP_EPS);
    const float xg = nh_fdiv((s1 * p1.x - s2 * p2.x + p2.z - p1.z), ds);
    const float zg = s2 * (xg - p2.x) + p2.z;
    const float zv1 = (p1.x - p2.x) * s2 + p2.z;            // line 1 vertical
    const float zv2 = (p2.x - p1.x) * s1 + p2.z;            // line 2 vertical
    v2 p;
    p.x = n1 ? p1.x : (n2 ? p2.x : xg);
    p.z = n1 ? zv1 : (n2 ? zv2 : zg);
    const float a1 = p.x - p1.x, a2 = p.z - p1.z, a3 = p.x - p2.x, a4 = p.z - p2.z;
    // quot_neg_fast's conditions for all four quotients at once, on the bit patterns of |a| (ordered like the
    // values; a NaN is the largest): every |a| is 0 or in [2^-100, inf) -- (bits - 1) wraps a zero out of the
    // minimum --, every |b| <= 2^20 (a NaN fails that compare too)
    const uint32_t u1 = nh_f2u(a1) & 0x7fffffffu, u2 = nh_f2u(a2) & 0x7fffffffu, u3 = nh_f2u(a3) & 0x7fffffffu,
                   u4 = nh_f2u(a4) & 0x7fffffffu;
    const uint32_t lo = nh_umin(nh_umin(u1 - 1u, u2 - 1u), nh_umin(u3 - 1u, u4 - 1u));
    const uint32_t hi = nh_umax(nh_umax(u1, u2), nh_umax(u3, u4));
    const uint32_t ub = nh_umax(nh_umax(nh_f2u(d1.x) & 0x7fffffffu, nh_f2u(d1.z) & 0x7fffffffu),
                                nh_umax(nh_f2u(d2.x) & 0x7fffffffu, nh_f2u(d2.z) & 0x7fffffffu));
    const bool fast = lo >= 0x0d800000u - 1u && hi < 0x7f800000u && ub <= 0x49800000u;      // 2^-100, inf, 2^20
    const bool neg = (a1 != 0.0f && ((nh_f2i(a1) ^ nh_f2i(d1.x)) < 0)) | (a2 != 0.0f && ((nh_f2i(a2) ^ nh_f2i(d1.z)) < 0))
                   | (a3 != 0.0f && ((nh_f2i(a3) ^ nh_f2i(d2.x)) < 0)) | (a4 != 0.0f && ((nh_f2i(a4) ^ nh_f2i(d2.z)) < 0));
    const v2 rel = vsub(des_local, vsub(p, ent_pos));
    const float ss = rel.x * rel.x + rel.z * rel.z;
    const bool short_sqrt = (ss >= 0x1p-90f && ss <= 0x1p90f) || ss == 0.0f;
    len = sqrt_rn_normal(ss);
    slow = ok & (!fast | !short_sqrt);
    out = p;
    return ok & !neg;
}

// compute_vo_edges, clearpath.c:130
NH_FN void vo_edges(const cpent &ent, const cpent &nb, v2 &out_right, v2 &out_left)
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
NH_FN void make_cone(const cpent &ent, const cpent &nb, bool hrvo, v2 &apex, v2 &left, v2 &right,
                     float &sl, float &sr)
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

// inside_pcr, clearpath.c:249.  A cone is two float4:
//   A = {apex.x, apex.z, slope(left), slope(right)}     B = {left.x, left.z, right.x, right.z}
// (both rays of a cone start at its apex, rays_repr :291; ray 2c is the left side, 2c+1 the right).
//
// One cone, evaluated exactly as the reference does (normalisation with IEEE sqrt/divide; the
// reference normalises test - apex once per ray, with identical operands both times):
// true when `test` is strictly inside the cone.
NH_FN bool cone_contains_exact(float4 A, float4 B, v2 test)
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
NH_FN int cone_contains_fast(float4 A, float4 B, v2 test)
{
    const float MARG = 2e-5f;
    const float px = test.x - A.x, pz = test.z - A.y;
    const float s = px * px + pz * pz;
    const float inv = nh_rsq_native(s);
    const float len = s * inv;
    // (a test point that IS the apex -- the intersection of two rays of one cone, or of two static neighbours'
    // cones, which all start at the entity's own position, often comes out bit-equal to it -- has length 0:
    // "not inside", clearpath.c:262, without the exact evaluation; in a jam that was half of all test steps)
    if(s == 0.0f) return 0;
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

// cone_contains_fast without a branch: every comparison is evaluated and the verdict picked in the order the
// function above returns in (a select chain).  For the wave-wide search, where a taken branch costs the whole
// wave its exec-mask bookkeeping and the lanes execute both sides anyway.
NH_FN int cone_test_bf(float4 A, float4 B, v2 test)
{
    const float MARG = 2e-5f;
    const float px = test.x - A.x, pz = test.z - A.y;
    const float s = px * px + pz * pz;
    const float inv = nh_rsq_native(s);
    const float len = s * inv;
    const float detl = (pz * B.x - px * B.y) * inv;
    const float detr = (pz * B.z - px * B.w) * inv;
    // (s == 0: the point is the apex -- length 0, "not inside" -- decided here; everything derived from it is NaN
    // and fails every compare below)
    const bool bad = !(s >= 0.0f) | !(s < 1e30f) | (fabsf(len - CP_EPS) <= CP_EPS * 1e-4f);
    int r = (detr > -CP_EPS) ? 0 : 1;
    r = (fabsf(detr + CP_EPS) <= MARG) ? 2 : r;
    r = (detl < CP_EPS) ? 0 : r;
    r = (fabsf(detl - CP_EPS) <= MARG) ? 2 : r;
    r = ((len < CP_EPS) | (s == 0.0f)) ? 0 : r;
    return bad ? 2 : r;
}

NH_FN bool cone_contains(float4 A, float4 B, v2 test)
{
    int v = cone_contains_fast(A, B, test);
    if(v == 2) v = cone_contains_exact(A, B, test) ? 1 : 0;
    return v == 1;
}
