// field_kernels.hip -- chunk flow-field build for gfx950 (MI355X), hand-written HIP.
//
// Reference semantics (permafrost-engine src/navigation/field.c):
//   N_FlowFieldUpdate :2030  = initial frontier (:1372,:1096,:1160) -> integration field
//   (field_build_integration :539, 4-connected, step cost = cost_base[neighbour], passable
//   neighbours only :203-251) -> flow bake (field_build_flow :734, field_flow_dir :355,
//   8-connected with the both-sides-passable diagonal guard and N,S,E,W,NW,NE,SW,SE
//   priority) -> portal fixup (field_fixup_portal_edges :830).
//
// Two kernels, both bit-exact with that path:
//
//  k_field_bfs      one WAVE per request, lane = field row.  Valid when every passable cell of
//                   the chunk has cost 1 (the only value the reference's cost writers produce
//                   besides 0xff: nav.c:339-342,416), so Dijkstra degenerates to a
//                   level-synchronous BFS.  A row of the field is one 64-bit mask held in a
//                   lane; a BFS level is ~20 VALU ops for the whole 64x64 field: W/E
//                   neighbours are 64-bit shifts, N/S neighbours are DPP wave shifts, the
//                   "frontier empty" test is one wave ballot.  Distances are kept bit-sliced
//                   (plane k of lane r = bit k of the distance of every cell of row r) and the
//                   bake is evaluated bit-sliced as well, so no per-cell loop exists until the
//                   4-bit direction codes are expanded to bytes.
//
//  k_field_generic  one 256-thread workgroup per request, u32 integration tile in LDS,
//                   chaotic min-plus relaxation to the fixpoint (== Dijkstra distances, integer
//                   exact) for arbitrary u8 costs and for faction ("attacking") passability.
//
// Integer/bit work only: no MFMA.  Compile with -ffp-contract=off (only matters for the
// int->float conversion of the optional integration output, which is exact anyway).
#include "navhip_internal.h"

#include "wave_bits.h"

__device__ __forceinline__ bool req_uses_bfs(const nh_map_view &map, const navhip_field_req &rq,
                                             int force_generic)
{
    if(force_generic) return false;
    if(rq.faction_id != NAVHIP_FACTION_ID_NONE) return false;
    if(rq.type == NAVHIP_TARGET_NEAREST_PATHABLE || (rq.flags & NAVHIP_REQ_ISLAND_NEAREST)) return false;
    const nh_layer_view &L = map.layers[rq.layer];
    return L.unit_cost[(int)rq.chunk_r * map.w + rq.chunk_c] != 0;
}

// first local-island label that is not ISLAND_NONE along a portal (a row or a column of tiles)
__device__ __forceinline__ uint16_t portal_first_iid(const uint16_t *li, int r0, int c0, int r1, int c1)
{
    for(int r = r0; r <= r1; r++)
        for(int c = c0; c <= c1; c++) {
            const uint16_t v = li[r * 64 + c];
            if(v != NAVHIP_ISLAND_NONE) return v;
        }
    return NAVHIP_ISLAND_NONE;
}

// Request flags that are resolved on the device: NAVHIP_REQ_IF_CHANGED (skip unless the chunk, or
// the next chunk of a portal target, changed) and NAVHIP_REQ_LIVE_IIDS (island ids re-read from
// the current labels).  Returns false when the request is to be skipped.
__device__ __forceinline__ bool req_prepare(const nh_map_view &map, navhip_field_req &rq)
{
    const nh_layer_view &L = map.layers[rq.layer];
    const bool portal = rq.type == NAVHIP_TARGET_PORTAL;
    if((rq.flags & NAVHIP_REQ_IF_CHANGED) && L.changed) {
        bool ch = L.changed[(int)rq.chunk_r * map.w + rq.chunk_c] != 0;
        if(portal) ch |= L.changed[(int)rq.next_chunk_r * map.w + rq.next_chunk_c] != 0;
        if(!ch) return false;
    }
    if((rq.flags & NAVHIP_REQ_LIVE_IIDS) && portal && L.local_islands) {
        // the label of the first tile of each portal that HAS one (a blocker on a portal tile leaves
        // ISLAND_NONE there; the planner's ids come from reachable tiles, nav.c:1893-1907).  A portal
        // that is blocked from end to end leads nowhere: the request is skipped.
        rq.port_iid = portal_first_iid(L.local_islands + ((size_t)((int)rq.chunk_r * map.w + rq.chunk_c) << 12),
                                       rq.port_r0, rq.port_c0, rq.port_r1, rq.port_c1);
        rq.next_iid = portal_first_iid(L.local_islands + ((size_t)((int)rq.next_chunk_r * map.w + rq.next_chunk_c) << 12),
                                       rq.next_r0, rq.next_c0, rq.next_r1, rq.next_c1);
        if(rq.port_iid == NAVHIP_ISLAND_NONE || rq.next_iid == NAVHIP_ISLAND_NONE) return false;
    }
    return true;
}

// Is the cell (r2,c2) of the chunk (cr2,cc2) a tile of the `next` portal that lies on local
// island next_iid?  One of the four candidates that can be at Manhattan distance 1 from a
// portal tile: the O(1) form of the scan in field_tile_adjacent_to_next_iid (field.c:1131).
__device__ __forceinline__ bool next_tile_matches(const nh_map_view &map, const navhip_field_req &rq,
                                                  int gr, int gc)
{
    if(gr < 0 || gc < 0) return false;
    int cr2 = gr >> 6, cc2 = gc >> 6, r2 = gr & 63, c2 = gc & 63;
    if(cr2 != rq.next_chunk_r || cc2 != rq.next_chunk_c) return false;
    if(cr2 >= map.h || cc2 >= map.w) return false;
    if(r2 < rq.next_r0 || r2 > rq.next_r1 || c2 < rq.next_c0 || c2 > rq.next_c1) return false;
    const uint16_t *li = map.layers[rq.layer].local_islands;
    return li[((size_t)(cr2 * map.w + cc2) << 12) + r2 * 64 + c2] == rq.next_iid;
}

// field_portal_initial_frontier (field.c:1160) for one portal tile, given its passability.
__device__ __forceinline__ bool portal_seed(const nh_map_view &map, const navhip_field_req &rq,
                                            int r, int c, bool passable)
{
    if(!passable) return false;
    if(r < rq.port_r0 || r > rq.port_r1 || c < rq.port_c0 || c > rq.port_c1) return false;
    const uint16_t *li = map.layers[rq.layer].local_islands;
    size_t cbase = (size_t)((int)rq.chunk_r * map.w + rq.chunk_c) << 12;
    if(rq.port_iid != NAVHIP_ISLAND_NONE && li[cbase + r * 64 + c] != rq.port_iid) return false;
    int gr = rq.chunk_r * 64 + r, gc = rq.chunk_c * 64 + c;
    return next_tile_matches(map, rq, gr - 1, gc) || next_tile_matches(map, rq, gr + 1, gc)
        || next_tile_matches(map, rq, gr, gc - 1) || next_tile_matches(map, rq, gr, gc + 1);
}

// direction written into cost-0 cells of a TARGET_PORTAL field (field.c:838-857)
__device__ __forceinline__ uint32_t portal_fix_dir(const navhip_field_req &rq)
{
    if(rq.next_chunk_r < rq.chunk_r) return NAVHIP_FD_N;
    if(rq.next_chunk_r > rq.chunk_r) return NAVHIP_FD_S;
    if(rq.next_chunk_c < rq.chunk_c) return NAVHIP_FD_W;
    return NAVHIP_FD_E;
}

// ---------------------------------------------------------------------------------------------
// derived per-chunk state: passability rows + "all passable costs are 1" flag
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_derive(const uint8_t *cost, const uint16_t *blockers,
                                                uint64_t *passmask, uint64_t *probemask, uint8_t *unit_cost,
                                                const uint32_t *chunk_list, int n)
{
    int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if(wave >= n) return;
    uint32_t chunk = chunk_list ? chunk_list[wave] : (uint32_t)wave;
    const uint8_t  *cb = cost + ((size_t)chunk << 12);
    const uint16_t *bl = blockers ? blockers + ((size_t)chunk << 12) : nullptr;
    uint64_t mine = 0, path = 0, blocked = 0;
    bool nonunit = false;
    for(int r = 0; r < 64; r++) {
        uint32_t cst = cb[r * 64 + lane];
        uint32_t blk = bl ? bl[r * 64 + lane] : 0;
        uint64_t m = __ballot(cst != NAVHIP_COST_IMPASSABLE && blk == 0);   // field.c:117-124
        uint64_t mp = __ballot(cst != NAVHIP_COST_IMPASSABLE), mb = __ballot(blk > 0);
        nonunit |= (cst != NAVHIP_COST_IMPASSABLE && cst != 1);
        if(lane == r) { mine = m; path = mp; blocked = mb; }
    }
    passmask[(size_t)chunk * 64 + lane] = mine;
    probemask[((size_t)chunk * 64 + lane) * 2]     = path;
    probemask[((size_t)chunk * 64 + lane) * 2 + 1] = blocked;
    bool any_nonunit = __any(nonunit);
    if(lane == 0) unit_cost[chunk] = any_nonunit ? 0 : 1;
}

void nh_launch_derive(navhip_ctx *ctx, int layer, const uint32_t *d_chunk_list, int n, hipStream_t s)
{
    navhip_layer &L = ctx->layers[layer];
    int blocks = (n + 3) / 4;
    hipLaunchKernelGGL(k_derive, dim3(blocks), dim3(256), 0, s, L.cost, L.blockers, L.passmask, L.probemask,
                       L.unit_cost, d_chunk_list, n);
}

// ---------------------------------------------------------------------------------------------
// k_field_bfs : one wave per request, lane = row
// ---------------------------------------------------------------------------------------------
#define NH_MAXP 12   /* distance bit-planes: unit-cost distances are < 4096 */
#define BFS_WAVES 4    /* waves (= requests) per workgroup */

template <bool WANT_INTEG>
__global__ __launch_bounds__(BFS_WAVES * 64) void k_field_bfs(nh_map_view map, const navhip_field_req *reqs,
                                                   int n, uint8_t *dirs, float *integ,
                                                   int force_generic, int32_t *gen_list, int gen_slot,
                                                   const int32_t *out_slot)
{
    // 4 KB of LDS per wave: staging buffer to turn "lane owns a 64-byte row" into fully
    // coalesced 16 B/lane global accesses (both for the INOUT read and for the final write).
    __shared__ __attribute__((aligned(16))) uint8_t stage[BFS_WAVES][NH_CELLS];

    const int wib  = threadIdx.x >> 6;
    const int wave = blockIdx.x * BFS_WAVES + wib;
    const int lane = threadIdx.x & 63;
    if(wave >= n) return;

    navhip_field_req rq = reqs[wave];
    if(!req_uses_bfs(map, rq, force_generic)) {
        // not a unit-cost BFS: hand the request to k_field_generic
        if(lane == 0) gen_list[2 + atomicAdd(&gen_list[gen_slot], 1)] = wave;
        return;
    }
    if(!req_prepare(map, rq)) return;

    const nh_layer_view &L = map.layers[rq.layer];
    const int chunk = (int)rq.chunk_r * map.w + rq.chunk_c;
    const u64x pass = mk(L.passmask[(size_t)chunk * 64 + lane]);

    // ---- initial frontier (field.c:1372) ----------------------------------------------------
    u64x seeds = u64x{0, 0};
    if(rq.type == NAVHIP_TARGET_TILE) {
        if(lane == rq.tile_r) seeds = mk(1ull << rq.tile_c) & pass;          // field.c:1096-1128
    }else{
        if(rq.port_c0 == rq.port_c1) {
            // portal on a vertical chunk edge: one tile per row, lane = row
            bool s = portal_seed(map, rq, lane, rq.port_c0, (to64(pass) >> rq.port_c0) & 1);
            if(s) seeds = mk(1ull << rq.port_c0);
        }else{
            // portal on a horizontal edge (or a general rectangle): lanes = columns, one ballot
            // per portal row
            for(int r = rq.port_r0; r <= rq.port_r1; r++) {
                uint64_t prow = __shfl(to64(pass), r);
                bool s = portal_seed(map, rq, r, lane, (prow >> lane) & 1);
                uint64_t m = __ballot(s);
                if(lane == r) seeds = mk(m);
            }
        }
    }

    // ---- level-synchronous BFS with bit-sliced distance counters ----------------------------
    // Invariant: every still-open cell carries the current level number in its planes; a cell
    // keeps the value it had when it was removed from `open`.
    u64x open = andn(pass, seeds);
    u64x frontier = seeds;
    u64x pl[NH_MAXP];
#pragma unroll
    for(int k = 0; k < NH_MAXP; k++) pl[k] = u64x{0, 0};

    // One BFS level: expand the frontier, stop when nothing new is reached, bump the shared
    // bit-sliced counter of the still-open cells (bits 0..ctz(level) flip), retire the newly
    // reached cells.  The loop is unrolled by 8 so that the number of low planes that flip is a
    // compile-time constant (1,2,1,3,1,2,1,3+): ~2 plane updates per level instead of a
    // predicated update of all 12.  (Testing the frontier on every second level only was tried:
    // 3571 instead of 3348 VALU instructions per field -- the compiler's code for the untested
    // level is worse than the two instructions saved.)
    int level = 0;
#define NH_BFS_LEVEL(NLOW)                                                                   \
    {                                                                                        \
        u64x nw = bfs_reach(frontier, open);                                                 \
        if(!__any(nz(nw))) break;                                                            \
        level++;                                                                             \
        pl[0] = pl[0] ^ open;                                                                \
        if(NLOW > 1) pl[1] = pl[1] ^ open;                                                   \
        if(NLOW > 2) pl[2] = pl[2] ^ open;                                                   \
        if(NLOW > 3) {                                                                       \
            const int tz = __builtin_ctz(level);      /* wave-uniform: scalar branches */    \
            _Pragma("unroll")                                                                \
            for(int k = 3; k < NH_MAXP; k++) {                                               \
                if(k <= tz) pl[k] = pl[k] ^ open;                                            \
            }                                                                                \
        }                                                                                    \
        open = andn(open, nw);                                                               \
        frontier = nw;                                                                       \
    }
    for(;;) {
        NH_BFS_LEVEL(1) NH_BFS_LEVEL(2) NH_BFS_LEVEL(1) NH_BFS_LEVEL(3)
        NH_BFS_LEVEL(1) NH_BFS_LEVEL(2) NH_BFS_LEVEL(1) NH_BFS_LEVEL(4)
    }
#undef NH_BFS_LEVEL
    const u64x reach = andn(pass, open);          // finite integration value
    // planes of never-reached cells hold garbage: clear them
#pragma unroll
    for(int k = 0; k < NH_MAXP; k++) pl[k] = pl[k] & reach;

    // number of planes that can be non-zero, plus one so that (d-1),(d-2) of d=0,1 cells
    // (all-ones patterns) can never alias a real distance
    const int P = 32 - __builtin_clz((unsigned)level + 1);

    if(WANT_INTEG) {
        float *out = integ + ((size_t)wave << 12);
        for(int c = 0; c < 64; c++) {
            uint32_t d = 0;
#pragma unroll
            for(int k = 0; k < NH_MAXP; k++) d |= (uint32_t)((to64(pl[k]) >> c) & 1) << k;
            bool fin = (to64(reach) >> c) & 1;
            out[lane * 64 + c] = fin ? (float)d : __builtin_inff();
        }
    }

    // ---- bit-sliced bake (field_flow_dir, field.c:355-433) -----------------------------------
    // zero = cells with integration value 0
    u64x anybit = u64x{0, 0};
#pragma unroll
    for(int k = 0; k < NH_MAXP; k++) anybit = anybit | pl[k];
    const u64x zero = andn(reach, anybit);
    const u64x act  = reach & anybit;              // cells that get a direction from the bake

    // Cardinal neighbours.  Adjacent finite cells differ by at most 1, so "neighbour == d-1"
    // is decided by the low two planes: (n + 1) == d  (mod 4).
    const u64x rN = from_n(reach), rS = from_s(reach), rW = from_w(reach), rE = from_e(reach);
    u64x eqN, eqS, eqW, eqE;
    {
        const u64x m0 = pl[0], m1 = pl[1];
        u64x n0, n1;
        n0 = from_n(pl[0]); n1 = from_n(pl[1]); eqN = andn(n0 ^ m0, n1 ^ n0 ^ m1) & rN;
        n0 = from_s(pl[0]); n1 = from_s(pl[1]); eqS = andn(n0 ^ m0, n1 ^ n0 ^ m1) & rS;
        n0 = from_w(pl[0]); n1 = from_w(pl[1]); eqW = andn(n0 ^ m0, n1 ^ n0 ^ m1) & rW;
        n0 = from_e(pl[0]); n1 = from_e(pl[1]); eqE = andn(n0 ^ m0, n1 ^ n0 ^ m1) & rE;
    }

    // Diagonal neighbours: exact comparison against f = d - 2 on all live planes.  A diagonal
    // whose two side tiles are both impassable is only reachable by a detour and may hold ANY
    // distance; the reference still returns it when it *equals* the minimum found through the
    // allowed neighbours (field.c:417-428 re-checks bounds only), so this must be exact.
    u64x neNW = u64x{0, 0}, neNE = neNW, neSW = neNW, neSE = neNW;   // "differs from d-2"
    {
        u64x b1 = u64x{~0u, ~0u};   // borrow chain of d-1
        u64x b2 = u64x{~0u, ~0u};   // borrow chain of (d-1)-1
#pragma unroll
        for(int k = 0; k < NH_MAXP; k++) {
            if(k < P) {
                u64x e = pl[k] ^ b1;  b1 = andn(b1, pl[k]);
                u64x f = e ^ b2;      b2 = andn(b2, e);
                u64x up = from_n(pl[k]), dn = from_s(pl[k]);
                neNW = neNW | (from_w(up) ^ f);
                neNE = neNE | (from_e(up) ^ f);
                neSW = neSW | (from_w(dn) ^ f);
                neSE = neSE | (from_e(dn) ^ f);
            }
        }
    }
    const u64x eqNW = andn(from_w(rN), neNW), eqNE = andn(from_e(rN), neNE);
    const u64x eqSW = andn(from_w(rS), neSW), eqSE = andn(from_e(rS), neSE);

    // diagonal admitted into the minimum only when both side tiles are finite (field.c:382-400)
    const u64x hasD = (eqNW & rN & rW) | (eqNE & rN & rE) | (eqSW & rS & rW) | (eqSE & rS & rE);
    const u64x h = act & hasD;          // minimum is d-2, only diagonals can equal it
    const u64x g = andn(act, hasD);     // minimum is d-1, a cardinal always equals it

    // first match in the order N, S, E, W / NW, NE, SW, SE (field.c:405-428)
    const u64x mN  = g & eqN;
    const u64x mS  = andn(g & eqS, eqN);
    const u64x mE  = andn(g & eqE, eqN | eqS);
    const u64x mW  = andn(g & eqW, eqN | eqS | eqE);
    const u64x mNW = h & eqNW;
    const u64x mNE = andn(h & eqNE, eqNW);
    const u64x mSW = andn(h & eqSW, eqNW | eqNE);
    const u64x mSE = andn(h & eqSE, eqNW | eqNE | eqSW);

    // enum flow_dir bit planes: NW1 N2 NE3 W4 E5 SW6 S7 SE8
    u64x D0 = mNW | mNE | mE | mS;
    u64x D1 = mN | mNE | mSW | mS;
    u64x D2 = mW | mE | mSW | mS;
    u64x D3 = mSE;
    if(rq.type == NAVHIP_TARGET_PORTAL) {          // field_fixup_portal_edges, field.c:830
        uint32_t fd = portal_fix_dir(rq);
        if(fd & 1) D0 = D0 | zero;
        if(fd & 2) D1 = D1 | zero;
        if(fd & 4) D2 = D2 | zero;
    }

    // ---- expand to one byte per cell and write 4 KB coalesced ---------------------------------
    // (out_slot: the request's slot of a field pool, navhip_pool_build; else its own index)
    uint8_t *out = dirs + ((size_t)(out_slot ? out_slot[wave] : wave) << 12);
    uint8_t *st = stage[wib];
    const bool inout = (rq.flags & NAVHIP_REQ_INOUT) != 0;
    if(inout) {
        // existing field: 16 B per lane, coalesced, into the staging tile
#pragma unroll
        for(int j = 0; j < 4; j++)
            *(uint4*)(st + j * 1024 + lane * 16) = *(const uint4*)(out + j * 1024 + lane * 16);
    }
    __builtin_amdgcn_wave_barrier();
    {
        uint32_t *row = (uint32_t*)(st + lane * 64);
        const uint64_t d0 = to64(D0), d1 = to64(D1), d2 = to64(D2), d3 = to64(D3);
        const uint64_t rc = to64(reach);
#pragma unroll
        for(int j = 0; j < 16; j++) {
            // spread 4 plane bits to the low bit of 4 bytes: (n * 0x00204081) & 0x01010101
            uint32_t b0 = (((uint32_t)(d0 >> (4 * j)) & 0xf) * 0x00204081u) & 0x01010101u;
            uint32_t b1 = (((uint32_t)(d1 >> (4 * j)) & 0xf) * 0x00204081u) & 0x01010101u;
            uint32_t b2 = (((uint32_t)(d2 >> (4 * j)) & 0xf) * 0x00204081u) & 0x01010101u;
            uint32_t b3 = (((uint32_t)(d3 >> (4 * j)) & 0xf) * 0x00204081u) & 0x01010101u;
            uint32_t v = b0 | (b1 << 1) | (b2 << 2) | (b3 << 3);
            if(inout) {
                // unreached cells keep the previous byte (field.c:744-745)
                uint32_t rm = ((((uint32_t)(rc >> (4 * j)) & 0xf) * 0x00204081u) & 0x01010101u) * 0xffu;
                v = (v & rm) | (row[j] & ~rm);
            }
            row[j] = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for(int j = 0; j < 4; j++)
        *(uint4*)(out + j * 1024 + lane * 16) = *(const uint4*)(st + j * 1024 + lane * 16);
}

// ---------------------------------------------------------------------------------------------
// k_field_generic : one workgroup per request, arbitrary costs / faction passability
// ---------------------------------------------------------------------------------------------
struct generic_lds {
    uint32_t dist[NH_CELLS];                                   // 16 KB
    __attribute__((aligned(16))) uint8_t pc[NH_CELLS];         // cost, 0xff = not passable
    uint8_t raw[NH_CELLS];      // cost_base as stored
    uint8_t fl[NH_CELLS];       // bit0 field_tile_passable (faction agnostic), bit1 region
    int s_list[128], s_dmin[128], s_nlist, s_D;
};

__device__ void field_generic_one(generic_lds &S, const nh_map_view &map, const navhip_field_req *reqs,
                                  int ri, uint8_t *dirs, float *integ, int force_generic,
                                  const int32_t *out_slot)
{
    uint32_t (&dist)[NH_CELLS] = S.dist;
    uint8_t (&pc)[NH_CELLS] = S.pc;
    uint8_t (&raw)[NH_CELLS] = S.raw;
    uint8_t (&fl)[NH_CELLS] = S.fl;
    int (&s_list)[128] = S.s_list;
    int (&s_dmin)[128] = S.s_dmin;
    int &s_nlist = S.s_nlist, &s_D = S.s_D;

    const int t = threadIdx.x;
    navhip_field_req rq = reqs[ri];
    if(req_uses_bfs(map, rq, force_generic)) return;      // the BFS kernel owns this request
    if(!req_prepare(map, rq)) return;

    const nh_layer_view &L = map.layers[rq.layer];
    const size_t cbase = (size_t)((int)rq.chunk_r * map.w + rq.chunk_c) << 12;
    const bool faction = rq.faction_id != NAVHIP_FACTION_ID_NONE;
    // repair builds: N_FlowFieldUpdateToNearestPathable (field.c:2247) and
    // N_FlowFieldUpdateIslandToNearest (field.c:2307); both work on an existing field
    const bool modeA = rq.type == NAVHIP_TARGET_NEAREST_PATHABLE;
    const bool modeB = !modeA && (rq.flags & NAVHIP_REQ_ISLAND_NEAREST) != 0;

    // ---- stage cost + passability: cell index i = k*256 + t (conflict-free LDS layout) --------
#pragma unroll 4
    for(int k = 0; k < 16; k++) {
        int i = k * 256 + t;
        uint32_t cst = L.cost[cbase + i];
        uint32_t blk = L.blockers ? L.blockers[cbase + i] : 0;
        bool passable;
        if(cst == NAVHIP_COST_IMPASSABLE) {
            passable = false;
        }else if(!faction) {
            passable = (blk == 0);                                        // field.c:117
        }else{
            // field_tile_passable_no_enemies, field.c:179-201
            bool enemies_only = true;
            if(L.factions) {
                const uint8_t *fp = L.factions + cbase * NAVHIP_MAX_FACTIONS + i;
                for(int f = 0; f < NAVHIP_MAX_FACTIONS; f++) {
                    if(fp[(size_t)f << 12] && !(rq.enemies & (1u << f))) { enemies_only = false; break; }
                }
            }
            passable = enemies_only || (blk == 0);
        }
        pc[i] = passable ? (uint8_t)cst : (uint8_t)NAVHIP_COST_IMPASSABLE;
        raw[i] = (uint8_t)cst;
        fl[i] = (cst != NAVHIP_COST_IMPASSABLE && blk == 0) ? 1 : 0;
        dist[i] = NH_INF_U32;
    }
    if(t == 0) { s_nlist = 0; s_D = 0x7fffffff; }
    if(t < 128) s_dmin[t] = 0x7fffffff;
    __syncthreads();

    if(modeA) {
        // ---- field_passable_frontier (field.c:1441): flood the NON-passable 4-connected region of
        // the start tile; the passable tiles bordering it are the frontier
        const int start = rq.tile_r * 64 + rq.tile_c;
        if(t == 0) fl[start] |= 2;
        __syncthreads();
        if(!(fl[start] & 1)) {
            for(;;) {
                int changed = 0;
#pragma unroll 4
                for(int k = 0; k < 16; k++) {
                    int i = k * 256 + t;
                    if(fl[i] & 3) continue;                 // passable, or already in the region
                    int r = i >> 6, c = i & 63;
                    bool nb = (r > 0 && (fl[i - 64] & 2)) || (r < 63 && (fl[i + 64] & 2))
                           || (c > 0 && (fl[i - 1] & 2)) || (c < 63 && (fl[i + 1] & 2));
                    if(nb) { fl[i] |= 2; changed = 1; }
                }
                if(!__syncthreads_or(changed)) break;
            }
#pragma unroll 4
            for(int k = 0; k < 16; k++) {
                int i = k * 256 + t;
                if(!(fl[i] & 1)) continue;
                int r = i >> 6, c = i & 63;
                bool nb = (r > 0 && (fl[i - 64] & 2)) || (r < 63 && (fl[i + 64] & 2))
                       || (c > 0 && (fl[i - 1] & 2)) || (c < 63 && (fl[i + 1] & 2));
                if(nb) dist[i] = 0;
            }
        }else if(t == 0) {
            dist[start] = 0;                                // a passable start is its own frontier
        }
        __syncthreads();
        // ---- field_build_integration_nonpass (field.c:643): relax only NON-passable tiles, step
        // cost = cost_base of the tile entered
        for(;;) {
            int changed = 0;
#pragma unroll 4
            for(int k = 0; k < 16; k++) {
                int i = k * 256 + t;
                if(fl[i] & 1) continue;
                int r = i >> 6, c = i & 63;
                uint32_t m = NH_INF_U32;
                if(r > 0)  m = min(m, dist[i - 64]);
                if(r < 63) m = min(m, dist[i + 64]);
                if(c > 0)  m = min(m, dist[i - 1]);
                if(c < 63) m = min(m, dist[i + 1]);
                if(m < NH_INF_U32) {
                    uint32_t nd = m + raw[i];
                    if(nd < dist[i]) { dist[i] = nd; changed = 1; }
                }
            }
            if(!__syncthreads_or(changed)) break;
        }
    }else{
        // ---- seeds (field_initial_frontier, field.c:1372); for the island repair they only form
        // the frontier from which the nearest island tiles are looked up
        if(rq.type == NAVHIP_TARGET_TILE) {
            if(t == 0) {
                int i = rq.tile_r * 64 + rq.tile_c;
                if(pc[i] != NAVHIP_COST_IMPASSABLE) {
                    if(modeB) s_list[s_nlist++] = i; else dist[i] = 0;
                }
            }
        }else{
            int nr = rq.port_r1 - rq.port_r0 + 1, nc = rq.port_c1 - rq.port_c0 + 1;
            for(int j = t; j < nr * nc; j += 256) {
                int r = rq.port_r0 + j / nc, c = rq.port_c0 + j % nc;
                if(portal_seed(map, rq, r, c, pc[r * 64 + c] != NAVHIP_COST_IMPASSABLE)) {
                    if(modeB) { int p = atomicAdd(&s_nlist, 1); if(p < 128) s_list[p] = r * 64 + c; }
                    else dist[r * 64 + c] = 0;
                }
            }
        }
        __syncthreads();
        if(modeB) {
            // completely blocked target: retry ignoring blockers (field.c:2352-2355; only the tile
            // frontier honours ignoreblock, field.c:1112-1115)
            if(t == 0 && s_nlist == 0 && rq.type == NAVHIP_TARGET_TILE)
                s_list[s_nlist++] = rq.tile_r * 64 + rq.tile_c;
            __syncthreads();
            const int nl = min(s_nlist, 128);
            const uint16_t *li = L.local_islands + cbase;
            const uint16_t *gi = L.islands ? L.islands + cbase : nullptr;
            const uint32_t want = rq.aux_iid;
            // field_closest_tiles_local (field.c:1010) for every frontier tile: the qualifying tiles
            // (passable, unblocked, local island `want`, same global island as the frontier tile) at
            // the minimum Manhattan distance
            for(int q = 0; q < nl; q++) {
                const int f = s_list[q], fr = f >> 6, fc = f & 63;
                const uint32_t giid = gi ? gi[f] : NAVHIP_ISLAND_NONE;
                int best = 0x7fffffff;
#pragma unroll 4
                for(int k = 0; k < 16; k++) {
                    int i = k * 256 + t;
                    if(!(fl[i] & 1)) continue;
                    if(want != NAVHIP_ISLAND_NONE && li[i] != want) continue;
                    if(giid != NAVHIP_ISLAND_NONE && gi[i] != giid) continue;
                    int d = abs((i >> 6) - fr) + abs((i & 63) - fc);
                    best = min(best, d);
                }
                if(best < 0x7fffffff) atomicMin(&s_dmin[q], best);
            }
            __syncthreads();
            if(t == 0) {
                int D = 0x7fffffff;
                for(int q = 0; q < nl; q++) D = min(D, s_dmin[q]);
                s_D = D;
            }
            __syncthreads();
            const int D = s_D;
            if(D < 0x7fffffff) {
                for(int q = 0; q < nl; q++) {
                    if(s_dmin[q] != D) continue;
                    const int f = s_list[q], fr = f >> 6, fc = f & 63;
                    const uint32_t giid = gi ? gi[f] : NAVHIP_ISLAND_NONE;
#pragma unroll 4
                    for(int k = 0; k < 16; k++) {
                        int i = k * 256 + t;
                        if(!(fl[i] & 1)) continue;
                        if(want != NAVHIP_ISLAND_NONE && li[i] != want) continue;
                        if(giid != NAVHIP_ISLAND_NONE && gi[i] != giid) continue;
                        if(abs((i >> 6) - fr) + abs((i & 63) - fc) == D) dist[i] = 0;
                    }
                }
            }
            __syncthreads();
        }

        // ---- chaotic (in-place) min-plus relaxation to the fixpoint --------------------------
        // d[x] = min(d[x], min_{4-nbrs y} d[y] + cost[x]) for passable x.  Values only decrease and
        // never drop below the true shortest distance, so the fixpoint is the Dijkstra result of
        // field_build_integration bit for bit; an iteration in which no thread lowered anything
        // proves every read saw final values.
        for(;;) {
            int changed = 0;
#pragma unroll 4
            for(int k = 0; k < 16; k++) {
                int i = k * 256 + t;
                uint32_t cst = pc[i];
                if(cst == NAVHIP_COST_IMPASSABLE) continue;
                int r = i >> 6, c = i & 63;
                uint32_t m = NH_INF_U32;
                if(r > 0)  m = min(m, dist[i - 64]);
                if(r < 63) m = min(m, dist[i + 64]);
                if(c > 0)  m = min(m, dist[i - 1]);
                if(c < 63) m = min(m, dist[i + 1]);
                if(m < NH_INF_U32) {
                    uint32_t nd = m + cst;
                    if(nd < dist[i]) { dist[i] = nd; changed = 1; }
                }
            }
            if(!__syncthreads_or(changed)) break;
        }
    }

    // ---- bake (field_flow_dir, field.c:355-433) + fixup; result staged in pc[] ----------------
    const bool inout = (rq.flags & NAVHIP_REQ_INOUT) != 0 || modeA || modeB;
    uint8_t *out = dirs + ((size_t)(out_slot ? out_slot[ri] : ri) << 12);
    const uint32_t fixdir = (rq.type == NAVHIP_TARGET_PORTAL) ? portal_fix_dir(rq) : NAVHIP_FD_NONE;
    uint8_t res[16];
#pragma unroll 4
    for(int k = 0; k < 16; k++) {
        int i = k * 256 + t;
        int r = i >> 6, c = i & 63;
        uint32_t d = dist[i];
        uint32_t dir;
        if(d >= NH_INF_U32) {
            dir = inout ? out[i] : NAVHIP_FD_NONE;
        }else if(d == 0) {
            dir = modeA ? out[i] : fixdir;           // field.c:2297: frontier tiles are left alone
        }else{
            const uint32_t I = NH_INF_U32;
            uint32_t dn = r > 0  ? dist[i - 64] : I, ds = r < 63 ? dist[i + 64] : I;
            uint32_t dw = c > 0  ? dist[i - 1]  : I, de = c < 63 ? dist[i + 1]  : I;
            uint32_t dnw = (r > 0  && c > 0)  ? dist[i - 65] : I;
            uint32_t dne = (r > 0  && c < 63) ? dist[i - 63] : I;
            uint32_t dsw = (r < 63 && c > 0)  ? dist[i + 63] : I;
            uint32_t dse = (r < 63 && c < 63) ? dist[i + 65] : I;
            uint32_t mc = min(min(dn, ds), min(dw, de));
            if(dn < I && dw < I) mc = min(mc, dnw);
            if(dn < I && de < I) mc = min(mc, dne);
            if(ds < I && dw < I) mc = min(mc, dsw);
            if(ds < I && de < I) mc = min(mc, dse);
            if(dn == mc)       dir = NAVHIP_FD_N;
            else if(ds == mc)  dir = NAVHIP_FD_S;
            else if(de == mc)  dir = NAVHIP_FD_E;
            else if(dw == mc)  dir = NAVHIP_FD_W;
            else if(dnw == mc) dir = NAVHIP_FD_NW;
            else if(dne == mc) dir = NAVHIP_FD_NE;
            else if(dsw == mc) dir = NAVHIP_FD_SW;
            else               dir = NAVHIP_FD_SE;
        }
        res[k] = (uint8_t)dir;
        if(integ) integ[((size_t)ri << 12) + i] = (d >= NH_INF_U32) ? __builtin_inff() : (float)d;
    }
    __syncthreads();            // every read of pc[] (none left) / out[] done before overwrite
#pragma unroll
    for(int k = 0; k < 16; k++) pc[k * 256 + t] = res[k];
    __syncthreads();
    *(uint4*)(out + t * 16) = *(const uint4*)(pc + t * 16);
}

// The launch is a fixed, small grid: workgroups stride over the list of requests the BFS kernel
// declined (gen_list: two counters, then the ids; gen_list == nullptr: every request, in order).  A
// tick whose requests are all unit-cost BFS fields pays for a few hundred workgroups that read one
// counter, not for one empty workgroup per request.  The two counters alternate between launches:
// this launch reads counter `gen_slot` and zeroes the other one for the next launch (which starts
// after this one in stream order) -- no reset pass, no completion atomics.
__global__ __launch_bounds__(256) void k_field_generic(nh_map_view map, const navhip_field_req *reqs,
                                                       int n, uint8_t *dirs, float *integ,
                                                       int force_generic, int32_t *gen_list, int gen_slot,
                                                       const int32_t *out_slot)
{
    __shared__ generic_lds S;
    const int count = gen_list ? gen_list[gen_slot] : n;
    if(gen_list && blockIdx.x == 0 && threadIdx.x == 0) gen_list[gen_slot ^ 1] = 0;
    for(int w = blockIdx.x; w < count; w += gridDim.x) {
        const int ri = gen_list ? gen_list[2 + w] : w;
        field_generic_one(S, map, reqs, ri, dirs, integ, force_generic, out_slot);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
void nh_launch_fields(navhip_ctx *ctx, const navhip_field_req *d_reqs, int n, uint8_t *d_dirs,
                      float *d_integ, int32_t *d_gen_list, hipStream_t s, const int32_t *d_out_slot)
{

    nh_map_view mv;
    mv.w = ctx->w;
    mv.h = ctx->h;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        const navhip_layer &L = ctx->layers[l];
        mv.layers[l] = nh_layer_view{L.cost, L.blockers, L.local_islands, L.factions,
                                     L.passmask, L.unit_cost, L.changed, L.islands};
    }
    const int force_generic = ctx->field_kernel_mode == 1;
    // (only launches that use the list alternate its counters)
    const int gen_slot = force_generic ? 0 : (int)(ctx->gen_launches++ & 1);
    if(!force_generic) {
        dim3 grid((n + BFS_WAVES - 1) / BFS_WAVES);
        if(d_integ)
            hipLaunchKernelGGL(k_field_bfs<true>, grid, dim3(BFS_WAVES * 64), 0, s, mv, d_reqs, n, d_dirs,
                               d_integ, force_generic, d_gen_list, gen_slot, d_out_slot);
        else
            hipLaunchKernelGGL(k_field_bfs<false>, grid, dim3(BFS_WAVES * 64), 0, s, mv, d_reqs, n, d_dirs,
                               d_integ, force_generic, d_gen_list, gen_slot, d_out_slot);
    }
    // The relaxation kernel strides over the requests the BFS kernel declined.  Usually that list is empty
    // or short (repairs, attacking paths): 512 workgroups that mostly read one counter.  A map with real
    // cost gradients sends EVERY request there: eight workgroups per CU (20 KB of LDS each) instead of two.
    bool heavy = force_generic;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) heavy = heavy || (ctx->layers[l].cost && ctx->layers[l].nonunit_costs);
    const int gmax = heavy ? 2048 : 512;
    hipLaunchKernelGGL(k_field_generic, dim3(n < gmax ? n : gmax), dim3(256), 0, s, mv, d_reqs, n, d_dirs,
                       d_integ, force_generic, force_generic ? (int32_t*)nullptr : d_gen_list, gen_slot, d_out_slot);
}
