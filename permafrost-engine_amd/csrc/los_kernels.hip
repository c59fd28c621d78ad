// los_kernels.hip -- line-of-sight fields for gfx950.
//
// Reference semantics: N_LOSFieldCreate, navigation/field.c:2085 (+ field_neighbours_grid_los :304,
// field_is_los_corner :435, field_create_wavefront_blocked_line :463, field_pad_wavefront :519).
// A wavefront expands from the destination tile (or from the visible tiles of the edge shared with
// the previous chunk's LOS field); whenever it meets an impassable tile that is a LOS corner it
// draws a Bresenham "wavefront blocked" line from that corner away from the destination, and tiles
// on such lines are never entered.  The result depends on the ORDER in which tiles leave the
// frontier: among tiles of equal distance it is the pop order of the reference's binary heap
// (lib/public/pqueue.h:112-191, strict comparisons, hole-style sift-down).  To be bit-exact the
// kernel therefore emulates that heap operation for operation: the wavefront is inherently
// sequential, so ONE LANE runs it out of LDS while the wave's other lanes only help with the
// staging, the final padding and the coalesced store.  Parallelism comes from the batch: one wave
// per LOS field -- so what bounds the throughput is how many fields a CU holds at once, i.e. LDS:
//   * the heap only ever holds TWO consecutive priorities (the popped level p and p + 1: every push is
//     p + 1, every entry was pushed by a tile popped at a level <= p), so a node needs no stored
//     priority: its level's parity rides in bit 12 of the tile index (one 16-bit word per node);
//   * the frontier of a 64 x 64 field is a few hundred tiles: the heap gets 1 022 nodes (2 KB), and a
//     field whose heap would overflow is left to a second launch of the same kernel with room for all
//     4 096 tiles (one look at a flag per request when nothing overflowed).
// 6 KB of LDS per field instead of 21: 26 fields per CU instead of 7.
//
// LOS fields are built once per (destination, chunk) when a path is planned and then cached
// (nav.c:1840-1847,2026-2039), so this kernel is latency- not throughput-critical.
#include "navhip_internal.h"

#define LF_VISIBLE   0x01      /* struct LOS_field bit 0 (field.h:48-54)                     */
#define LF_WFB       0x02      /* struct LOS_field bit 1: wavefront_blocked                  */
#define LF_INHEAP    0x04
#define LF_ASSIGNED  0x08      /* integration value finite                                    */
#define LF_COSTLY    0x10      /* neighbour cost > 1: cost_base > 1 or not passable (:337-348) */
#define LF_RAWBLK    0x20      /* cost_base == 0xff || blockers > 0 (field_is_los_corner)     */

struct los_heap { int size; };      // nodes live in LDS: node[1..size] = tile | (priority & 1) << 12

#define LH_TILE(v)  ((int)((v) & 0x0fff))
// priority of a node relative to the level `lo` being popped: 0 (== lo) or 1 (== lo + 1)
#define LH_REL(v, lo) ((int)((((v) >> 12) ^ (lo)) & 1))

// pq_coord_push, pqueue.h:150.  p is lo or lo + 1 and no node is above lo + 1: `prio[parent] > p` can only
// hold for p == lo under a parent at lo + 1.
__device__ __forceinline__ void lh_push(los_heap &h, uint16_t *node, int lo, int p, int c)
{
    int curr = h.size + 1, parent = curr / 2;
    const int rel = (p ^ lo) & 1;
    while(curr > 1 && LH_REL(node[parent], lo) > rel) {
        node[curr] = node[parent];
        curr = parent;
        parent = parent / 2;
    }
    node[curr] = (uint16_t)(c | ((p & 1) << 12));
    h.size++;
}

__device__ __forceinline__ float los_len(float x, float z) { return __builtin_sqrtf(x * x + z * z); }

// field_create_wavefront_blocked_line, field.c:463
__device__ void los_blocked_line(uint8_t *fl, float map_x, float map_z, const navhip_los_req &rq,
                                 int corner_r, int corner_c)
{
    // M_Tile_Bounds (tile.c:356) centres of the target and the corner tile
    const float tbx = (map_x - (float)(rq.target_chunk_c * 256)) - (float)(rq.target_tile_c * 4);
    const float tbz = (map_z + (float)(rq.target_chunk_r * 256)) + (float)(rq.target_tile_r * 4);
    const float cbx = (map_x - (float)(rq.chunk_c * 256)) - (float)(corner_c * 4);
    const float cbz = (map_z + (float)(rq.chunk_r * 256)) + (float)(corner_r * 4);
    const float tcx = tbx - 4.0f / 2.0f, tcz = tbz + 4.0f / 2.0f;
    const float ccx = cbx - 4.0f / 2.0f, ccz = cbz + 4.0f / 2.0f;
    float sxf = tcx - ccx, szf = tcz - ccz;
    const float len = los_len(sxf, szf);
    sxf = __fdiv_rn(sxf, len);
    szf = __fdiv_rn(szf, len);
    const int dx = abs((int)(sxf * 1000));
    const int dy = -abs((int)(szf * 1000));
    const int sx = sxf > 0.0f ? 1 : -1;
    const int sy = szf < 0.0f ? 1 : -1;
    int err = dx + dy;
    int r = corner_r, c = corner_c;
    do {
        fl[r * 64 + c] |= LF_WFB;
        const int e2 = 2 * err;
        if(e2 >= dy) { err += dy; c += sx; }
        if(e2 <= dx) { err += dx; r += sy; }
    } while(r >= 0 && r < 64 && c >= 0 && c < 64);
}

// field_is_los_corner, field.c:435
__device__ __forceinline__ bool los_corner(const uint8_t *fl, int r, int c)
{
    if(r > 0 && r < 63) {
        bool a = (fl[(r - 1) * 64 + c] & LF_RAWBLK) != 0, b = (fl[(r + 1) * 64 + c] & LF_RAWBLK) != 0;
        if(a ^ b) return true;
    }
    if(c > 0 && c < 63) {
        bool a = (fl[r * 64 + c - 1] & LF_RAWBLK) != 0, b = (fl[r * 64 + c + 1] & LF_RAWBLK) != 0;
        if(a ^ b) return true;
    }
    return false;
}

// CAP: heap nodes.  overflow[ri]: set by the CAP = 1022 launch for a field it gave up on; the CAP = 4096
// launch (REDO) only builds those.
template <int CAP, bool REDO>
__global__ __launch_bounds__(64) void k_los_field(nh_map_view map, const navhip_los_req *reqs, int n,
                                                  const uint8_t *prev_fields, uint8_t *out_fields,
                                                  float map_x, float map_z, uint8_t *overflow)
{
    __shared__ __attribute__((aligned(4))) uint16_t h_node[CAP + 2];
    __shared__ __attribute__((aligned(16))) uint8_t fl[NH_CELLS];
    __shared__ int s_over;
    const int ri = blockIdx.x, lane = threadIdx.x;
    if(ri >= n) return;
    if(REDO && !overflow[ri]) return;
    if(lane == 0) s_over = 0;
    const navhip_los_req rq = reqs[ri];
    const nh_layer_view &L = map.layers[rq.layer];
    const size_t cbase = (size_t)((int)rq.chunk_r * map.w + rq.chunk_c) << 12;
    const bool faction = rq.faction_id != NAVHIP_FACTION_ID_NONE;

    // ---- stage the per-tile predicates (all lanes) ----------------------------------------------
    for(int k = 0; k < 64; k++) {
        const int i = k * 64 + lane;
        const uint32_t cst = L.cost[cbase + i];
        const uint32_t blk = L.blockers ? L.blockers[cbase + i] : 0;
        bool passable;
        if(cst == NAVHIP_COST_IMPASSABLE) {
            passable = false;
        }else if(!faction) {
            passable = blk == 0;
        }else{
            bool enemies_only = true;
            if(L.factions) {
                const uint8_t *fp = L.factions + cbase * NAVHIP_MAX_FACTIONS + i;
                for(int f = 0; f < NAVHIP_MAX_FACTIONS; f++)
                    if(fp[(size_t)f << 12] && !(rq.enemies & (1u << f))) { enemies_only = false; break; }
            }
            passable = enemies_only || blk == 0;
        }
        uint8_t v = 0;
        if(!passable || cst > 1) v |= LF_COSTLY;
        if(cst == NAVHIP_COST_IMPASSABLE || blk > 0) v |= LF_RAWBLK;
        fl[i] = v;
    }
    __syncthreads();

    // ---- the sequential wavefront (lane 0) -------------------------------------------------------
    if(lane == 0) {
        los_heap H;
        H.size = 0;
        const bool first = rq.prev_dr == 0 && rq.prev_dc == 0;
        if(first) {
            // case 1, field.c:2111-2115: the destination chunk
            const int t = rq.target_tile_r * 64 + rq.target_tile_c;
            lh_push(H, h_node, 0, 0, t);
            fl[t] |= LF_ASSIGNED | LF_INHEAP;
        }else{
            // case 2, field.c:2122-2193: carry the shared edge over from the previous chunk's field
            const uint8_t *prev = prev_fields + ((size_t)ri << 12);
            const bool horizontal = rq.prev_dr == 0;
            int curr_edge, prev_edge;
            if(!horizontal) { curr_edge = rq.prev_dr < 0 ? 0 : 63; prev_edge = rq.prev_dr < 0 ? 63 : 0; }
            else            { curr_edge = rq.prev_dc < 0 ? 0 : 63; prev_edge = rq.prev_dc < 0 ? 63 : 0; }
            for(int k = 0; k < 64; k++) {
                const int ci = horizontal ? k * 64 + curr_edge : curr_edge * 64 + k;
                const int pi = horizontal ? k * 64 + prev_edge : prev_edge * 64 + k;
                const uint8_t pv = prev[pi] & (LF_VISIBLE | LF_WFB);
                fl[ci] = (uint8_t)((fl[ci] & ~(LF_VISIBLE | LF_WFB)) | pv);
                if(pv & LF_WFB)
                    los_blocked_line(fl, map_x, map_z, rq, ci >> 6, ci & 63);
                if(fl[ci] & LF_VISIBLE) {
                    lh_push(H, h_node, 0, 0, ci);
                    fl[ci] |= LF_ASSIGNED | LF_INHEAP;
                }
            }
        }
        int cprio = 0;                          // the level being popped (the root's priority)
        bool over = false;
        int hsize = H.size;
        while(hsize > 0 && !over) {
            const uint16_t top = h_node[1];
            const int cur = LH_TILE(top);
            cprio += LH_REL(top, cprio);          // the root is the minimum: lo, or lo + 1 once lo is used up
            // ---- pq_coord_pop + _pq_balance (pqueue.h:112-133,173-183), specialised for two priorities: the
            // last node moves to the root; one of the current level stays there (no child is strictly
            // smaller), one of the next level sinks below the nodes of the current level -- at every node
            // the left child if it is of the current level, else the right one if it is, else it stops.
            // Both children come in one 32-bit read.
            {
                const uint16_t lastv = h_node[hsize];
                hsize--;
                int root = 1;
                if(LH_REL(lastv, cprio)) {
                    for(;;) {
                        const int l = root * 2;
                        if(l > hsize) break;
                        const uint32_t pair = *(const uint32_t*)&h_node[l];
                        const uint16_t lv = (uint16_t)(pair & 0xffffu), rv = (uint16_t)(pair >> 16);
                        if(LH_REL(lv, cprio) == 0)                       { h_node[root] = lv; root = l; }
                        else if(l < hsize && LH_REL(rv, cprio) == 0)     { h_node[root] = rv; root = l + 1; }
                        else break;
                    }
                }
                h_node[root] = lastv;
            }
            const int r = cur >> 6, c = cur & 63;
            // field_neighbours_grid_los :304: the 4 neighbours that are not wavefront blocked, collected
            // BEFORE any of them is processed -- their flag bytes are fetched together (processing one
            // neighbour changes another's byte only by drawing a blocked line: `redraw`)
            const int ni[4] = {cur - 64, cur - 1, cur + 1, cur + 64};
            const bool have[4] = {r > 0, c > 0, c < 63, r < 63};
            uint8_t nf[4];
#pragma unroll
            for(int k = 0; k < 4; k++) nf[k] = have[k] ? fl[ni[k]] : (uint8_t)LF_WFB;
            fl[cur] &= (uint8_t)~LF_INHEAP;
            bool redraw = false;
#pragma unroll
            for(int k = 0; k < 4; k++) {
                if(nf[k] & LF_WFB) continue;
                if(nf[k] & LF_COSTLY) {
                    if(!los_corner(fl, ni[k] >> 6, ni[k] & 63)) continue;
                    los_blocked_line(fl, map_x, map_z, rq, ni[k] >> 6, ni[k] & 63);
                    redraw = true;
                }else{
                    // unit steps popped in non-decreasing order: `new_cost < integration[n]` holds exactly
                    // when n has no value yet; and a push (pq_coord_push, pqueue.h:150) carries the highest
                    // priority in the heap, so its sift-up never moves anything: an append
                    uint8_t add = LF_VISIBLE;
                    if(!(nf[k] & LF_ASSIGNED)) {
                        add |= LF_ASSIGNED;
                        if(!(nf[k] & LF_INHEAP)) {
                            if(hsize >= CAP) { over = true; break; }
                            h_node[++hsize] = (uint16_t)(ni[k] | (((cprio + 1) & 1) << 12));
                            add |= LF_INHEAP;
                        }
                    }
                    fl[ni[k]] = (uint8_t)((redraw ? fl[ni[k]] : nf[k]) | add);
                }
            }
        }
        s_over = over ? 1 : 0;
        if(!REDO) overflow[ri] = over ? 1 : 0;
    }
    __syncthreads();
    if(s_over) return;                     // (left to the launch with room for every tile)

    // ---- field_pad_wavefront (:519): tiles within one tile of a blocked tile are not visible.
    // In place: only VISIBLE bits are cleared while only WFB bits are read.
    for(int k = 0; k < 64; k++) {
        const int i = k * 64 + lane, r = k, c = lane;
        bool near = false;
        for(int rr = r - 1; rr <= r + 1; rr++)
            for(int cc = c - 1; cc <= c + 1; cc++)
                if(rr >= 0 && rr < 64 && cc >= 0 && cc < 64 && (fl[rr * 64 + cc] & LF_WFB)) near = true;
        if(near) fl[i] &= (uint8_t)~LF_VISIBLE;
    }
    __syncthreads();
    for(int k = 0; k < 64; k++) fl[k * 64 + lane] &= (uint8_t)(LF_VISIBLE | LF_WFB);
    __syncthreads();
    uint8_t *out = out_fields + ((size_t)ri << 12);
#pragma unroll
    for(int j = 0; j < 4; j++)
        *(uint4*)(out + j * 1024 + lane * 16) = *(const uint4*)(fl + j * 1024 + lane * 16);
}

void nh_launch_los(navhip_ctx *ctx, const navhip_los_req *d_reqs, int n, const uint8_t *d_prev,
                   uint8_t *d_out, float map_x, float map_z, hipStream_t s)
{
    nh_map_view mv;
    mv.w = ctx->w;
    mv.h = ctx->h;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        const navhip_layer &L = ctx->layers[l];
        mv.layers[l] = nh_layer_view{L.cost, L.blockers, L.local_islands, L.factions,
                                     L.passmask, L.unit_cost, L.changed, L.islands};
    }
    if(n > 0) {
        uint8_t *d_over = nullptr;
        if(navhip_stage_reserve(ctx, 42, (size_t)n, (void**)&d_over) != NAVHIP_OK) return;
        hipLaunchKernelGGL((k_los_field<1022, false>), dim3(n), dim3(64), 0, s, mv, d_reqs, n, d_prev, d_out, map_x,
                           map_z, d_over);
        hipLaunchKernelGGL((k_los_field<4096, true>), dim3(n), dim3(64), 0, s, mv, d_reqs, n, d_prev, d_out, map_x,
                           map_z, d_over);
    }
}
