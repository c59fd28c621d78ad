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
// Arithmetic mirrors the reference's C expression by expression (agent_math.h); all
// order-dependent float sums are evaluated in the reference's own order.
//
// Kernels of one tick (navhip_api.hip wires the streams):
//   k_sp_count .. k_sp_place   device spatial hash: fixed-point cell binning, scan, and the POOL -- one
//                 16-byte record {pos, radius, flag bits | uid} + one velocity per inserted entity, in
//                 the cell order and per-cell order bg_ent_cleanup produces after inserting uids
//                 0..n-1, so that a query's candidates are contiguous runs and a hit needs no
//                 second gather.
//   k_agent_nbr   one ROW of 16 lanes per entity, pool order: separation force + ClearPath neighbour
//                 lists in one walk (agent_group.h).  Needs only the snapshot: runs beside the field
//                 builds.
//   k_cohesion    the O(N*F) exp-weighted flock centroid, four lanes per member.
//   k_agent_mid   one THREAD per entity: the scalar chain -- flow sampling, arrive force, priority
//                 ladder -> preferred velocity.  Agents without ClearPath neighbours are truncated +
//                 position-tested here; the rest go to device-side work lists by neighbour count.
//   k_cp_small / k_cp_rows / k_cp_heavy
//                 ClearPath for the listed agents: a ROW of 16 lanes per agent -- 1..4 neighbours on a
//                 lean kernel of its own (one attempt, no queue; eight waves per SIMD), 5..16 on the
//                 general search --, a WORKGROUP per agent with 17..64 (a crowd: its waves share the ray
//                 pairs); lanes spread over cones / ray pairs, branch and bound on the distance to
//                 des_v, a candidate queue, a lexicographic arg-min that reproduces the reference's
//                 first-wins rule; work units numbered heaviest first, drawn by tickets.  Rows on the
//                 caller's stream, the rest beside them on a side stream.
//   k_agent_full  one WAVE per listed agent, the whole step (irregular gathers: garrisoned
//                 neighbours, wide queries).
#include "navhip_internal.h"
#include "agent_internal.h"
#include "agent_group.h"

__constant__ double c_exp2_64[64] = { NH_EXP2_64_TABLE };

// ---------------------------------------------------------------------------------------------
// spatial hash (bitmap_grid.h): build
// ---------------------------------------------------------------------------------------------
#define SP_MAX_QUERY_R 30   /* largest query radius of the movement tick (separation, movement.c:1695) */
// Optional slab filter: when a rank steps only the entities [work_begin, work_end), nothing farther
// than the largest query radius of the tick (r = 30) from the bounding box of THOSE entities can be
// returned by any of its queries, and leaving such entities out changes neither the order nor the
// caps of what is returned.  box = {max(-ix), max(ix), max(-iy), max(iy)} over the slab in the
// x256 fixed point the queries compare in; INT_MIN-initialised.
// (Four-wave workgroups: a 1024-thread block waits for sixteen free wave slots on one CU -- 26 us beside the
// field builds, in front of the whole spatial hash.  Two boxes alternate between builds: the build that
// consumes box[p] (k_sp_count) re-initialises box[p ^ 1] for its successor, so there is no memset.)
__global__ __launch_bounds__(256) void k_sp_bbox(const float *pos_xz, int begin, int end, int32_t *box)
{
    __shared__ int32_t part[4][4];
    int32_t v[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    for(int i = begin + blockIdx.x * 256 + threadIdx.x; i < end; i += gridDim.x * 256) {
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
        for(int w = 0; w < 4; w++) m = max(m, part[w][threadIdx.x]);
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
// Pass 1: cell of every entity + its arrival rank in the cell (the counters are zero on entry:
// cleared at allocation, then by k_sp_scan_add of the previous build).
__global__ __launch_bounds__(256) void k_sp_count(nh_grid G, const float *pos_xz, int n,
                                                  int32_t *ent_cell, int32_t *ent_rank,
                                                  int32_t *cell_count, const int32_t *box, int32_t *box_next,
                                                  int32_t *n_active)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if(box_next && i < 4) box_next[i] = INT32_MIN;           // (the box of the NEXT build)
    if(n_active && i == 0) *n_active = 0;                    // (the list k_sp_place fills: its reader, the last walk, is through)
    if(i >= n) return;
    const int32_t ix = bg_scale(pos_xz[2 * i]), iy = bg_scale(pos_xz[2 * i + 1]);
    if(!sp_in_box(box, ix, iy)) { ent_cell[i] = -1; return; }
    const int c = sp_cell_of(G, ix, iy);
    ent_cell[i] = c;
    ent_rank[i] = atomicAdd(&cell_count[c], 1);
}

// A rank that steps a slab only fills the cells its box (+ the query reach) covers; every other cell is
// empty and is never looked at by a query of the slab either.  The scans skip the blocks of cells that lie
// entirely in grid rows outside the box: their block sum is 0 and their cell_start entries stay unwritten.
// (rows [r0, r1] of the box in cells; a block is a run of NH_SCAN_T consecutive cells, row-major)
__device__ __forceinline__ bool sp_block_outside_box(const nh_grid &G, const int32_t *box, int first_cell, int ncells)
{
    if(!box) return false;
    const int32_t m = SP_MAX_QUERY_R * 256 + 256;
    const int64_t y0 = -(int64_t)box[2] - m, y1 = (int64_t)box[3] + m;          // fixed-point rows of the box
    if(y1 < y0) return true;                                                     // empty slab: nothing is inserted
    // cell rows (clamped like sp_cell_of clamps an element into the grid)
    const int64_t r0 = min(max((y0 - G.origin_y) >> 12, (int64_t)0), (int64_t)G.grid_h - 1);
    const int64_t r1 = min(max((y1 - G.origin_y) >> 12, (int64_t)0), (int64_t)G.grid_h - 1);
    const int last_cell = min(first_cell + NH_SCAN_T, ncells) - 1;
    // (one row more at the end: a query reads cell_start one past its last cell -- the first cell of the
    // row after r1)
    return last_cell < r0 * G.grid_w || first_cell >= (r1 + 2) * G.grid_w;
}

// exclusive scan of cell_count[0..ncells) -> cell_start[0..ncells], two passes over NH_SCAN_T-cell
// blocks: (1) block-local exclusive scan + block totals, (2) add the sum of the preceding totals.
// (Blocks of four waves: a 1024-thread block needs sixteen free wave slots on ONE compute unit at the
// same moment, and beside the cohesion kernel's stream of one-wave blocks it waited for them for
// 50 us -- on the critical path of the tick.)
__global__ __launch_bounds__(NH_SCAN_T) void k_sp_scan_local(const int32_t *cell_count, int32_t *cell_start,
                                                             int32_t *block_sum, int ncells,
                                                             nh_grid G, const int32_t *box)
{
    __shared__ int32_t wsum[NH_SCAN_T / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if(sp_block_outside_box(G, box, blockIdx.x * NH_SCAN_T, ncells)) {
        if(t == 0) block_sum[blockIdx.x] = 0;
        return;
    }
    const int i = blockIdx.x * NH_SCAN_T + t;
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
    for(int k = 0; k < NH_SCAN_T / 64; k++) {
        int32_t x = wsum[k];
        if(k < w) woff += x;
        tot += x;
    }
    if(i < ncells) cell_start[i] = woff + incl - v;
    if(t == 0) block_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(NH_SCAN_T) void k_sp_scan_add(int32_t *cell_start, const int32_t *block_sum,
                                                           int ncells, int nblocks, int32_t *zero_counts,
                                                           nh_grid G, const int32_t *box)
{
    __shared__ int32_t wsum[NH_SCAN_T / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // (the last block always runs: it writes the grand total, cell_start[ncells])
    if((int)blockIdx.x != nblocks - 1 && sp_block_outside_box(G, box, blockIdx.x * NH_SCAN_T, ncells)) return;
    // sum of the totals of the blocks before this one (and, in the last block, of all blocks)
    int32_t part = 0, all = 0;
    for(int k = t; k < nblocks; k += NH_SCAN_T) {
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
    for(int k = 0; k < NH_SCAN_T / 64; k++) tot += wsum[k];
    __syncthreads();
    if(lane == 0) wsum[w] = part;
    __syncthreads();
    int32_t off = 0;
#pragma unroll
    for(int k = 0; k < NH_SCAN_T / 64; k++) off += wsum[k];
    const int i = blockIdx.x * NH_SCAN_T + t;
    if(i < ncells) cell_start[i] += off;
    if(zero_counts && i < ncells) zero_counts[i] = 0;     // consumed by k_sp_scan_local: clean for the next build
    if(last && t == 0) cell_start[ncells] = tot;
}
// Pass 3: entities into their cell's range in arrival order (no atomics: the rank is known)
__global__ __launch_bounds__(256) void k_sp_scatter(const int32_t *ent_cell, const int32_t *ent_rank, int n,
                                                    const int32_t *cell_start, int32_t *tmp_id)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= n) return;
    const int c = ent_cell[i];
    if(c < 0) return;                            // outside the slab filter
    tmp_id[cell_start[c] + ent_rank[i]] = i;
}

// Pass 4: per-cell order + the pool records.  bg_ent_insert pushes at the head of the cell's
// overflow chain and bg_ent_cleanup copies the chain head-first (bitmap_grid.h:1102-1121,
// 1515-1521), so after inserting uids 0..n-1 each cell holds its elements in DESCENDING uid order:
// the final slot of an element is its cell's start + the number of cell mates with a larger uid
// (one thread per element counts them: cells hold a handful of elements).
#define SP_BLOCK 64
__global__ __launch_bounds__(SP_BLOCK) void k_sp_place(nh_grid G, const float *pos_xz, nh_pack_src src,
                                                  const int32_t *ent_cell, const int32_t *tmp_id,
                                                  int n, int work_begin, int work_end,
                                                  float4 *recA, float2 *recV, int32_t *pool_of,
                                                  int32_t *active, int32_t *n_active)
{
    // one thread per ENTITY (not per pool slot): its inputs are coalesced loads that do not wait for the
    // slot search, the only gathers are the cell's bounds and its handful of ids, and the record goes out
    // as a scattered store.  (Per slot the kernel was a chain of five dependent gathers -- id, cell, bounds,
    // cell mates, the entity's six attribute arrays -- and took 60 us beside the cohesion kernel.)
    const int i = blockIdx.x * SP_BLOCK + threadIdx.x;
    const int c = i < n ? ent_cell[i] : -1;      // (-1: outside the slab filter)
    int slot = -1;
    bool walks = false;
    if(c >= 0) {
        float4 a;
        float2 v;
        pool_record(i, pos_xz, src, work_begin, work_end, a, v);
        const int b = G.cell_start[c], e = G.cell_start[c + 1];
        int larger = 0;
        for(int q = b; q < e; q++) larger += tmp_id[q] > i;
        slot = b + larger;
        recA[slot] = a;
        recV[slot] = v;
        pool_of[i] = slot;
        walks = !(__float_as_uint(a.w) & NH_PB_IDLE);
    }
    // A rank that steps a slab: the pool slots whose entity has a work item, as a LIST (any order), so that the neighbour
    // walk runs one row per listed slot instead of striding rows over a pool of which seven eighths are idle -- a row
    // then walked up to six entities one after the other and the launch took as long for an eighth of the entities as
    // for all of them (38 against 47 us).  One atomic per wave (the slab is a contiguous uid range: few waves have any).
    if(active) {
        const unsigned long long m = __ballot(walks);
        if(m) {
            const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
            int base = 0;
            if(lane == leader) base = atomicAdd(n_active, __popcll(m));
            base = __shfl(base, leader);
            if(walks) active[base + __popcll(m & ((1ull << lane) - 1ull))] = slot;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// spatial hash: query.  All 64 lanes cooperate on ONE query; returns the number written (wave
// uniform).  Visiting order == bg_*_inrange_circle (bitmap_grid.h:1408-1466): coarse 8x8 blocks
// row-major, inside a block fine rows top to bottom, cells left to right, packed elements in
// order.  Cells of one fine row are contiguous in the cell-sorted pool, so a (block,row) pair is
// one contiguous range that the lanes test 64 elements at a time; ballot + prefix popcount
// appends hits in order and enforces `maxout` exactly where the reference stops.
// The hits are POOL SLOTS (uid = recA[slot].w >> 8).
// ---------------------------------------------------------------------------------------------
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

// The whole build for a SMALL world in one workgroup: counts, scan, arrival order and placement out of LDS between
// barriers instead of five dependent launches.  The front of the step is a chain of launches of ~5 us each whatever
// they do; for a thousand entities that chain IS the front (31 us of a 97-us tick at configs[0]), and with the step's
// hand-overs at 2-3 us and the cohesion term enqueued first nothing else is in front of k_agent_mid any more.  (Round 6
// built this once before the hand-overs changed and removed it: the tick was bound by its events and the host then,
// profiles/r06_ab_small_world_hash_rejected.txt.)  Same results: the order inside a cell is fixed by the uids, not by
// who arrives first.  Leaves the global counters untouched (they stay zero for the next large build).
#define SP_SMALL_N     1024       /* entities */
#define SP_SMALL_CELLS 8192       /* cells */
#define SP_SMALL_T     256
__global__ __launch_bounds__(SP_SMALL_T) void k_sp_build_small(nh_grid G, const float *pos_xz, nh_pack_src src, int n, int ncells,
                                                            int work_begin, int work_end, int32_t *cell_start, float4 *recA, float2 *recV, int32_t *pool_of)
{
    __shared__ int32_t  start[SP_SMALL_CELLS + 1];          // counts, then the exclusive scan
    __shared__ uint16_t ecell[SP_SMALL_N], erank[SP_SMALL_N], order[SP_SMALL_N];
    __shared__ int32_t  wsum[SP_SMALL_T / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for(int c = t; c <= ncells; c += SP_SMALL_T) start[c] = 0;
    __syncthreads();
    for(int i = t; i < n; i += SP_SMALL_T) {
        const int c = sp_cell_of(G, bg_scale(pos_xz[2 * i]), bg_scale(pos_xz[2 * i + 1]));
        ecell[i] = (uint16_t)c;
        erank[i] = (uint16_t)atomicAdd(&start[c], 1);
    }
    __syncthreads();
    // exclusive scan, SP_SMALL_T cells at a time with a running carry
    int32_t carry = 0;
    for(int base = 0; base < ncells; base += SP_SMALL_T) {
        const int c = base + t;
        const int32_t v = c < ncells ? start[c] : 0;
        const int32_t incl = wave_incl_scan(v);
        if(lane == 63) wsum[w] = incl;
        __syncthreads();
        int32_t woff = 0, tot = 0;
#pragma unroll
        for(int k = 0; k < SP_SMALL_T / 64; k++) { const int32_t x = wsum[k]; if(k < w) woff += x; tot += x; }
        if(c < ncells) { start[c] = carry + woff + incl - v; cell_start[c] = carry + woff + incl - v; }
        carry += tot;
        __syncthreads();
    }
    if(t == 0) { start[ncells] = carry; cell_start[ncells] = carry; }
    __syncthreads();
    for(int i = t; i < n; i += SP_SMALL_T) order[start[ecell[i]] + erank[i]] = (uint16_t)i;
    __syncthreads();
    // descending uid inside a cell (k_sp_place)
    for(int i = t; i < n; i += SP_SMALL_T) {
        float4 a;
        float2 v;
        pool_record(i, pos_xz, src, work_begin, work_end, a, v);
        const int c = ecell[i], b = start[c], e = start[c + 1];
        int larger = 0;
        for(int q = b; q < e; q++) larger += order[q] > i;
        recA[b + larger] = a;
        recV[b + larger] = v;
        pool_of[i] = b + larger;
    }
}

// Running totals of the work units of a kernel's sub-lists, by the first wave of the workgroup: entry k
// holds `cnt[k]` agents = (cnt[k] + per - 1) / per units; unit_end[k] = units of the entries up to and
// including k.  (One thread adding up 128-256 entries in front of every workgroup's first barrier was 770-1 500
// wave instructions per workgroup: a fifth of everything k_cp_small executed.)
template <typename F>
__device__ __forceinline__ void unit_totals(int32_t *unit_end, int ntab, F units_of)
{
    if(threadIdx.x >= 64) return;
    const int lane = threadIdx.x, per_lane = (ntab + 63) >> 6;        // consecutive entries per lane
    int local = 0;
    for(int j = 0; j < per_lane; j++) {
        const int k = lane * per_lane + j;
        if(k < ntab) local += units_of(k);
    }
    int run = wave_incl_scan(local) - local;
    for(int j = 0; j < per_lane; j++) {
        const int k = lane * per_lane + j;
        if(k < ntab) { run += units_of(k); unit_end[k] = run; }
    }
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
                const float4 c = G.recA[k];
                int64_t dx = (int64_t)bg_scale(c.x) - icx, dy = (int64_t)bg_scale(c.y) - icy;
                d2 = dx * dx + dy * dy;
                hit = d2 <= ir2;
            }
            uint64_t m = __ballot(hit);
            int p = written + __popcll(m & ((1ull << lane) - 1ull));
            if(hit && p < maxout) {
                out_ids[p] = (uint32_t)k;
                if(out_d2) out_d2[p] = (int32_t)d2;
            }
            written += __popcll(m);
            if(written >= maxout) return maxout;
        }
        return written;
    }

    // A SEGMENT = the cells of one fine row inside one coarse block, contiguous in the pool.  One
    // pass resolves 64 segments in visiting order -- lane = ((coarse row, block column) << 3) | fine
    // row, with CB (a power of two) block columns and 8 / CB coarse rows per pass, so the r = 30 and
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
                    const float4 c = G.recA[k];
                    const int32_t sx = bg_scale(c.x), sy = bg_scale(c.y);
                    if(small) {
                        // (elements clamped into a border cell may be far away: range-check before
                        // squaring in 32 bits)
                        const int32_t dx = sx - icx, dy = sy - icy;
                        const bool near = (uint32_t)(dx + 32767) < 65535u && (uint32_t)(dy + 32767) < 65535u;
                        d2s = near ? dx * dx + dy * dy : 0x7fffffff;
                        hit = d2s <= (int32_t)ir2;
                    }else{
                        const int64_t dx = (int64_t)sx - icx, dy = (int64_t)sy - icy;
                        const int64_t d2 = dx * dx + dy * dy;
                        hit = d2 <= ir2;
                        d2s = (int32_t)d2;
                    }
                }
                uint64_t m = __ballot(hit);
                int p = written + __popcll(m & ((1ull << lane) - 1ull));
                if(hit && p < maxout) {
                    out_ids[p] = (uint32_t)k;
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
__device__ __forceinline__ uint32_t slot_bits(const nh_grid &G, uint32_t slot)
{
    return __float_as_uint(G.recA[slot].w);
}

__device__ int filter_garrisoned_wave(const nh_grid &G, uint32_t *ids, int count, int lane)
{
    bool any = false;
    for(int base = 0; base < count; base += 64) {
        int k = base + lane;
        any |= (k < count) && (slot_bits(G, ids[k]) & NH_PB_GARRISONED);
    }
    if(!__any(any)) return count;
    int ret = count;
    if(lane == 0) {
        for(int i = count - 1; i >= 0; i--) {
            if(slot_bits(G, ids[i]) & NH_PB_GARRISONED) {
                ids[i] = ids[ret - 1];
                ret--;
            }
        }
    }
    wave_sync();
    return __shfl(ret, 0);
}

// ---------------------------------------------------------------------------------------------
// cohesion_force (movement.c:1653)
// ---------------------------------------------------------------------------------------------
// (cohesion_t_f32 / cohesion_t_f64 -- float t = (len - 50.0f*0.75) / 50.0f of movement.c:1668 -- live in agent_math.h)

#define COH_BINS 257       /* 256 Morton blocks + 1 bin for members that take no cohesion force */
// k_coh_plan: wave_off[f] = number of 16-member (COH_APW) waves of the flocks before f (exclusive
// scan); one workgroup, chunked.  The launch uses the lane grouping the PREVIOUS regrouping left behind
// (k_coh_bin .. k_coh_scatter: perm[] + its bin prefix bin_start[]) when that was built for exactly these
// flock offsets and this work range (saved[]: offsets, then work_begin, work_end) -- flock f then gets
// ceil(members of f inside the work range / 16) waves -- and otherwise (*perm_valid = 0: first tick, flocks
// or slab changed) the identity over whole flocks: ceil(flock size / 16) waves.
__global__ __launch_bounds__(256) void k_coh_plan(const int32_t *bin_start, const int32_t *flock_offsets,
                                                  const int32_t *saved, int n_flocks, int work_begin, int work_end,
                                                  int members_key, int32_t *wave_off, int32_t *perm_valid)
{
    __shared__ int32_t wsum[4];
    __shared__ int32_t carry;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if(t == 0) carry = 0;
    bool same = saved[n_flocks + 1] == work_begin && saved[n_flocks + 2] == work_end && saved[n_flocks + 3] == members_key;
    for(int f = t; f <= n_flocks; f += 256) same = same && saved[f] == flock_offsets[f];
    const int valid = __syncthreads_and(same);
    for(int base = 0; base < n_flocks; base += 256) {
        const int f = base + t;
        int32_t v = 0;
        if(f < n_flocks) {
            const int32_t cnt = valid ? bin_start[(f + 1) * COH_BINS] - bin_start[f * COH_BINS]
                                      : flock_offsets[f + 1] - flock_offsets[f];
            v = (cnt + 15) >> 4;                                                       // COH_APW
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
    if(t == 0) { wave_off[n_flocks] = carry; *perm_valid = valid; }
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

// Members outside the work range [work_begin, work_end) -- on a rank that steps one slab of a large job
// that is most of the snapshot -- leave before the flock search and get no lane at all (bin_of = -1): the
// grouping holds the members INSIDE the range, flock by flock, active bins first.  saved[]: what the
// grouping was built for: flock offsets, the work range, and the MEMBERSHIP KEY.  A member outside the range
// has no lane, so a grouping made for a slab is only valid while flock_members is unchanged: equal offsets and
// bounds do not show that (two units swap flocks of equal size; an entity removed elsewhere shifts a uid across
// the slab boundary), and the unit that moved into the slab would keep another entity's stale force.  The key
// is 0 for a step over the whole snapshot (every member has a lane, a change is harmless), the caller's
// navhip_world.static_epoch for a slab step, and a number that never repeats when the caller gave none.
__global__ __launch_bounds__(256) void k_coh_bin(nh_step_params P, int32_t *bin_of, int32_t *bin_count,
                                                 int32_t *saved)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    for(int f = g; f <= P.n_flocks; f += gridDim.x * 256) saved[f] = P.flock_offsets[f];
    if(g == 0) { saved[P.n_flocks + 1] = P.work_begin; saved[P.n_flocks + 2] = P.work_end; saved[P.n_flocks + 3] = P.members_key; }
    if(g >= P.flock_offsets[P.n_flocks]) return;
    const int m = P.flock_members[g];
    if(m < P.work_begin || m >= P.work_end) { bin_of[g] = -1; return; }
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
    if(bin < 0) return;                                   // (outside the work range: no lane)
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
#define COH_NP  2             /* entry pairs per lane and batch: a batch is COH_G = 8 * COH_NP entries
                                 (2 measured 1.3 % faster per tick than 4: fewer VGPRs, more waves).  EVEN values only:
                                 the queue is read four entries at a time (COH_NP = 1 compiles, loads nothing and is
                                 11 % faster and wrong -- an A/B without a parity step found that out) */
static_assert(COH_NP % 2 == 0, "the cohesion queue is read four entries at a time: COH_NP must be even");
#define COH_G   (8 * COH_NP)

// an empty statement the optimiser cannot see through: keeps two scalar chains from being paired into packed math
// (paired into one v_pk_add_f32 the two ordered sums take their operands from two v_mov_b32_dpp: three instructions per
// queue entry where each add can take its DPP operand itself: 118.5 -> 107.5 us, profiles/r05_ab_compiler_flags.txt)
#ifdef NH_HOSTSIM
__device__ __forceinline__ float coh_keep(float x) { return x; }
#else
__device__ __forceinline__ float coh_keep(float x) { asm volatile("" : "+v"(x)); return x; }
#endif

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
                // (kept apart: paired into one v_pk_add_f32 the two sums take their operands from two v_mov_b32_dpp
                // -- three instructions per entry; on their own each add takes its DPP operand itself, two)
                comx = coh_keep(comx + quad_bcast(tx, sb));
                comz = coh_keep(comz + quad_bcast(tz, sb));
            }
        }
    }
}

// perm / bin_start: the lane grouping of the previous regrouping (the members inside the work range, flock f's
// at perm[bin_start[f * COH_BINS] .. bin_start[(f + 1) * COH_BINS])); *perm_valid = 0: the identity over the
// whole flock instead (k_coh_plan).
// INLINE_PLAN (at most 64 flocks: lane = flock): every wave works the launch plan out for itself -- the check
// that the grouping was built for these flock offsets and this work range, the wave prefix over the flocks --
// instead of reading what a one-workgroup kernel in front of the launch left (k_coh_plan: 6 us and a launch
// gap on the chain that gates k_agent_mid; here ~40 instructions per wave).
template <bool INLINE_PLAN>
__global__ __launch_bounds__(64) void k_cohesion(nh_step_params P, const int32_t *wave_off,
                                                 const int32_t *perm, const int32_t *perm_valid,
                                                 float *coh_xz, const int32_t *bin_start, const int32_t *saved)
{
    __shared__ double tab[64];
    // the members of the current tile that survive the box test, in member order (+ carry-over):
    // entry k lives in slot (k & 3) * COH_QS + (k >> 2)
    __shared__ __attribute__((aligned(16))) float qx[4 * COH_QS];
    __shared__ __attribute__((aligned(16))) float qz[4 * COH_QS];
    const int t = threadIdx.x, sub = t & 3;
    const int wv = blockIdx.x;
    int f = 0, first_wave = 0;
    bool use_perm;
    if(INLINE_PLAN) {
        const int nf = P.n_flocks;
        bool ok = saved[nf + 1] == P.work_begin && saved[nf + 2] == P.work_end && saved[nf + 3] == P.members_key;
        if(t <= nf) ok = ok && saved[t] == P.flock_offsets[t];
        if(t == 0 && nf == 64) ok = ok && saved[64] == P.flock_offsets[64];
        use_perm = __all(ok);
        int v = 0;
        if(t < nf) {
            const int cnt = use_perm ? bin_start[(t + 1) * COH_BINS] - bin_start[t * COH_BINS]
                                     : P.flock_offsets[t + 1] - P.flock_offsets[t];
            v = (cnt + COH_APW - 1) / COH_APW;
        }
        const int incl = wave_incl_scan(v), excl = incl - v;
        if(wv >= __shfl(incl, 63)) return;
        // the last flock whose first wave is <= wv (empty flocks share their successor's prefix)
        f = __popcll(__ballot(t < nf && excl <= wv)) - 1;
        first_wave = __shfl(excl, f);
    }else{
        if(wv >= wave_off[P.n_flocks]) return;
        // flock of this wave: binary search over the wave prefix (uniform)
        int lo = 0, hi = P.n_flocks;
        while(hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if(wave_off[mid] <= wv) lo = mid; else hi = mid;
        }
        f = lo;
        first_wave = wave_off[f];
        use_perm = *perm_valid != 0;
    }
    tab[t] = c_exp2_64[t];
    const float scaled_max_force = (float)((double)(0.75f / (float)P.hz) * 20.0);
    const int b = P.flock_offsets[f], e = P.flock_offsets[f + 1];
    const int pb = use_perm ? bin_start[f * COH_BINS] : b;
    const int pe = use_perm ? bin_start[(f + 1) * COH_BINS] : e;
    const int gp = pb + (wv - first_wave) * COH_APW + (t >> 2);
    const bool mine = gp < pe;
    // CSR entry of this quad's member (any permutation of the flock's entries serves; a grouping
    // made for other flock offsets or another work range is ignored)
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
// wave-per-agent pieces of k_agent_full
// ---------------------------------------------------------------------------------------------
// waves (= agents) per workgroup of the wave-per-agent kernels; 2 measured best for the round-1
// k_agent_step (1: 0.490, 2: 0.483, 4: 0.494, 8: 0.521 ms/tick in one session)
#define AG_WAVES 2
struct wave_lds {
    uint32_t ids30[128];                       // separation query result (cap 128, :1695)
    union {
        uint32_t ids10[512];                   // ClearPath neighbour query result (cap 512, :2779)
        float    sep[256];                     // earlier: separation terms (x, z)[128]
    } u;
    int32_t  d2_30[128];                       // squared fixed-point distances of the r=30 hits
    uint32_t ids10d[128];                      // r=10 list derived from the r=30 list
};

// separation_force, movement.c:1690.  ids30/n30 already gathered (pool slots); wave-uniform result.
__device__ v2 separation_wave(const nh_grid &G, uint32_t my_slot, v2 me, float my_radius,
                              uint32_t my_bits, const uint32_t *ids30, int n30, float *sep,
                              float scaled_max_force, int lane, const double *exp_tab)
{
    if(n30 == 0) return mkv(0.0f, 0.0f);
    for(int base = 0; base < n30; base += 64) {
        int k = base + lane;
        if(k < n30) {
            uint32_t curr = ids30[k];
            const float4 ra = G.recA[curr];                                 // {pos, radius, bits}
            uint32_t fl = __float_as_uint(ra.w);
            v2 term = mkv(0.0f, 0.0f);
            bool skip = (curr == my_slot) || !(fl & NH_PB_MOVABLE) || ((my_bits ^ fl) & NH_PB_AIR);
            if(!skip) {
                v2 t2;
                if(separation_term(me, my_radius, mkv(ra.x, ra.y), ra.z, exp_tab, t2)) term = t2;
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

// find_neighbours, movement.c:2768: classify the r=10 query result into dynamic / static lists
__device__ void classify_neighbours(const nh_grid &G, uint32_t my_slot, uint32_t my_bits,
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
            const float4 ra = G.recA[curr];
            uint32_t fl = __float_as_uint(ra.w);
            float rad = ra.z;
            bool skip = (curr == my_slot) || !(fl & NH_PB_MOVABLE) || (rad == 0.0f)
                     || ((my_bits ^ fl) & NH_PB_AIR);
            if(!skip) {
                rec[0] = ra.x; rec[1] = ra.y;
                rec[4] = rad;
                if(fl & NH_PB_STATIC) {
                    cls = 2;                       // static: velocity forced to zero (:2820)
                }else{
                    const float2 vel = G.recV[curr];
                    cls = 1;
                    rec[2] = vel.x; rec[3] = vel.y;
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

// ---------------------------------------------------------------------------------------------
// k_agent_nbr: one ROW of 16 lanes per pool slot (NBR_BLOCK / 16 entities per workgroup)
// ---------------------------------------------------------------------------------------------
// STRIDED (a rank that steps a slab of a large job; the name is history): the launch is sized by the slab and takes its
// pool slots from the list k_sp_place made of the entities with a work item.  With the whole snapshot stepped every
// pool slot has its own row.
#define NBR_WAVES 7        /* 72 VGPRs; 8 needs five spilled dwords per lane */
// Threads per workgroup of the front's kernels.  ONE wave: they run beside the cohesion kernel's stream of
// one-wave workgroups and the field builds, which take every wave slot the moment it frees up -- a workgroup
// of four waves waits until four slots of ONE compute unit are free at the same moment.
#define NBR_BLOCK 64
#define SP_BLOCK 64
template <bool STRIDED>
__global__ __launch_bounds__(NBR_BLOCK) __attribute__((amdgpu_waves_per_eu(NBR_WAVES, 8)))
void k_agent_nbr(nh_grid G, int npool_max, nh_nbr NB, float scaled_max_force)
{
    __shared__ double exp_tab[64];
    __shared__ __attribute__((aligned(16))) float2 terms[NBR_BLOCK / 16][16];
    if(threadIdx.x < 64) exp_tab[threadIdx.x] = c_exp2_64[threadIdx.x];
    __syncthreads();
    const int grp_i = threadIdx.x >> 4;
    if(!STRIDED) {
        const int k = blockIdx.x * (NBR_BLOCK / 16) + grp_i;
        if(k >= npool_max || k >= G.cell_start[G.grid_w * G.grid_h]) return;
        if(__float_as_uint(G.recA[k].w) & NH_PB_IDLE) return;         // no work item (or outside the slab)
        nbr_walk_row(G, k, scaled_max_force, exp_tab, terms[grp_i], NB);
    }else{
        // a slab: one row per pool slot k_sp_place listed (the entities with a work item)
        const int r = blockIdx.x * (NBR_BLOCK / 16) + grp_i;
        if(r >= *G.n_active) return;
        nbr_walk_row(G, G.active[r], scaled_max_force, exp_tab, terms[grp_i], NB);
    }
}

// wave-aggregated append to a device work list: one atomic per wave and list, on the wave's sub-list
__device__ __forceinline__ void worklist_push(const nh_worklists &WL, int which, bool want, int uid)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(want);
    if(!m) return;
    const int sub = (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (NH_WL_SUB - 1));
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if(lane == leader) base = atomicAdd(&WL.count[which * NH_WL_SUB + sub], __popcll(m));
    base = __shfl(base, leader);
    if(want) WL.ids[((size_t)which * NH_WL_SUB + sub) * WL.cap + base + __popcll(m & ((1ull << lane) - 1ull))] = uid;
}

// ---------------------------------------------------------------------------------------------
// k_agent_mid: the per-agent scalar chain -- desired direction, arrive force, probes, priority
// ladder, vpref -- in uid order (every per-entity input / output is contiguous), ONE thread per entity.
// The work is a chain of dependent loads and IEEE divide / sqrt sequences at 1.5 waves per SIMD; with the loads
// requested ahead of the chain (mid_thread, sample_flow) one lane per entity beats two or four
// (profiles/archive/r03_ab_mid_lanes.txt).  Splitting it -- the sampling half beside the cohesion term, the rest behind the
// join (round 5), or the rest at the head of every ClearPath search (round 6) -- was measured twice and lost twice
// (profiles/r05_ab_split_mid_*.txt, r06_ab_step_without_mid_rejected.txt): the launch is latency, and the chip is
// not idle beside it.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_agent_mid(nh_step_params P, nh_nbr NB, const float *coh_xz,
                                                  nh_mid_rec *mid, nh_worklists WL, nh_step_outs O,
                                                  float scaled_max_force, double force_thresh)
{
    const int uid = P.work_begin + (int)(blockIdx.x * 64 + threadIdx.x);
    const bool live = uid < P.work_end;
    int disp = DISP_DONE;
    if(live) {
        nh_mid_rec R;
        v2 out_vel;
        disp = mid_thread(P, uid, NB, coh_xz, scaled_max_force, force_thresh, R, out_vel);
        if(O.vdes_xz)  { O.vdes_xz[2 * uid] = R.vdes[0]; O.vdes_xz[2 * uid + 1] = R.vdes[1]; }
        if(O.vpref_xz) { O.vpref_xz[2 * uid] = R.vpref[0]; O.vpref_xz[2 * uid + 1] = R.vpref[1]; }
        if(disp == DISP_DONE) {
            post_thread(P, uid, mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]), P.state[uid], P.flags[uid],
                        P.radius[uid], out_vel, R.vel_cap, R.status, O);
        }else{
            mid[uid] = R;
        }
    }
#pragma unroll
    for(int w = 0; w <= NH_WL_FULL; w++)
        worklist_push(WL, w, live && disp == DISP_ROW0 + w, uid);
}

// ---------------------------------------------------------------------------------------------
// ClearPath for every listed agent: two launches by problem size (the per-agent cost spans three
// orders of magnitude; one kernel for all needed the registers of the largest and the LDS of each),
// side by side on two streams.
// ---------------------------------------------------------------------------------------------
#define CP_WAVES 4
// waves per workgroup of k_cp_rows (its waves work on their own: the workgroup only shares the unit tables)
#define CPR_WAVES 4
// problems on the workgroup lists from which one wave takes one problem (k_cp_heavy_solo) instead of a team
#define CP_SOLO_MIN 8192

// first k with end[k] > u (n - 1 when there is none), for a wave-uniform u, by the whole wave: the running
// totals do not decrease, so it is the number of entries <= u -- one or two ballots instead of a binary search
// of seven dependent LDS reads
__device__ __forceinline__ int first_above(const int32_t *end, int n, int u)
{
    const int lane = threadIdx.x & 63;
    int below = 0;
    for(int base = 0; base < n; base += 64) {
        const int k = base + lane;
        below += __popcll(__ballot(k < n && end[k] <= u));
    }
    return min(below, n - 1);
}

// Drawing work units, every wave on its own (no workgroup barrier: a wave that holds a long unit does
// not keep its neighbours waiting).  The units are numbered heaviest first.  All but the last one to
// two rounds are dealt out statically -- wave w takes units w, w + nw, ... : every wave gets the same
// mix of heavy and light ones, and no atomics --; the rest, the lightest units, are drawn by ticket and
// fill the gaps the uneven ones left.  Ticket unit v belongs to stripe v % NH_CP_STRIPES; a wave draws
// from its stripe's counter (one unit per ticket) and moves on to the next stripe when one is
// exhausted.  (One counter for everything was an atomic on ONE address per unit, ~5 ns each in a row,
// and two memory round trips before every unit -- more than the light units themselves cost; purely
// static dealing left the chip waiting for the unluckiest waves.)  The stripes' counters lie in
// separate 128-byte lines; a counter seen exhausted by a plain load is not drawn from again.
// Returns -1 when nothing is left.
struct unit_draw { int stripe, abandoned, round; };
__device__ __forceinline__ int next_unit(unit_draw &D, int32_t *counters, int total, int gw, int nw, int lane)
{
#define NH_CP_TICKET_ROUNDS 1
    const int static_rounds = max(0, total / nw - NH_CP_TICKET_ROUNDS);
    const int r = D.round++;
    if(r < static_rounds) return gw + r * nw;
    if(r == static_rounds) { D.stripe = gw % NH_CP_STRIPES; D.abandoned = 0; }
    const int first = static_rounds * nw, rest = total - first;       // ticket units: first .. total-1
    while(D.abandoned < NH_CP_STRIPES) {
        const int st = D.stripe;
        const int n_s = st < rest ? (rest - st + NH_CP_STRIPES - 1) / NH_CP_STRIPES : 0;
        int32_t *cnt = counters + 32 * st;
        if(__atomic_load_n(cnt, __ATOMIC_RELAXED) < n_s) {
            int t = 0;
            if(lane == 0) t = atomicAdd(cnt, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if(t < n_s) return first + t * NH_CP_STRIPES + st;
        }
        D.stripe = (st + 1) % NH_CP_STRIPES;
        D.abandoned++;
    }
    return -1;
}


// ---- k_cp_small: the lists of 1-2 and 3-4 neighbours -- three quarters of the searching agents
// outside a crowd.  One wave per unit of four agents, one attempt each (clearpath_small_row); an agent
// without any admissible candidate goes onto the retry list, which a launch of k_cp_rows works off. ----
#define CPS_WAVES 4          /* waves (units) per workgroup */
__global__ __launch_bounds__(CPS_WAVES * 64) void k_cp_small(nh_step_params P, nh_nbr NB, const nh_mid_rec *mid,
                                                            nh_worklists WL, nh_step_outs O)
{
    __shared__ __attribute__((aligned(16))) float4 cones[CPS_WAVES * 4][8];
    __shared__ int32_t unit_end[2 * NH_WL_SUB];
    __shared__ int32_t sub_cnt[2 * NH_WL_SUB];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for(int k = threadIdx.x; k < 2 * NH_WL_SUB; k += CPS_WAVES * 64)
        sub_cnt[k] = WL.count[(NH_WL_ROW1 - k / NH_WL_SUB) * NH_WL_SUB + k % NH_WL_SUB];
    __syncthreads();
    unit_totals(unit_end, 2 * NH_WL_SUB, [&](int k) { return (sub_cnt[k] + 3) >> 2; });
    __syncthreads();
    const int u = blockIdx.x * CPS_WAVES + wib;               // one unit per wave
    bool live = u < unit_end[2 * NH_WL_SUB - 1];
    int uid = 0;
    bool found = true;
    if(live) {
        const int k = first_above(unit_end, 2 * NH_WL_SUB, u), rel = u - (k ? unit_end[k - 1] : 0);
        const int list = NH_WL_ROW1 - k / NH_WL_SUB, sub = k % NH_WL_SUB;
        const int idx = rel * 4 + (lane >> 4);
        live = idx < sub_cnt[k];                                // (else: a row beyond the end of its sub-list)
        if(live) {
            uid = WL.ids[((size_t)list * NH_WL_SUB + sub) * WL.cap + idx];
            const nh_mid_rec R = mid[uid];
            const uint32_t c = NB.cnt[uid];
            const int n_dyn = (int)(c & 0xff), n_stat = (int)((c >> 8) & 0xff);
            cpent ent;
            ent.pos = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
            ent.vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
            ent.radius = P.radius[uid];
            const int gl = lane & 15;
            const bool have = gl < n_dyn + n_stat, isdyn = gl < n_dyn;
            cpent nb; nb.pos = mkv(0, 0); nb.vel = mkv(0, 0); nb.radius = 0;
            if(have) nb = nbr_load(NB, uid, isdyn ? gl : 32 + gl - n_dyn);
            const v2 nv = clearpath_small_row(ent, mkv(R.vpref[0], R.vpref[1]), nb, isdyn, have,
                                              cones[wib * 4 + (lane >> 4)], found);
            if(found && gl == 0)
                post_thread(P, uid, ent.pos, P.state[uid], P.flags[uid], ent.radius, nv, R.vel_cap, R.status, O);
        }
    }
    worklist_push(WL, NH_WL_RETRY, live && !found && (lane & 15) == 0, uid);
}

// ---- k_cp_rows: the four row lists (1-16 neighbours), a row of 16 lanes per agent, four agents per
// unit, the units numbered heaviest list first (9-16, 5-8, 3-4, 1-2 neighbours) ---------------------
// (the lists list0, list0 - 1, ... : nlists of them, at most four; ticket_set: which set of stripe
// counters -- the retry launch runs beside the main one)
#define CP_ROWS_OCC 4           /* (pinned like k_cp_heavy: four waves per SIMD, 128 registers -- see the note on hole inheritance below) */
__attribute__((amdgpu_waves_per_eu(CP_ROWS_OCC, CP_ROWS_OCC)))
__global__ __launch_bounds__(CPR_WAVES * 64) void k_cp_rows(nh_step_params P, nh_nbr NB, const nh_mid_rec *mid,
                                                           nh_worklists WL, nh_step_outs O, int list0, int nlists,
                                                           int ticket_set, nh_signal lists_ready)
{
    // (this launch follows k_agent_mid on its stream: that it runs says the work lists are complete)
    if(lists_ready.flag && blockIdx.x == 0 && threadIdx.x == 0) {
#ifdef NH_HOSTSIM
        *lists_ready.flag = lists_ready.seq;
#else
        __hip_atomic_store(lists_ready.flag, lists_ready.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    __shared__ cp_lds<16> lds[CPR_WAVES * 4];
    // unit_end[k] = units of the sub-lists up to and including k (k = order * NH_WL_SUB + sub)
    __shared__ int32_t unit_end[4 * NH_WL_SUB];
    __shared__ int32_t sub_cnt[4 * NH_WL_SUB];
    // HOLE INHERITANCE (profiles/HISTORY.md 3.7).  This launch reaches the device a few microseconds before k_cp_heavy (which waits
    // for an event of the other stream) and fills every SIMD; k_cp_heavy's persistent workgroups move into the
    // register and LDS ranges its workgroups leave behind and keep them for the whole launch.  A hole smaller than a
    // k_cp_heavy wave (128 registers) or workgroup is lost to it: 119 instead of 121 registers here (120 instead of
    // 128 allocated), or 37.5 KB of LDS there against 36 here, cost the crowded world a quarter of k_cp_heavy's
    // waves (4.9 -> 6.1 ms per tick, profiles/archive/r04_ab_hole_inheritance.txt).  So: this kernel allocates the same 128
    // registers per lane (v127 named as clobbered), and its workgroup owns at least k_cp_heavy's LDS.
    asm volatile("" ::: "v127");
    static_assert(sizeof(cp_lds<16>) * CPR_WAVES * 4 + 8 * 4 * NH_WL_SUB >= sizeof(cp_lds<64>) * CP_WAVES + 8 * NH_WL_SUB + 4 + sizeof(cp_team),
                  "k_cp_rows' workgroup must own at least k_cp_heavy's LDS (hole inheritance)");
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ntab = nlists * NH_WL_SUB;
    // (usually nothing to do for the retry launch: one parallel look)
    if(nlists == 1 && !__any(WL.count[list0 * NH_WL_SUB + lane] != 0)) return;
    for(int k = threadIdx.x; k < ntab; k += CPR_WAVES * 64)
        sub_cnt[k] = WL.count[(list0 - k / NH_WL_SUB) * NH_WL_SUB + k % NH_WL_SUB];
    __syncthreads();
    unit_totals(unit_end, ntab, [&](int k) { return (sub_cnt[k] + 3) >> 2; });
    __syncthreads();
    const int total = unit_end[ntab - 1];
    int32_t *counters = WL.count + NH_WL_LISTS * NH_WL_SUB + 32 * (1 + ticket_set * NH_CP_STRIPES);
    const int gw = blockIdx.x * CPR_WAVES + wib, nw = gridDim.x * CPR_WAVES;
    cp_lds<16> &S = lds[wib * 4 + (lane >> 4)];
    unit_draw D; D.round = 0;
    for(;;) {
        const int u = next_unit(D, counters, total, gw, nw, lane);
        if(u < 0) break;
        const int k = first_above(unit_end, ntab, u), rel = u - (k ? unit_end[k - 1] : 0);
        const int list = list0 - k / NH_WL_SUB, sub = k % NH_WL_SUB;
        const int idx = rel * 4 + (lane >> 4);
        if(idx < sub_cnt[k]) {                  // (else: a row beyond the end of its sub-list)
            const int uid = WL.ids[((size_t)list * NH_WL_SUB + sub) * WL.cap + idx];
            const nh_mid_rec R = mid[uid];
            const uint32_t c = NB.cnt[uid];
            const int n_dyn = (int)(c & 0xff), n_stat = (int)((c >> 8) & 0xff);
            cpent ent;
            ent.pos = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
            ent.vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
            ent.radius = P.radius[uid];
            cp_load_lists<16>(P.grid, NB, uid, n_dyn, n_stat, S);
            const v2 nv = clearpath_grp<16>(ent, mkv(R.vpref[0], R.vpref[1]), n_dyn, n_stat, S);
            if((lane & 15) == 0)
                post_thread(P, uid, ent.pos, P.state[uid], P.flags[uid], ent.radius, nv, R.vel_cap, R.status, O);
        }
    }
}

// ---- k_cp_heavy: the heavy list (33-64 neighbours), then the wave list (17-32).  A search of up to 16 000 ray
// pairs on one wave takes hundreds of microseconds -- as long as everything else of the tick --, and even among
// problems of 20-30 neighbours the cost spreads over a factor of ten (how early an admissible candidate turns up
// decides how much is pruned).  So: one problem per WORKGROUP, its waves search it as a team (clearpath_grp<64, true>)
// -- the launch is bound by its LONGEST problems, and a team quarters every problem's latency.  In a jam, from
// CP_SOLO_MIN problems on (92 000 in the crowded world), there are more problems than waves, the load balances over
// problems, and a team would only repeat the cone / rank construction four times: every WAVE takes a problem of its
// own, its first one by wave number, then a ticket per wave and problem.  (A two-pass schedule -- every wave on its
// own, searches that find no bound handed over to teams -- was measured and lost: profiles/archive/r04_ab_cp_bail_100.txt.)
#define CP_HEAVY_OCC 4          /* (pinned: a few registers above 128 would silently cost a wave per SIMD) */
__attribute__((amdgpu_waves_per_eu(CP_HEAVY_OCC, CP_HEAVY_OCC)))
__global__ __launch_bounds__(CP_WAVES * 64) void k_cp_heavy(nh_step_params P, nh_nbr NB, const nh_mid_rec *mid,
                                                            nh_worklists WL, nh_step_outs O, int32_t *zero_next)
{
    __shared__ cp_lds<64> lds[CP_WAVES];
    __shared__ int32_t hv_end[2 * NH_WL_SUB];       // sub-lists of the heavy list, then of the wave list
    __shared__ int32_t h_ticket;
    __shared__ cp_team team;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // The last launch of the step on the side stream clears the OTHER set of list counters for the next step.
    // On THIS stream because the library copies every step's counters to pinned host memory behind the step, on
    // this stream too (navhip_step_lists_peek): the clearing of a set is ordered behind the copy of that set by the
    // stream itself.  (Its users -- the previous step -- finished before this step's k_agent_mid started.)
    if(blockIdx.x == 0 && zero_next)
        for(int i = threadIdx.x; i < (int)NH_WL_COUNTERS; i += CP_WAVES * 64) zero_next[i] = 0;
    cp_lds<64> &S = lds[wib];
    // (outside a crowd there is nothing to do: one parallel look at the 128 counters)
    if(!__any((WL.count[NH_WL_HEAVY * NH_WL_SUB + lane] | WL.count[NH_WL_WAVE * NH_WL_SUB + lane]) != 0)) return;
    unit_totals(hv_end, 2 * NH_WL_SUB, [&](int k) {
        return WL.count[(k < NH_WL_SUB ? NH_WL_HEAVY : NH_WL_WAVE) * NH_WL_SUB + k % NH_WL_SUB]; });
    __syncthreads();
    const int n_heavy = hv_end[2 * NH_WL_SUB - 1];
    int32_t *ticket = WL.count + NH_WL_LISTS * NH_WL_SUB;
    if(n_heavy >= CP_SOLO_MIN) {
        // ---- a jam: one problem per wave
        const int nw = (int)gridDim.x * CP_WAVES;
        for(int round = 0; ; round++) {
            int t = (int)blockIdx.x * CP_WAVES + wib;
            if(round > 0) {
                int v = 0x7fffffff;
                if(lane == 0 && nw + __atomic_load_n(ticket, __ATOMIC_RELAXED) < n_heavy) v = nw + atomicAdd(ticket, 1);
                t = __shfl(v, 0);
            }
            if(t >= n_heavy) break;
            const int k = first_above(hv_end, 2 * NH_WL_SUB, t), idx = t - (k ? hv_end[k - 1] : 0);
            const int uid = WL.ids[((size_t)(k < NH_WL_SUB ? NH_WL_HEAVY : NH_WL_WAVE) * NH_WL_SUB + k % NH_WL_SUB) * WL.cap + idx];
            const nh_mid_rec R = mid[uid];
            const uint32_t c = NB.cnt[uid];
            const int n_dyn = (int)(c & 0xff), n_stat = (int)((c >> 8) & 0xff);
            cpent ent;
            ent.pos = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
            ent.vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
            ent.radius = P.radius[uid];
            cp_load_lists<64>(P.grid, NB, uid, n_dyn, n_stat, S);
            const v2 nv = clearpath_grp<64>(ent, mkv(R.vpref[0], R.vpref[1]), n_dyn, n_stat, S);
            if(lane == 0)
                post_thread(P, uid, ent.pos, P.state[uid], P.flags[uid], ent.radius, nv, R.vel_cap, R.status, O);
        }
        return;
    }
    // ---- one problem per workgroup
    for(int round = 0; ; round++) {
        int t = blockIdx.x;
        if(round > 0) {
            __syncthreads();
            if(threadIdx.x == 0) {
                int v = 0x7fffffff;
                if((int)gridDim.x + __atomic_load_n(ticket, __ATOMIC_RELAXED) < n_heavy)
                    v = (int)gridDim.x + atomicAdd(ticket, 1);
                h_ticket = v;
            }
            __syncthreads();
            t = h_ticket;
        }
        if(t >= n_heavy) break;
        const int k = first_above(hv_end, 2 * NH_WL_SUB, t), idx = t - (k ? hv_end[k - 1] : 0);
        const int uid = WL.ids[((size_t)(k < NH_WL_SUB ? NH_WL_HEAVY : NH_WL_WAVE) * NH_WL_SUB + k % NH_WL_SUB) * WL.cap + idx];
        const nh_mid_rec R = mid[uid];
        const uint32_t c = NB.cnt[uid];
        const int n_dyn = (int)(c & 0xff), n_stat = (int)((c >> 8) & 0xff);
        cpent ent;
        ent.pos = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
        ent.vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
        ent.radius = P.radius[uid];
        const v2 vpref = mkv(R.vpref[0], R.vpref[1]);
        cp_load_lists<64>(P.grid, NB, uid, n_dyn, n_stat, S);
        const v2 nv = clearpath_grp<64, true>(ent, vpref, n_dyn, n_stat, S, wib, CP_WAVES, &team);
        if(wib == 0 && lane == 0)
            post_thread(P, uid, ent.pos, P.state[uid], P.flags[uid], ent.radius, nv, R.vel_cap, R.status, O);
    }
}

// ---------------------------------------------------------------------------------------------
// k_agent_full: one WAVE per listed agent, the whole neighbour-dependent part on the wave: r = 30
// query + garrison filter + separation, the priority ladder, r = 10 neighbours, ClearPath.
// (The exact path for what the row walk declines.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AG_WAVES * 64) void k_agent_full(nh_step_params P, const float *coh_xz,
                                                              const nh_mid_rec *mid, nh_worklists WL,
                                                              nh_step_outs O, float scaled_max_force,
                                                              double force_thresh, int32_t *zero_next)
{
    __shared__ wave_lds lds[AG_WAVES];
    __shared__ cp_lds<64> cps[AG_WAVES];
    __shared__ double exp_tab[64];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (zero_next: the other set of list counters, when this launch is the one that clears it -- k_cp_heavy is)
    if(blockIdx.x == 0 && zero_next)
        for(int i = threadIdx.x; i < (int)NH_WL_COUNTERS; i += AG_WAVES * 64) zero_next[i] = 0;
    // usually there is nothing to do: one parallel look at the 64 sub-list counters
    if(!__any(WL.count[NH_WL_FULL * NH_WL_SUB + lane] != 0)) return;
    if(threadIdx.x < 64) exp_tab[threadIdx.x] = c_exp2_64[threadIdx.x];
    __syncthreads();
    wave_lds &W = lds[wib];
    cp_lds<64> &S = cps[wib];
    const nh_grid &G = P.grid;
    for(int sub = 0; sub < NH_WL_SUB; sub++) {
    const int count = WL.count[NH_WL_FULL * NH_WL_SUB + sub];
    for(int idx = blockIdx.x * AG_WAVES + wib; idx < count; idx += gridDim.x * AG_WAVES) {
        const int uid = WL.ids[((size_t)NH_WL_FULL * NH_WL_SUB + sub) * WL.cap + idx];
        const nh_mid_rec R = mid[uid];
        const uint32_t my_slot = (uint32_t)G.pool_of[uid];
        const uint32_t my_bits = slot_bits(G, my_slot);
        const v2 me = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
        const v2 vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
        const float my_radius = P.radius[uid];
        int n30raw = -1;                 // size of the unfiltered r=30 list (-1: no such query)
        v2 vpref = mkv(0.0f, 0.0f);
        wave_sync();
        if(R.mode != AM_ZERO_VPREF) {
            // separation (movement.c:1690): r = 30 query, cap 128
            int n30 = sp_query_wave(G, me.x, me.z, 30.0f, 128, W.ids30, lane, W.d2_30);
            wave_sync();
            n30raw = n30;
            const int n10d = derive_r10(G, me, W.ids30, W.d2_30, n30raw, W.ids10d, lane);
            n30 = filter_garrisoned_wave(G, W.ids30, n30, lane);
            if(n10d < 0) n30raw = -1; else n30raw = n10d;
            const v2 separation = separation_wave(G, my_slot, me, my_radius, my_bits, W.ids30, n30,
                                                  W.u.sep, scaled_max_force, lane, exp_tab);
            vpref = vpref_from_forces(P, uid, R.mode, me, vel, P.flock[uid], mkv(R.arrive[0], R.arrive[1]),
                                      separation, R.probes, coh_xz, scaled_max_force, force_thresh);
        }
        // find_neighbours :2768: r = 10 query, cap 512 -- taken from the r = 30 list when that list
        // is complete (n30raw now holds the derived count, -1 = not derivable)
        uint32_t *ids10 = W.u.ids10;
        int n10;
        if(n30raw >= 0) {
            ids10 = W.ids10d;
            n10 = n30raw;
        }else{
            n10 = sp_query_wave(G, me.x, me.z, 10.0f, 512, W.u.ids10, lane);
            wave_sync();
        }
        n10 = filter_garrisoned_wave(G, ids10, n10, lane);
        int n_dyn, n_stat;
        classify_neighbours(G, my_slot, my_bits, ids10, n10, S.dyn, n_dyn, S.stat, n_stat, lane);
        cpent ent; ent.pos = me; ent.vel = vel; ent.radius = my_radius;
        const v2 nv = clearpath_grp<64>(ent, vpref, n_dyn, n_stat, S);
        if(lane == 0) {
            if(O.vpref_xz) { O.vpref_xz[2 * uid] = vpref.x; O.vpref_xz[2 * uid + 1] = vpref.z; }
            post_thread(P, uid, me, P.state[uid], P.flags[uid], my_radius, nv, R.vel_cap, R.status, O);
        }
    }
    }
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
    uint32_t *mine = out_ids + (size_t)q * maxout;
    int n = sp_query_wave(G, query_xz[2 * q], query_xz[2 * q + 1], range, maxout, mine, lane);
    wave_sync();
    for(int k = lane; k < n; k += 64) mine[k] = slot_bits(G, mine[k]) >> NH_PB_UID_SHIFT;   // slot -> uid
    if(lane == 0) out_counts[q] = n;
}

// ---------------------------------------------------------------------------------------------
// the arrival arm of entity_compute_update (movement.c:2303; see include/navhip.h): a row of 16 lanes
// per unit -- the scalar tests on every lane, the flock-mate scan (:953) shared by the lanes
// ---------------------------------------------------------------------------------------------
// the ARRIVED members of every flock, compacted to the front of the flock's range of the member list (any order)
__global__ __launch_bounds__(256) void k_arrived_compact(nh_step_params P, float4 *arrived, int32_t *arrived_n)
{
    __shared__ int count;
    const int f = blockIdx.x;
    if(threadIdx.x == 0) count = 0;
    __syncthreads();
    const int b = P.flock_offsets[f], e = P.flock_offsets[f + 1];
    for(int k0 = b; k0 < e; k0 += 256) {
        const int k = k0 + (int)threadIdx.x;
        int m = -1;
        if(k < e) { m = P.flock_members[k]; if(P.state[m] != NAVHIP_STATE_ARRIVED) m = -1; }
        const uint64_t bal = __ballot(m >= 0);
        int base = 0;
        if((threadIdx.x & 63) == 0 && bal) base = atomicAdd(&count, __popcll(bal));
        base = __shfl(base, 0);
        if(m >= 0) {
            const int at = b + base + __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull));
            // (the scratch holds n_ents rows -- an entity belongs to at most one flock; a device-side member list that
            // breaks that promise loses rows of the scan instead of writing past the buffer)
            if(at < P.n_ents) arrived[at] = make_float4(P.pos_xz[2 * m], P.pos_xz[2 * m + 1], P.radius[m], __int_as_float(m));
        }
    }
    __syncthreads();
    if(threadIdx.x == 0) arrived_n[f] = count;
}

__global__ __launch_bounds__(256) void k_state_update(nh_step_params P, navhip_state_in in, const float4 *arrived,
                                                      const int32_t *arrived_n, uint8_t *out_state, uint8_t *out_flags)
{
    typedef grp<16> g;
    const int uid = P.work_begin + ((blockIdx.x * 256 + threadIdx.x) >> 4);
    const int gl = g::lane();
    if(uid >= P.work_end) return;
    const int state = P.state[uid];
    uint8_t flags = 0, next = (uint8_t)state;
    const int flock = P.flock[uid];
    const uint32_t eflags = P.flags[uid];
    const float radius = P.radius[uid];
    const int layer = nav_layer_for(eflags, radius);
    bool decided = false;
    if(eflags & NAVHIP_ENTITY_FLAG_GARRISONED) {                           // :2344-2351
        if(!state_is_still(state)) { flags = NAVHIP_SU_SET_STATE; next = NAVHIP_STATE_ARRIVED; }
        decided = true;
    }else if(state == NAVHIP_STATE_SEEK_ENEMIES || state == NAVHIP_STATE_ARRIVED) {
        decided = true;                                                    // :2521-2528, :2643: no transition
    }else if((state != NAVHIP_STATE_MOVING && state != NAVHIP_STATE_MOVING_IN_FORMATION) || flock < 0
          || (in.skip && in.skip[uid]) || !P.map.layers[layer].cost || in.flock_layer[flock] != layer) {
        flags = NAVHIP_SU_HOST;
        decided = true;
    }
    if(!decided) {
        const v2 np = mkv(in.new_pos_xz[2 * uid], in.new_pos_xz[2 * uid + 1]);
        const v2 target = mkv(P.flock_target_xz[2 * flock], P.flock_target_xz[2 * flock + 1]);
        if(pos_pathable(P, layer, np.x, np.z)) {                           // :2437
            // ---- arrived(uid, new_pos), :2170
            const float thresh = radius * 1.5f;
            bool arr = vlen(vsub(target, np)) < thresh;
            if(!arr) {
                // N_IsAdjacentToImpassable, nav.c:4745: a 4-neighbour tile that n_tile_blocked (:235)
                tiledesc t;
                bool adj = false;
                if(tile_for_point(P, np.x, np.z, t)) {
                    const int ar = t.chunk_r * 64 + t.tile_r, ac = t.chunk_c * 64 + t.tile_c;
                    const int dr[4] = {-1, 0, 0, 1}, dc[4] = {0, -1, 1, 0};
#pragma unroll
                    for(int k = 0; k < 4; k++) {
                        const int r = ar + dr[k], c = ac + dc[k];
                        if(r < 0 || c < 0 || r >= P.map.h * 64 || c >= P.map.w * 64) continue;      // M_Tile_RelativeDesc
                        tiledesc a;
                        a.chunk_r = r >> 6; a.chunk_c = c >> 6; a.tile_r = r & 63; a.tile_c = c & 63;
                        adj = adj || tile_probe(P, layer, a) != 1u;          // impassable or blocked
                    }
                }
                if(adj) {
                    // N_IsMaximallyClose, nav.c:4727-4740: any of the destination's closest island tiles
                    // within the threshold (centre as the reference computes it: map_pos -/+ tile * 4)
                    bool close = false;
                    for(int k = in.flock_tiles_off[flock] + gl; k < in.flock_tiles_off[flock + 1]; k += 16) {
                        const float cx = P.map_x - (float)in.flock_tiles[2 * k + 1] * 4.0f;
                        const float cz = P.map_z + (float)in.flock_tiles[2 * k] * 4.0f;
                        close = close || vlen(vsub(mkv(cx, cz), np)) <= thresh;
                    }
                    arr = g::any(close);
                }
            }
            if(!arr) {
                const v2 nearest = mkv(in.flock_nearest_xz[2 * flock], in.flock_nearest_xz[2 * flock + 1]);
                if(nearest.x == nearest.x) arr = vlen(vsub(nearest, np)) < thresh;                   // :2187-2192
            }
            if(!arr) {
                // ---- a flock mate that touches us has arrived, :2480-2497 (positions and states of the
                // snapshot: adjacent_flock_members reads the tick's tables).  An existence test -- order does not
                // matter --, so the row scans the ARRIVED members only: k_arrived_compact has put {x, z, radius, uid}
                // of those, flock by flock, where the flock's member list starts (a fresh world: none; the scan of
                // every member cost 405 us per 100 000 units, nine tenths of the state pass)
                const v2 me = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
                bool hit = false;
                const int b = P.flock_offsets[flock], e = min(b + arrived_n[flock], P.n_ents);
                for(int k0 = b; k0 < e && !g::any(hit); k0 += 16) {
                    const int k = k0 + gl;
                    if(k < e) {
                        const float4 a = arrived[k];
                        if(__float_as_int(a.w) != uid)
                            hit = vlen(vsub(me, mkv(a.x, a.y))) <= radius + a.z + 5.0f;                  // ADJACENCY_SEP_DIST
                    }
                }
                arr = g::any(hit);
            }
            if(arr) {
                flags = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK; next = NAVHIP_STATE_ARRIVED;
            }else{
                const v2 vdes = mkv(in.vdes_xz[2 * uid], in.vdes_xz[2 * uid + 1]);
                if(vlen(vdes) < 1.0f / 1024.0f) {                          // :2508
                    flags = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK; next = NAVHIP_STATE_WAITING;
                }
            }
        }
    }
    if(gl == 0) { out_state[uid] = next; out_flags[uid] = flags; }
}

void nh_launch_state_update(const nh_step_params &P, const navhip_state_in &in, float4 *d_arrived, int32_t *d_arrived_n,
                            uint8_t *d_state, uint8_t *d_flags, hipStream_t s)
{
    const int n = P.work_end - P.work_begin;
    if(n <= 0) return;
    if(P.n_flocks > 0)
        hipLaunchKernelGGL(k_arrived_compact, dim3(P.n_flocks), dim3(256), 0, s, P, d_arrived, d_arrived_n);
    hipLaunchKernelGGL(k_state_update, dim3((n + 15) / 16), dim3(256), 0, s, P, in, (const float4*)d_arrived,
                       (const int32_t*)d_arrived_n, d_state, d_flags);
}

// N_DesiredGroupArrivalVelocity (nav.c:3561): direction under each point in the chunk field of its mapping
// row + "the tile is a sink inside the zone's disc" (:3596-3600)
__global__ __launch_bounds__(256) void k_region_lookup(nh_step_params P, int nq, const float *pos, const int32_t *rows,
                                                       const int32_t *centre_abs, const int32_t *radius,
                                                       uint8_t *out_dir, uint8_t *out_at_slot)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if(q >= nq) return;
    uint8_t dir = 0xff, at = 0;
    tiledesc t;
    const int row = rows[q];
    if(row >= 0 && tile_for_point(P, pos[2 * q], pos[2 * q + 1], t)) {
        const int slot = P.region_field_slot[(size_t)row * (P.map.w * P.map.h) + t.chunk_r * P.map.w + t.chunk_c];
        if(slot >= 0) {
            dir = P.field_pool[((size_t)slot << 12) + t.tile_r * 64 + t.tile_c] & 0xf;
            if(dir == NAVHIP_FD_NONE && centre_abs) {
                // M_Tile_Distance(res, &centre_tile, &tile, &dr, &dc), tile.c:414
                const int dr = (t.chunk_r * 64 + t.tile_r) - centre_abs[2 * q];
                const int dc = (t.chunk_c * 64 + t.tile_c) - centre_abs[2 * q + 1];
                at = (dr * dr + dc * dc) <= radius[q] * radius[q];
            }
        }
    }
    out_dir[q] = dir;
    if(out_at_slot) out_at_slot[q] = at;
}

void nh_launch_region_lookup(const nh_step_params &P, int nq, const float *d_pos, const int32_t *d_rows,
                             const int32_t *d_centre_abs, const int32_t *d_radius, uint8_t *d_dir, uint8_t *d_at_slot,
                             hipStream_t s)
{
    if(nq > 0)
        hipLaunchKernelGGL(k_region_lookup, dim3((nq + 255) / 256), dim3(256), 0, s, P, nq, d_pos, d_rows, d_centre_abs,
                           d_radius, d_dir, d_at_slot);
}

// G_ClearPath_NewVelocity for nq independent problems on groups of GW lanes (GW = 64: any problem;
// GW = 16: n_dyn + n_stat <= 16 -- the row path of the agent step)
template <int GW>
__global__ __launch_bounds__(128) void k_clearpath(int nq, const float *ent, const float *des_v,
                                                   const float *dyn, const int32_t *n_dyn,
                                                   const float *stat, const int32_t *n_stat,
                                                   float *out)
{
    __shared__ cp_lds<GW> lds[128 / GW];
    const int gi = threadIdx.x / GW, gl = threadIdx.x & (GW - 1);
    const int q = uni<GW>((int)(blockIdx.x * (128 / GW) + gi));           // (a wave-wide group: the same in every lane)
    if(q >= nq) return;
    cp_lds<GW> &S = lds[gi];
    const int nd = n_dyn[q], ns = n_stat[q];
    for(int i = gl; i < nd * 5; i += GW) S.dyn[i] = dyn[(size_t)q * 160 + i];
    for(int i = gl; i < ns * 5; i += GW) S.stat[i] = stat[(size_t)q * 160 + i];
    wave_sync();
    cpent e; e.pos = mkv(ent[5 * q], ent[5 * q + 1]); e.vel = mkv(ent[5 * q + 2], ent[5 * q + 3]);
    e.radius = ent[5 * q + 4];
    const v2 r = clearpath_grp<GW>(e, mkv(des_v[2 * q], des_v[2 * q + 1]), nd, ns, S);
    if(gl == 0) { out[2 * q] = r.x; out[2 * q + 1] = r.z; }
}

// the same for a team of CP_WAVES waves per problem (what k_cp_heavy runs)
__global__ __launch_bounds__(CP_WAVES * 64) void k_clearpath_team(int nq, const float *ent, const float *des_v,
                                                                  const float *dyn, const int32_t *n_dyn,
                                                                  const float *stat, const int32_t *n_stat,
                                                                  float *out)
{
    __shared__ cp_lds<64> lds[CP_WAVES];
    __shared__ cp_team team;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x;
    cp_lds<64> &S = lds[wib];
    const int nd = n_dyn[q], ns = n_stat[q];
    for(int i = lane; i < nd * 5; i += 64) S.dyn[i] = dyn[(size_t)q * 160 + i];
    for(int i = lane; i < ns * 5; i += 64) S.stat[i] = stat[(size_t)q * 160 + i];
    wave_sync();
    cpent e; e.pos = mkv(ent[5 * q], ent[5 * q + 1]); e.vel = mkv(ent[5 * q + 2], ent[5 * q + 3]);
    e.radius = ent[5 * q + 4];
    const v2 r = clearpath_grp<64, true>(e, mkv(des_v[2 * q], des_v[2 * q + 1]), nd, ns, S, wib, CP_WAVES, &team);
    if(threadIdx.x == 0) { out[2 * q] = r.x; out[2 * q + 1] = r.z; }
}

// ClearPath retry statistics (developer diagnostics: scripts/, bench.py --cp-stats)
extern "C" int navhip_debug_cp_attempts(unsigned long long out[9], int reset)
{
    if(hipDeviceSynchronize() != hipSuccess) return 1;
    if(hipMemcpyFromSymbol(out, HIP_SYMBOL(nh_cp_attempts), 9 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if(reset) {
        unsigned long long z[9] = {0};
        if(hipMemcpyToSymbol(HIP_SYMBOL(nh_cp_attempts), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}



// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// Four dependent launches, no memset (cell_count is zeroed by k_sp_scan_add once it has been
// consumed; the box of the slab filter is the exception).
void nh_launch_spatial_build(nh_grid &G, const float *d_pos_xz, nh_spatial_scratch &S,
                             int slab_begin, int slab_end, hipStream_t s)
{
    const int n = G.n, ncells = G.grid_w * G.grid_h;
    // a strict sub-range of the entities is stepped: hash only what its queries can reach
    const int32_t *box = nullptr;
    int32_t *box_next = nullptr;
    if(S.box && (slab_begin > 0 || slab_end < n)) {
        int32_t *mine = S.box + 4 * (S.box_parity & 1);
        box_next = S.box + 4 * ((S.box_parity & 1) ^ 1);
        if(slab_end > slab_begin)
            hipLaunchKernelGGL(k_sp_bbox, dim3(min(128, (slab_end - slab_begin + 255) / 256)), dim3(256), 0, s,
                               d_pos_xz, slab_begin, slab_end, mine);
        box = mine;
    }
    G.cell_start = S.cell_start; G.recA = S.recA; G.recV = S.recV; G.pool_of = S.pool_of;
    // (a slab: the list of pool slots with a work item lives in ent_rank's buffer, which is free once k_sp_scatter has
    // read it; its length behind the two slab boxes)
    int32_t *active = box ? S.ent_rank : nullptr, *n_active = box ? S.box + 8 : nullptr;
    G.active = active; G.n_active = n_active;
    if(!box && n > 0 && n <= SP_SMALL_N && ncells <= SP_SMALL_CELLS) {
        // a small world, all of it stepped: one workgroup instead of five launches
        hipLaunchKernelGGL(k_sp_build_small, dim3(1), dim3(SP_SMALL_T), 0, s, G, d_pos_xz, S.src, n, ncells, slab_begin, slab_end, S.cell_start, S.recA,
                           S.recV, S.pool_of);
        return;
    }
    if(n > 0)
        hipLaunchKernelGGL(k_sp_count, dim3((n + 255) / 256), dim3(256), 0, s, G, d_pos_xz, n,
                           S.ent_cell, S.ent_rank, S.cell_count, box, box_next, n_active);
    const int nblocks = (ncells + NH_SCAN_T - 1) / NH_SCAN_T;
    hipLaunchKernelGGL(k_sp_scan_local, dim3(nblocks), dim3(NH_SCAN_T), 0, s, S.cell_count, S.cell_start,
                       S.block_sum, ncells, G, box);
    hipLaunchKernelGGL(k_sp_scan_add, dim3(nblocks), dim3(NH_SCAN_T), 0, s, S.cell_start, S.block_sum, ncells,
                       nblocks, S.cell_count, G, box);
    if(n > 0) {
        hipLaunchKernelGGL(k_sp_scatter, dim3((n + 255) / 256), dim3(256), 0, s, S.ent_cell, S.ent_rank, n,
                           S.cell_start, S.tmp_id);
        hipLaunchKernelGGL(k_sp_place, dim3((n + SP_BLOCK - 1) / SP_BLOCK), dim3(SP_BLOCK), 0, s, G, d_pos_xz, S.src,
                           S.ent_cell, S.tmp_id, n, slab_begin, slab_end, S.recA, S.recV, S.pool_of, active, n_active);
    }
}

void nh_launch_agent_nbr(const nh_step_params &P, const nh_nbr &NB, hipStream_t s)
{
    if(P.n_ents > 0 && P.work_end > P.work_begin) {
        const float smf = (float)((double)(0.75f / (float)P.hz) * 20.0);
        const int slab = P.work_end - P.work_begin;
        if(slab == P.n_ents) {
            hipLaunchKernelGGL(k_agent_nbr<false>, dim3((P.n_ents + NBR_BLOCK / 16 - 1) / (NBR_BLOCK / 16)), dim3(NBR_BLOCK), 0, s,
                               P.grid, P.n_ents, NB, smf);
        }else{
            // a row per entity of the slab (the list of k_sp_place holds at most that many)
            const int rows = slab;
            hipLaunchKernelGGL(k_agent_nbr<true>, dim3((rows + NBR_BLOCK / 16 - 1) / (NBR_BLOCK / 16)), dim3(NBR_BLOCK), 0, s,
                               P.grid, P.n_ents, NB, smf);
        }
    }
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
    C.nb = n_flocks * COH_BINS; C.nblocks = (C.nb + NH_SCAN_T - 1) / NH_SCAN_T;
    C.wave_off = scratch;
    C.bin_count = C.wave_off + n_flocks + 1;
    C.bin_fill = C.bin_count + C.nb;
    C.bin_start = C.bin_fill + C.nb;                      // [nb + 1]
    C.block_sum = C.bin_start + C.nb + 1;
    C.bin_of = C.block_sum + C.nblocks;
    C.perm[0] = C.bin_of + n_members;
    C.perm[1] = C.perm[0] + n_members;
    C.saved[0] = C.perm[1] + n_members;
    C.saved[1] = C.saved[0] + n_flocks + 4;
    C.valid = C.saved[1] + n_flocks + 4;
    return C;
}
size_t nh_cohesion_scratch_bytes(int n_flocks, int n_members)
{
    const size_t nb = (size_t)n_flocks * COH_BINS;
    return sizeof(int32_t) * (3 * ((size_t)n_flocks + 4) + 3 * nb + 1 + (nb + NH_SCAN_T - 1) / NH_SCAN_T
                              + 3 * (size_t)n_members + 1);
}

// after (re)allocation: no grouping has been built for any flock layout yet
hipError_t nh_cohesion_scratch_reset(int32_t *scratch, int n_flocks, int n_members, hipStream_t s)
{
    const coh_scratch C = coh_layout(scratch, n_flocks, n_members);
    return hipMemsetAsync(C.saved[0], 0xff, sizeof(int32_t) * 2 * ((size_t)n_flocks + 4), s);
}

__global__ void k_zero_i32(int32_t *p, int n)
{
    for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

// the counting sort that regroups the lanes of every flock (k_coh_bin .. k_coh_scatter) into perm[which]
static void coh_regroup(const nh_step_params &P, const coh_scratch &C, int which, hipStream_t s)
{
    // (one launch: hipMemsetAsync of an unaligned range is up to three fill kernels)
    hipLaunchKernelGGL(k_zero_i32, dim3(min(64, (2 * C.nb + 255) / 256)), dim3(256), 0, s, C.bin_count, 2 * C.nb);
    const int gm = (P.n_members + 255) / 256;
    hipLaunchKernelGGL(k_coh_bin, dim3(gm), dim3(256), 0, s, P, C.bin_of, C.bin_count, C.saved[which]);
    hipLaunchKernelGGL(k_sp_scan_local, dim3(C.nblocks), dim3(NH_SCAN_T), 0, s, C.bin_count, C.bin_start,
                       C.block_sum, C.nb, P.grid, (const int32_t*)nullptr);
    hipLaunchKernelGGL(k_sp_scan_add, dim3(C.nblocks), dim3(NH_SCAN_T), 0, s, C.bin_start, C.block_sum, C.nb,
                       C.nblocks, (int32_t*)nullptr, P.grid, (const int32_t*)nullptr);
    hipLaunchKernelGGL(k_coh_scatter, dim3(gm), dim3(256), 0, s, P, C.bin_of, C.bin_start, C.bin_fill,
                       C.perm[which]);
}

// The cohesion term of one tick.  *parity (in/out, kept by the context) = which of the two perm
// buffers the NEXT regrouping writes.
// k_cohesion starts at once on the grouping the PREVIOUS tick left behind (any permutation of a flock's
// entries is valid; the grouping only has to be spatially coherent, and agents move ~1 wu per tick;
// k_coh_plan checks that it was built for these flock offsets and this work range, else the identity is
// used), and the regrouping for the next tick follows it -- nh_launch_cohesion_regroup, which the caller
// launches AFTER recording its "cohesion done" event: five dependent small launches leave the tick's
// critical path.  That holds for a rank that steps a slab as well: its grouping holds the slab's members
// only (k_coh_bin), so the members of the other ranks occupy no lanes.
#define COH_INLINE_PLAN_MAX 64
bool nh_launch_cohesion(const nh_step_params &P, int32_t *scratch, float *d_coh, int *parity, hipStream_t s)
{
    if(!(P.n_ents > 0 && P.n_flocks > 0 && P.n_members > 0)) return false;
    const coh_scratch C = coh_layout(scratch, P.n_flocks, P.n_members);
    // upper bound of the number of 16-member (COH_APW) waves (the identity fallback needs a lane per
    // member of the whole snapshot); surplus waves exit at once
    const int nwaves = (P.n_members + 15) / 16 + P.n_flocks;
    const int prev = *parity ^ 1;
    if(P.n_flocks <= COH_INLINE_PLAN_MAX) {
        hipLaunchKernelGGL(k_cohesion<true>, dim3(nwaves), dim3(64), 0, s, P, (const int32_t*)C.wave_off,
                           (const int32_t*)C.perm[prev], (const int32_t*)C.valid, d_coh, (const int32_t*)C.bin_start,
                           (const int32_t*)C.saved[prev]);
        return true;
    }
    hipLaunchKernelGGL(k_coh_plan, dim3(1), dim3(256), 0, s, (const int32_t*)C.bin_start, P.flock_offsets,
                       (const int32_t*)C.saved[prev], P.n_flocks, P.work_begin, P.work_end, P.members_key, C.wave_off, C.valid);
    hipLaunchKernelGGL(k_cohesion<false>, dim3(nwaves), dim3(64), 0, s, P, (const int32_t*)C.wave_off,
                       (const int32_t*)C.perm[prev], (const int32_t*)C.valid, d_coh, (const int32_t*)C.bin_start,
                       (const int32_t*)C.saved[prev]);
    return true;                                  // caller: record the event, then ..._regroup
}

void nh_launch_cohesion_regroup(const nh_step_params &P, int32_t *scratch, int *parity, hipStream_t s)
{
    const coh_scratch C = coh_layout(scratch, P.n_flocks, P.n_members);
    coh_regroup(P, C, *parity, s);
    *parity ^= 1;
}
// entries a sub-list can receive: its producers are the k_agent_mid waves with index = sub (mod NH_WL_SUB), each of
// which steps 64 entities
int nh_worklist_cap(int n_work)
{
    const int waves = (n_work + 63) / 64;
    // (+ 16: the retry list is filled by k_cp_small's waves -- four entries each, a few more of them per
    // sub-list than there are k_agent_mid waves' worth)
    return ((waves + NH_WL_SUB - 1) / NH_WL_SUB) * 64 + 16;
}

// k_agent_mid + the consumers of its work lists.  The list counters alternate between two sets:
// a launch sequence uses one and zeroes the other for its successor (no memset on the stream).
// Returns whether anything was launched (the caller flips the parity only then: a step that launches nothing
// does not clear the other set either).
bool nh_launch_agent_finish(const nh_step_params &P, const nh_nbr &NB, float *d_coh, nh_mid_rec *d_mid,
                            nh_worklists WL, int parity, const nh_step_outs &O, hipStream_t s,
                            hipStream_t side, navhip_ctx *ctx)
{
    const int nwork = P.work_end - P.work_begin;
    if(!(P.n_ents > 0 && nwork > 0)) return false;
    // SCALED_MAX_FORCE and the 1 % force threshold of movement.c:1870-1905 (same for every agent)
    const float smf = (float)((double)(0.75f / (float)P.hz) * 20.0);
    const double thresh = ((double)(0.75f / (float)P.hz) * 20.0) * 0.01;
    int32_t *zero_next = WL.count + (parity ^ 1) * NH_WL_COUNTERS;
    WL.count += parity * NH_WL_COUNTERS;
    hipLaunchKernelGGL(k_agent_mid, dim3((nwork + 63) / 64), dim3(64), 0, s, P, NB, (const float*)d_coh, d_mid, WL, O, smf, thresh);
    // the ClearPath launches.  s: the rows of 5-16 neighbours, the irregular agents.  side: the agents with 1-4
    // neighbours (most of them, outside a crowd), whatever of them needs the retry logic, then the workgroup problems
    // (17-64 neighbours).  Every wave / workgroup keeps drawing units until none are left.  (The workgroup problems on
    // a third stream beside the small ones were measured and lost -- one more fork and join on the agent stream, and in
    // a jam the searches race k_cp_rows for the chip instead of inheriting it: profiles/archive/r04_ab_cp_three_streams.txt.)
    // side: the fork.  Its hand-overs go through device memory (stream_set.hip): k_cp_rows -- it follows k_agent_mid on s --
    // stores "the work lists are complete" when it starts, and a one-lane kernel in front of k_cp_small waits for that.
    const bool fork = side != nullptr && side != s;
    hipStream_t sh = fork ? side : s;
    const nh_signal lists_ready = fork ? nh_handover_by_kernel(ctx, NH_HO_MID, s) : nh_signal{nullptr, 0};
    ctx->lists_signalled = fork;
    const int nblk = min(4096 / CP_WAVES, (nwork + 15) / 16 + 1);      // 4096 persistent waves: four per SIMD
    const int nblk_rows = min(4096 / CPR_WAVES, (nwork + 15) / 16 * (CP_WAVES / CPR_WAVES) + 1);
    // (the rows first: behind a host that is not ahead of the device -- the tick after a synchronisation -- every
    // launch in front of it delays its start by one enqueue)
    hipLaunchKernelGGL(k_cp_rows, dim3(nblk_rows), dim3(CPR_WAVES * 64), 0, s, P, NB, (const nh_mid_rec*)d_mid, WL, O,
                       (int)NH_WL_ROW3, 2, 0, lists_ready);
    if(fork) nh_handover_wait(ctx, NH_HO_MID, sh);
    hipLaunchKernelGGL(k_cp_small, dim3((nwork / 4 + 2 * NH_WL_SUB + CPS_WAVES - 1) / CPS_WAVES + 1), dim3(CPS_WAVES * 64), 0, sh, P, NB,
                       (const nh_mid_rec*)d_mid, WL, O);
    hipLaunchKernelGGL(k_cp_rows, dim3(64 * CP_WAVES / CPR_WAVES), dim3(CPR_WAVES * 64), 0, sh, P, NB, (const nh_mid_rec*)d_mid, WL, O,
                       (int)NH_WL_RETRY, 1, 1, nh_signal{nullptr, 0});
    // (the last launch on `side` clears the other set of list counters: see k_cp_heavy)
    hipLaunchKernelGGL(k_cp_heavy, dim3(nblk), dim3(CP_WAVES * 64), 0, sh, P, NB, (const nh_mid_rec*)d_mid, WL, O, zero_next);
    if(fork) nh_handover_signal(ctx, NH_HO_CP, sh);
    hipLaunchKernelGGL(k_agent_full, dim3(min(1024, (nwork + AG_WAVES - 1) / AG_WAVES)), dim3(AG_WAVES * 64), 0, s, P,
                       (const float*)d_coh, (const nh_mid_rec*)d_mid, WL, O, smf, thresh, (int32_t*)nullptr);
    if(fork) {
        // the join; the waiting kernel also says that the step has ended on s: a prefetch that follows directly starts its
        // side streams behind that (NAVHIP_PREFETCH_FOLLOWS_STEP)
        nh_handover_wait(ctx, NH_HO_CP, s, -1, NH_HO_END);
        ctx->step_end_on = ctx->ho->by_events ? nullptr : s;         // (an event is no word: nobody can follow it that way)
        ctx->step_end_signalled = true;
    }
    return true;
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
                         int rows, hipStream_t s)
{
    if(nq <= 0) return;
    if(rows == 2)
        hipLaunchKernelGGL(k_clearpath_team, dim3(nq), dim3(CP_WAVES * 64), 0, s, nq, ent, des_v, dyn,
                           n_dyn, stat, n_stat, out);
    else if(rows)
        hipLaunchKernelGGL(k_clearpath<16>, dim3((nq + 7) / 8), dim3(128), 0, s, nq, ent, des_v, dyn,
                           n_dyn, stat, n_stat, out);
    else
        hipLaunchKernelGGL(k_clearpath<64>, dim3((nq + 1) / 2), dim3(128), 0, s, nq, ent, des_v, dyn,
                           n_dyn, stat, n_stat, out);
}
