// region_kernels.hip -- flow fields over arbitrary square tile regions that straddle chunk
// boundaries, gfx950.
//
// Reference semantics (navigation/field.c): field_build_integration_region :582 (4-connected
// Dijkstra over the region in GLOBAL tile coordinates, neighbours via field_neighbours_grid_global
// :252, optional overlay mask of extra blocked tiles), then either
//   field_build_flow_unaligned :800  (whole region, two 4-bit directions per byte, set_flow_cell
//                                     :786) -- N_CellArrivalFieldCreate :2445,
//                                     N_GroupArrivalFieldCreate :2525 (96x96 regions, formation.c:78)
//   field_build_flow_region :763     (a 64x64 window of the region written in place into a chunk
//                                     field) -- field_update_enemies :1537, field_update_entity
//                                     :1615, field_update_zone :1822 (128x128 padded regions).
// The seeds (target tile, open formation slots, enemy / entity / zone footprints) are produced by
// the host's game-side queries and handed over as a tile list.
//
// One 256-thread workgroup per region; u32 integration tile + u8 cost tile in dynamic LDS
// (96x96: 45 KB, 128x128: 80 KB); chaotic min-plus relaxation to the integer-exact fixpoint, then
// the bake with the reference's neighbour priority.
#include "navhip_internal.h"

static_assert(sizeof(navhip_region_req) == 32, "navhip_region_req must stay 32 bytes");

__global__ __launch_bounds__(256) void k_region_field(nh_map_view map, const navhip_region_req *reqs,
                                                      int n, const int16_t *seeds, const int16_t *overlay,
                                                      uint8_t *out, size_t out_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int ri = blockIdx.x, t = threadIdx.x;
    if(ri >= n) return;
    const navhip_region_req rq = reqs[ri];
    const int dim = rq.rdim, cells = dim * dim;
    uint32_t *dist = (uint32_t*)smem;                    // [cells]
    uint8_t *pc = smem + (size_t)cells * 4;              // [cells] cost, 0xff = cannot be entered
    const nh_layer_view &L = map.layers[rq.layer];
    const int H = map.h * 64, W = map.w * 64;

    // ---- stage: which region tiles may the wavefront enter, and at what cost -----------------
    for(int i = t; i < cells; i += 256) {
        const int ar = rq.base_abs_r + i / dim, ac = rq.base_abs_c + i % dim;
        uint8_t v = NAVHIP_COST_IMPASSABLE;
        if(ar >= 0 && ar < H && ac >= 0 && ac < W) {                // M_Tile_RelativeDesc: on the map
            const size_t cell = ((size_t)((ar >> 6) * map.w + (ac >> 6)) << 12) + (ar & 63) * 64 + (ac & 63);
            const uint32_t cst = L.cost[cell];
            const uint32_t blk = L.blockers ? L.blockers[cell] : 0;
            bool passable;
            if(cst == NAVHIP_COST_IMPASSABLE) {
                passable = false;
            }else if(rq.enemies == 0) {
                passable = blk == 0;                                 // field_tile_passable
            }else{
                bool enemies_only = true;                            // field_tile_passable_no_enemies
                if(L.factions) {
                    const size_t cb = (size_t)((ar >> 6) * map.w + (ac >> 6)) << 12;
                    const uint8_t *fp = L.factions + cb * NAVHIP_MAX_FACTIONS + (ar & 63) * 64 + (ac & 63);
                    for(int f = 0; f < NAVHIP_MAX_FACTIONS; f++)
                        if(fp[(size_t)f << 12] && !(rq.enemies & (1u << f))) { enemies_only = false; break; }
                }
                passable = enemies_only || blk == 0;
            }
            if(passable) v = (uint8_t)cst;
        }
        pc[i] = v;
        dist[i] = NH_INF_U32;
    }
    __syncthreads();
    // overlay mask (build_overlay_mask :571): extra blocked tiles, never entered
    for(uint32_t k = t; k < rq.overlay_count; k += 256) {
        const int dr = overlay[2 * (rq.overlay_begin + k)] - rq.base_abs_r;
        const int dc = overlay[2 * (rq.overlay_begin + k) + 1] - rq.base_abs_c;
        if(dr >= 0 && dr < dim && dc >= 0 && dc < dim) pc[dr * dim + dc] = NAVHIP_COST_IMPASSABLE;
    }
    __syncthreads();
    // seeds get 0 unconditionally (pushed without a passability test, e.g. field.c:2497,2577)
    for(uint32_t k = t; k < rq.seed_count; k += 256) {
        const int dr = seeds[2 * (rq.seed_begin + k)] - rq.base_abs_r;
        const int dc = seeds[2 * (rq.seed_begin + k) + 1] - rq.base_abs_c;
        if(dr >= 0 && dr < dim && dc >= 0 && dc < dim) dist[dr * dim + dc] = 0;
    }
    __syncthreads();

    // ---- relaxation to the fixpoint (== Dijkstra distances, integer exact) ---------------------
    for(;;) {
        int changed = 0;
        for(int i = t; i < cells; i += 256) {
            const uint32_t cst = pc[i];
            if(cst == NAVHIP_COST_IMPASSABLE) continue;
            const int r = i / dim, c = i % dim;
            uint32_t m = NH_INF_U32;
            if(r > 0)       m = min(m, dist[i - dim]);
            if(r < dim - 1) m = min(m, dist[i + dim]);
            if(c > 0)       m = min(m, dist[i - 1]);
            if(c < dim - 1) m = min(m, dist[i + 1]);
            if(m < NH_INF_U32) {
                const uint32_t nd = m + cst;
                if(nd < dist[i]) { dist[i] = nd; changed = 1; }
            }
        }
        if(!__syncthreads_or(changed)) break;
    }

    // ---- bake ---------------------------------------------------------------------------------
    uint8_t *o = out + (size_t)ri * out_stride;
    const bool window = rq.out_mode == 1;
    const int wdim = window ? min(64, dim) : dim;
    const int ncell_out = wdim * (window ? wdim : dim);
    for(int j = t; j < (window ? ncell_out : cells / 2); j += 256) {
        // packed mode: one thread owns one output byte = two horizontally adjacent cells
        const int ncell = window ? 1 : 2;
        uint8_t byte = 0;
        bool keep_byte = false;
        for(int h = 0; h < ncell; h++) {
            int r, c;
            if(window) { r = j / wdim + rq.roff; c = j % wdim + rq.coff; }
            else       { r = (2 * j + h) / dim;  c = (2 * j + h) % dim; }
            const int i = r * dim + c;
            const uint32_t d = dist[i];
            uint32_t dir;
            bool write = true;
            if(d >= NH_INF_U32) {
                dir = NAVHIP_FD_NONE;             // packed: stays 0 (memset); window: left untouched
                write = !window;
            }else if(d == 0) {
                dir = NAVHIP_FD_NONE;
            }else{
                const uint32_t I = NH_INF_U32;
                const uint32_t dn = r > 0 ? dist[i - dim] : I, ds = r < dim - 1 ? dist[i + dim] : I;
                const uint32_t dw = c > 0 ? dist[i - 1] : I,   de = c < dim - 1 ? dist[i + 1] : I;
                const uint32_t dnw = (r > 0 && c > 0) ? dist[i - dim - 1] : I;
                const uint32_t dne = (r > 0 && c < dim - 1) ? dist[i - dim + 1] : I;
                const uint32_t dsw = (r < dim - 1 && c > 0) ? dist[i + dim - 1] : I;
                const uint32_t dse = (r < dim - 1 && c < dim - 1) ? dist[i + dim + 1] : I;
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
            if(window) {
                if(write) byte = (uint8_t)dir; else keep_byte = true;
            }else{
                // set_flow_cell :786: even column -> high nibble, odd column -> low nibble
                byte |= (uint8_t)(h == 0 ? dir << 4 : dir);
            }
        }
        if(window) { if(!keep_byte) o[(j / wdim) * 64 + (j % wdim)] = byte; }
        else       o[j] = byte;
    }
}

void nh_launch_region_fields(navhip_ctx *ctx, const navhip_region_req *d_reqs, int n, int max_dim,
                             const int16_t *d_seeds, const int16_t *d_overlay, uint8_t *d_out,
                             size_t out_stride, hipStream_t s)
{
    nh_map_view mv;
    mv.w = ctx->w;
    mv.h = ctx->h;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        const navhip_layer &L = ctx->layers[l];
        mv.layers[l] = nh_layer_view{L.cost, L.blockers, L.local_islands, L.factions,
                                     L.passmask, L.unit_cost, L.changed, L.islands};
    }
    const size_t lds = (size_t)max_dim * max_dim * 5;
    // (per device, not per process: every context's device needs the attribute)
    static bool attr_set[64] = {false};
    const int devi = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    if(!attr_set[devi]) {
        hipFuncSetAttribute((const void*)k_region_field, hipFuncAttributeMaxDynamicSharedMemorySize,
                            128 * 128 * 5);
        attr_set[devi] = true;
    }
    if(n > 0)
        hipLaunchKernelGGL(k_region_field, dim3(n), dim3(256), lds, s, mv, d_reqs, n, d_seeds,
                           d_overlay, d_out, out_stride);
}
