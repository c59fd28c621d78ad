// tests/hostsim/serial_thread.h -- TEST INFRASTRUCTURE ONLY.
//
// Serial (one "thread" per agent) statements of the two group-parallel parts of the device step -- the
// neighbour walk (nbr_walk_row, agent_group.h) and one ClearPath attempt (clearpath_grp) -- so that the
// host-compiled unit tests can drive the per-agent bodies that DO run on the device unchanged
// (pool_record, mid_thread, post_thread in permafrost-engine_amd/csrc/agent_thread.h) end to end and pin
// them to the reference build on a machine without a GPU.  Not part of the product: nothing here is
// compiled into libnavhip.so.
#pragma once
#include "agent_thread.h"
// ---------------------------------------------------------------------------------------------
// neighbour walk: the SERIAL statement of what nbr_walk_row (agent_group.h) computes with 16 lanes.
// Compiled for the host tests only (tests/hostsim), which pin it to the reference build; the GPU
// tests then pin the device's 16-lane form to the same reference.
// ---------------------------------------------------------------------------------------------
// Visiting order == bg_*_inrange_circle (bitmap_grid.h:1408-1466): coarse 8x8 blocks row-major,
// inside a block fine rows top to bottom, cells left to right, packed elements in order.  The cells
// of one fine row inside one coarse block are contiguous in the cell-sorted pool.
//
// The r = 10 query of find_neighbours is the r = 30 walk restricted to d <= 10: its box lies inside
// the r = 30 box, the visiting key (coarse block, fine row, cell, slot) does not depend on the query,
// and an element of a cell outside the r = 10 box fails the distance test anyway.  Both caps are
// applied where the reference's two separate queries stop.
//
// Left to the wave-per-agent path (NH_NB_IRREGULAR): a garrisoned entity among the hits
// (filter_garrisoned, position.c:100-119, permutes the list) and queries that take the reference's
// wide linear scan or miss the grid.
NH_FN void nbr_walk_thread(const nh_grid &G, int k, float scaled_max_force, const double *exp_tab,
                           const nh_nbr &NB)
{
    const float4 me4 = G.recA[k];
    const uint32_t mybits = nh_f2u(me4.w);
    const int uid = (int)(mybits >> NH_PB_UID_SHIFT);
    const v2 me = mkv(me4.x, me4.y);
    const float my_radius = me4.z;
    const int32_t icx = bg_scale(me.x), icy = bg_scale(me.z);
    const int32_t ir30 = bg_scale(30.0f), ir10 = bg_scale(10.0f);
    sp_extent E, E10;
    if(!sp_query_extent(G, icx, icy, ir30, E) || !sp_query_extent(G, icx, icy, ir10, E10)
    || E.wide || E10.wide) {
        NB.cnt[uid] = (NH_NB_IRREGULAR | NH_NB_DONE) << 16;
        return;
    }
    const int32_t lim30 = ir30 * ir30, lim10 = ir10 * ir10;
    int raw30 = 0, raw10 = 0, n_dyn = 0, n_stat = 0;
    bool irregular = false, stop = false;
    v2 acc = mkv(0.0f, 0.0f);
    const int cxc_lo = E.cx_lo >> 3, cxc_hi = E.cx_hi >> 3;
    const int cyc_lo = E.cy_lo >> 3, cyc_hi = E.cy_hi >> 3;
    for(int cyc = cyc_lo; cyc <= cyc_hi && !stop; cyc++) {
    for(int cxc = cxc_lo; cxc <= cxc_hi && !stop; cxc++) {
        const int fy0 = cyc * 8 > E.cy_lo ? cyc * 8 : E.cy_lo;
        const int fy1 = cyc * 8 + 7 < E.cy_hi ? cyc * 8 + 7 : E.cy_hi;
        const int fx0 = cxc * 8 > E.cx_lo ? cxc * 8 : E.cx_lo;
        const int fx1 = cxc * 8 + 8 < E.cx_hi + 1 ? cxc * 8 + 8 : E.cx_hi + 1;
        for(int fy = fy0; fy <= fy1 && !stop; fy++) {
            int sx0 = fx0, sx1 = fx1;
            if(raw30 >= NH_SEP_CAP) {
                // the r = 30 query has stopped: only cells of the r = 10 box can still contribute
                if(fy < E10.cy_lo || fy > E10.cy_hi) continue;
                sx0 = fx0 > E10.cx_lo ? fx0 : E10.cx_lo;
                sx1 = fx1 < E10.cx_hi + 1 ? fx1 : E10.cx_hi + 1;
                if(sx0 >= sx1) continue;
            }
            const int b = G.cell_start[fy * G.grid_w + sx0], e = G.cell_start[fy * G.grid_w + sx1];
            for(int q = b; q < e; q++) {
                const float4 c = G.recA[q];
                // (elements clamped into a border cell may be far away: range-check before squaring
                // in 32 bits; r = 30: |d| <= 7680 + 4095 otherwise)
                const int32_t dx = bg_scale(c.x) - icx, dy = bg_scale(c.y) - icy;
                const bool nearby = (uint32_t)(dx + 32767) < 65535u && (uint32_t)(dy + 32767) < 65535u;
                const int32_t d2 = nearby ? dx * dx + dy * dy : 0x7fffffff;
                if(d2 > lim30) continue;
                const bool hit30 = raw30 < NH_SEP_CAP;
                const bool hit10 = d2 <= lim10 && raw10 < NH_NEAR_CAP;
                if(!hit30 && !hit10) continue;
                const uint32_t bits = nh_f2u(c.w);
                if(bits & NH_PB_GARRISONED) { irregular = true; stop = true; break; }
                const bool other = q != k && (bits & NH_PB_MOVABLE) && !((mybits ^ bits) & NH_PB_AIR);
                if(hit30) {
                    raw30++;
                    if(other) {
                        v2 term;
                        if(separation_term(me, my_radius, mkv(c.x, c.y), c.z, exp_tab, term))
                            acc = vadd(acc, term);
                    }
                }
                if(hit10) {
                    raw10++;
                    if(other && c.z != 0.0f) {
                        if(bits & NH_PB_STATIC) {
                            if(n_stat < NH_MAX_NEIGHBOURS) { nbr_store(NB, uid, 32 + n_stat, c, make_float2(0.0f, 0.0f)); n_stat++; }
                        }else{
                            if(n_dyn < NH_MAX_NEIGHBOURS) { nbr_store(NB, uid, n_dyn, c, G.recV[q]); n_dyn++; }
                        }
                    }
                }
                if(raw30 >= NH_SEP_CAP && raw10 >= NH_NEAR_CAP) { stop = true; break; }
            }
        }
    }}
    if(irregular) {
        NB.cnt[uid] = (NH_NB_IRREGULAR | NH_NB_DONE) << 16;
        return;
    }
    v2 sep = mkv(0.0f, 0.0f);
    if(raw30 > 0)                                   // `if(0 == num_near) return 0`, movement.c:1737
        sep = vtrunc(vscale(acc, -1.0f), scaled_max_force);
    NB.sep[uid] = make_float2(sep.x, sep.z);
    NB.cnt[uid] = (uint32_t)n_dyn | ((uint32_t)n_stat << 8) | (NH_NB_DONE << 16);
}


// ---------------------------------------------------------------------------------------------
// ClearPath candidate search: the SERIAL statement of one attempt of clearpath_grp (agent_group.h),
// for the host tests (tests/hostsim) -- same candidate set, same bound, same tie-break.
// ---------------------------------------------------------------------------------------------
// cones: 2 float4 per neighbour at cones[i * cstride].  Returns false when no candidate lies outside
// the combined obstacle (the device then runs remove_furthest, :390, and retries).
NH_FN bool cp_light_thread(const nh_nbr &NB, int uid, const cpent &ent, v2 des_v, int n_dyn, int n_stat,
                           float4 *cones, int cstride, v2 &result)
{
    int n_cones = 0;
    const int n = n_dyn + n_stat;
    for(int j = 0; j < n; j++) {
        const bool is_stat = j >= n_dyn;
        const cpent nb = nbr_load(NB, uid, is_stat ? 32 + (j - n_dyn) : j);
        if(vlen(vsub(nb.pos, ent.pos)) < CP_EPS) continue;
        v2 apex, left, right; float sl, sr;
        make_cone(ent, nb, !is_stat, apex, left, right, sl, sr);
        cones[(2 * n_cones) * cstride]     = make_float4(apex.x, apex.z, sl, sr);
        cones[(2 * n_cones + 1) * cstride] = make_float4(left.x, left.z, right.x, right.z);
        n_cones++;
    }
    const int n_rays = 2 * n_cones;
    const int npairs = n_rays * n_rays;
    {   // clearpath_new_velocity :604: des_v admissible as it is?
        bool in = false;
        for(int c = 0; c < n_cones && !in; c++)
            in = cone_contains(cones[(2 * c) * cstride], cones[(2 * c + 1) * cstride], vadd(ent.pos, des_v));
        if(!in) { result = des_v; return true; }
    }
    // compute_vnew (:368) keeps the first strictly-smaller distance in candidate order = the minimum
    // of (distance, order index) over the candidates outside the obstacle: the order of evaluation is
    // free, and a candidate that cannot beat the best one so far needs no inside-obstacle test.
    // Projections of des_v first (order index npairs + i): they tighten the bound at once.
    float best_len = INFINITY;
    int best_idx = 0x7fffffff;
    v2 best = mkv(0.0f, 0.0f);
    bool any = false;
#define NH_CP_CONSIDER(PT, IDX) do {                                                                   \
        const v2 curr_ = vsub((PT), ent.pos);                                                          \
        const float len_ = vlen(vsub(des_v, curr_));                                                   \
        if(!any || len_ < best_len || (len_ == best_len && (IDX) < best_idx)) {                        \
            bool inside_ = false;                                                                      \
            for(int c_ = 0; c_ < n_cones && !inside_; c_++)                                            \
                inside_ = cone_contains(cones[(2 * c_) * cstride], cones[(2 * c_ + 1) * cstride], (PT)); \
            if(!inside_) {                                                                             \
                if(len_ < best_len || (len_ == best_len && (IDX) < best_idx)) {                        \
                    best_len = len_; best_idx = (IDX); best = curr_;                                   \
                }                                                                                      \
                any = true;                                                                            \
            }                                                                                          \
        }                                                                                              \
    } while(0)
    for(int i = 0; i < n_rays; i++) {
        const float4 Ai = cones[(i & ~1) * cstride], Bi = cones[(i | 1) * cstride];
        const v2 dir = (i & 1) ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), point = mkv(Ai.x, Ai.y);
        const float plen = vdot(dir, des_v);
        const v2 pt = vadd(point, vscale(dir, plen));
        NH_CP_CONSIDER(pt, npairs + i);
    }
    for(int i = 0; i < n_rays; i++) {
        const float4 Ai = cones[(i & ~1) * cstride], Bi = cones[(i | 1) * cstride];
        const v2 pi = mkv(Ai.x, Ai.y), di = (i & 1) ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y);
        const float si = (i & 1) ? Ai.w : Ai.z;
        for(int j = 0; j < n_rays; j++) {
            if(i == j) continue;
            const float4 Aj = cones[(j & ~1) * cstride], Bj = cones[(j | 1) * cstride];
            v2 pt;
            if(!ray_isect(pi, di, si, mkv(Aj.x, Aj.y), (j & 1) ? mkv(Bj.z, Bj.w) : mkv(Bj.x, Bj.y),
                          (j & 1) ? Aj.w : Aj.z, pt))
                continue;
            NH_CP_CONSIDER(pt, i * n_rays + j);
        }
    }
#undef NH_CP_CONSIDER
    result = best;
    return any;
}
