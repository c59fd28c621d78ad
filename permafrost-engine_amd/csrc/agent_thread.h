// agent_thread.h -- the thread-per-agent bodies of the movement step (one THREAD = one agent).
//
//   nbr_walk_thread   the two neighbour gathers of move_velocity_work in ONE walk over the spatial
//                     hash: separation_force (movement.c:1690, r = 30, cap 128) accumulated in the
//                     reference's candidate order, and find_neighbours (movement.c:2768, r = 10, cap
//                     512, 32 + 32) as lists of pool slots.  Needs only the entity snapshot.
//   mid_thread        desired direction (flow sampling), arrive force, tile probes, the priority
//                     ladder of point_seek_vpref (:1870) and friends -> preferred velocity; then the
//                     first question of clearpath_new_velocity (clearpath.c:604): is des_v already
//                     outside every velocity obstacle?  Most agents finish here.
//   cp_light_thread   the ClearPath candidate search (compute_vo_xpoints :321,
//                     compute_vdes_proj_points :344, compute_vnew :368) for agents with at most
//                     NH_LIGHT_MAX neighbours, sequentially in the reference's own order.
//   post_thread       vec2_truncate(v, max_speed / hz) (:3464) + the position accept test of
//                     entity_compute_update (:2336-2358).
//
// Why a thread per agent: in the benchmark world an agent has ~35 separation candidates and ~2
// ClearPath neighbours; a wave per agent left most lanes idle in every phase (1 017 wave
// instructions per agent, round 1).  Agents that DO fill a wave -- more than NH_LIGHT_MAX ClearPath
// neighbours: >= 100 ray pairs x cone tests -- go to the wave-per-agent kernels (agent_kernels.hip).
//
// No cross-lane operations here: compiles for the device and, with -DNH_HOSTSIM, for the host-side
// unit tests (tests/hostsim).
#pragma once
#include "agent_math.h"

enum { AM_IDLE = 0,        // still or combat held: velocity 0, no neighbour work
       AM_ZERO_VPREF,      // turning / formation assignment not ready: vpref = 0, ClearPath still runs
       AM_POINT_SEEK, AM_ENEMY_SEEK, AM_FORM_CELL, AM_FORM_POINT,
       AM_UNSUPPORTED };   // formation state without formation inputs

enum { DISP_DONE = 0,      // out_vel is final (before truncation)
       DISP_ROW0, DISP_ROW1, DISP_ROW2, DISP_ROW3,   // ClearPath on a row of 16 lanes: 1-2, 3-4, 5-8, 9-16 neighbours
       DISP_WAVE,          // ClearPath on a workgroup: 17-32 neighbours
       DISP_HEAVY,         // ClearPath on a workgroup: 33-64 neighbours
       DISP_FULL };        // whole step on a wave (irregular gather)

#define NH_SEP_CAP   128   /* near_ents[128],  movement.c:1695 */
#define NH_NEAR_CAP  512   /* near_ents[512],  movement.c:2779 */
#define NH_MAX_NEIGHBOURS 32
#define NH_ROW_MAX  16   /* most ClearPath neighbours an agent may have for the 16-lane path */

// ---------------------------------------------------------------------------------------------
// pool record of entity i (written by the last pass of the spatial-hash build)
// ---------------------------------------------------------------------------------------------
NH_FN void pool_record(int i, const float *pos_xz, const nh_pack_src &src, int work_begin, int work_end,
                       float4 &recA, float2 &recV)
{
    uint32_t bits = (uint32_t)i << NH_PB_UID_SHIFT;
    float radius = 0.0f;
    v2 vel = mkv(0.0f, 0.0f);
    const v2 pos = mkv(pos_xz[2 * i], pos_xz[2 * i + 1]);
    if(src.flags) {
        const uint32_t fl = src.flags[i];
        const int state = src.state[i];
        radius = src.radius[i];
        vel = mkv(src.vel_xz[2 * i], src.vel_xz[2 * i + 1]);
        if(fl & NAVHIP_ENTITY_FLAG_MOVABLE)    bits |= NH_PB_MOVABLE;
        if(fl & NAVHIP_ENTITY_FLAG_AIR)        bits |= NH_PB_AIR;
        if(fl & NAVHIP_ENTITY_FLAG_WATER)      bits |= NH_PB_WATER;
        if(fl & NAVHIP_ENTITY_FLAG_GARRISONED) bits |= NH_PB_GARRISONED;
        // find_neighbours :2815-2817: G_Arrival_NeighbourSettling (arrival.c:1042) -- committed to a
        // valid slot and within 1.5 radii (ARRIVAL_SINK_TOLERANCE) of it
        bool at_slot = false;
        if(src.arrival_flags && (src.arrival_flags[i] & 1)) {
            const v2 sink = mkv(src.arrival_sink_xz[2 * i], src.arrival_sink_xz[2 * i + 1]);
            at_slot = vlen(vsub(sink, pos)) < radius * 1.5f;
        }
        if(state_is_still(state) || vlen(vel) < 0.3f || at_slot) bits |= NH_PB_STATIC;   // CLEARPATH_STILL_SPEED
        if(state_is_still(state) || (fl & NAVHIP_ENTITY_FLAG_COMBAT_HELD) || i < work_begin || i >= work_end)
            bits |= NH_PB_IDLE;
    }
    recA = make_float4(pos.x, pos.z, radius, nh_u2f(bits));
    recV = make_float2(vel.x, vel.z);
}


// ---------------------------------------------------------------------------------------------
// position accept + outputs
// ---------------------------------------------------------------------------------------------
NH_FN void post_thread(const nh_step_params &P, int uid, v2 me, int state, uint32_t my_flags,
                       float my_radius, v2 raw_vel, float vel_cap, uint32_t status, const nh_step_outs &O)
{
    v2 new_pos = me;
    // vec2_truncate(new velocity, max_speed / hz), movement.c:3464
    const v2 out_vel = vtrunc(raw_vel, vel_cap);
    O.vel_xz[2 * uid] = out_vel.x; O.vel_xz[2 * uid + 1] = out_vel.z;
    // entity_compute_update: a garrisoned entity returns before the position update (:2341-2348);
    // the heading gate (:2322-2334) stays with the host's state machine
    if(!state_is_still(state) && !(my_flags & NAVHIP_ENTITY_FLAG_GARRISONED)) {
        const int layer = nav_layer_for(my_flags, my_radius);
        v2 cand = vadd(me, out_vel);
        const bool on_blocked = pos_blocked(P, layer, me.x, me.z);
        bool cand_path = false, cand_blk = false;
        tiledesc t;
        if(tile_for_point(P, cand.x, cand.z, t)) {               // one lookup, both predicates
            const uint32_t pb = tile_probe(P, layer, t);
            cand_path = (pb & 1u) != 0;
            cand_blk = (pb & 2u) != 0;
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
// ClearPath neighbour access: pool slot -> struct cp_ent (find_neighbours, movement.c:2799-2826)
// ---------------------------------------------------------------------------------------------
NH_FN cpent nbr_cpent(const nh_grid &G, int slot, bool is_static)
{
    const float4 a = G.recA[slot];
    cpent nb;
    nb.pos = mkv(a.x, a.y);
    nb.radius = a.z;
    nb.vel = mkv(0.0f, 0.0f);                      // static: velocity forced to zero (:2820)
    if(!is_static) { const float2 v = G.recV[slot]; nb.vel = mkv(v.x, v.y); }
    return nb;
}

// the same struct cp_ent as a record of the neighbour table (nh_nbr.rec): written by the neighbour walk
// from the pool record it has just tested, read back by the ClearPath kernels
NH_FN void nbr_store(const nh_nbr &NB, int uid, int idx, const float4 &a, float2 v)
{
    float *d = NB.rec + (size_t)uid * NB.stride + 5 * idx;
    d[0] = a.x; d[1] = a.y; d[2] = v.x; d[3] = v.y; d[4] = a.z;
}

NH_FN cpent nbr_load(const nh_nbr &NB, int uid, int idx)
{
    const float *s = NB.rec + (size_t)uid * NB.stride + 5 * idx;
    cpent nb;
    nb.pos = mkv(s[0], s[1]); nb.vel = mkv(s[2], s[3]); nb.radius = s[4];
    return nb;
}

// ---------------------------------------------------------------------------------------------
// per-agent line of sight to the destination (N_HasDestLOS, nav.c:4026, cache-hit path)
// ---------------------------------------------------------------------------------------------
NH_FN bool dest_los(const nh_step_params &P, int uid, int flock, uint32_t &status)
{
    const uint8_t given = P.has_dest_los[uid];
    if(given != NAVHIP_LOS_LOOKUP || !P.los_pool || !P.flock_los_slot)
        return given != 0 && given != NAVHIP_LOS_LOOKUP;
    if(flock < 0) return false;                       // (compute_los_state: no flock, no LOS)
    const float *lp = P.los_pos_xz ? P.los_pos_xz : P.pos_xz;
    tiledesc t;
    if(!tile_for_point(P, lp[2 * uid], lp[2 * uid + 1], t)) return false;
    const int slot = P.flock_los_slot[(size_t)flock * (P.map.w * P.map.h) + t.chunk_r * P.map.w + t.chunk_c];
    if(slot < 0) { status |= NAVHIP_ST_LOS_MISS; return false; }
    return (P.los_pool[((size_t)slot << 12) + t.tile_r * 64 + t.tile_c] & 1) != 0;
}

// ---------------------------------------------------------------------------------------------
// steering forces -> preferred velocity
// ---------------------------------------------------------------------------------------------
// point_seek_vpref :1870 / enemy_seek_vpref :1946 / cell_arrival_seek_vpref :1908 /
// formation_seek_vpref :1985, given the arrive and separation terms and the five tile probes
NH_FN v2 vpref_from_forces(const nh_step_params &P, int uid, int mode, v2 me, v2 vel, int flock, v2 arrive,
                           v2 separation, uint32_t probes, const float *coh_xz, float scaled_max_force,
                           double force_thresh)
{
    const int hz = P.hz;
    v2 steer;
    if(mode == AM_ENEMY_SEEK) {
        // enemy_seek_vpref :1946 (no priorities, no nullify)
        v2 a = vscale(arrive, 0.5f), s = vscale(separation, 0.6f);
        v2 ret = mkv(0.0f, 0.0f);
        ret = vadd(ret, a); ret = vadd(ret, s);
        steer = vtrunc(ret, scaled_max_force);
    }else{
        const bool to_cell = mode == AM_FORM_CELL, formm = mode != AM_POINT_SEEK;
        v2 cohesion, align = mkv(0.0f, 0.0f), cell = mkv(0.0f, 0.0f);
        if(formm) {
            cohesion = mkv(P.form_cohesion_xz[2 * uid], P.form_cohesion_xz[2 * uid + 1]);
            align = mkv(P.form_align_xz[2 * uid], P.form_align_xz[2 * uid + 1]);
            cell = mkv(P.cell_pos_xz[2 * uid], P.cell_pos_xz[2 * uid + 1]);
        }else{
            cohesion = (flock >= 0) ? mkv(coh_xz[2 * uid], coh_xz[2 * uid + 1]) : mkv(0.0f, 0.0f);
        }
        steer = mkv(0.0f, 0.0f);
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
            steer = nullify_impass_bits(probes, steer);
            if((double)vlen(steer) > force_thresh) break;
        }
    }
    v2 accel = vscale(steer, 1.0f / 1.0f);
    v2 vpref = vtrunc(vadd(vel, accel), P.speed[uid] / (float)hz);
    if(mode == AM_FORM_CELL || mode == AM_FORM_POINT) {
        const v2 f_drag = mkv(P.form_drag_xz[2 * uid], P.form_drag_xz[2 * uid + 1]);
        if(vlen(f_drag) > CP_EPS)                            // :1935 / :2018
            vpref = vtrunc(vpref, (float)(((double)P.speed[uid] * 0.75) / (double)hz));
    }
    return vpref;
}

// ---------------------------------------------------------------------------------------------
// the per-agent scalar chain: desired direction, arrive force, probes, ladder -> vpref
// ---------------------------------------------------------------------------------------------
// The chain in two halves.  Half A is everything that reads the snapshot, the flow / LOS fields and the map only -- the
// desired direction (a chain of dependent loads), the arrive force, the tile probes --: nothing of the neighbour walk
// or the cohesion term, so it can run in the shadow of k_cohesion behind the neighbour walk.  Half B joins the three:
// forces -> vpref -> which ClearPath list.  mid_thread = A then B on one thread (the fused kernel, the wave-per-agent
// path and the host-side unit tests); k_agent_mid_a / _b run them as two launches with the record in between.
// R.mode == AM_IDLE after half A: a still / combat-held entity (nothing more to do).
NH_FN void mid_thread_a(const nh_step_params &P, int uid, float scaled_max_force, nh_mid_rec &R)
{
    const int state = P.state[uid];
    const uint32_t my_flags = P.flags[uid];
    R.vpref[0] = R.vpref[1] = R.vdes[0] = R.vdes[1] = R.arrive[0] = R.arrive[1] = 0.0f;
    R.probes = 0; R.status = 0; R.mode = AM_IDLE;
    R.vel_cap = P.max_speed[uid] / (float)P.hz;
    if(state_is_still(state) || (my_flags & NAVHIP_ENTITY_FLAG_COMBAT_HELD))
        return;

    const v2 me = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
    const v2 vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
    const float my_radius = P.radius[uid], max_speed = P.max_speed[uid];
    const int flock = P.flock[uid], hz = P.hz;
    const int layer = nav_layer_for(my_flags, my_radius);
    // (the tile probes are requested here, in front of the dependent chain of the flow sampling)
    const uint32_t probes_here = P.map.layers[layer].cost ? probe_tiles_bits(P, layer, me) : 0u;
    uint32_t status = 0;
    v2 vdes = mkv(0.0f, 0.0f), arrive = mkv(0.0f, 0.0f);
    int mode;
    const bool form = state == NAVHIP_STATE_MOVING_IN_FORMATION || state == NAVHIP_STATE_ARRIVING_TO_CELL;
    // seek target: G_Arrival_SeekTarget (arrival.c:1034) -- the unit's slot once it is committed to
    // it and the flock's arrival region is filling, else the flock target (movement.c:1751-1752)
    v2 target = me;
    if(flock >= 0) target = mkv(P.flock_target_xz[2 * flock], P.flock_target_xz[2 * flock + 1]);
    if(flock >= 0 && P.arrival_flags && (P.arrival_flags[uid] & 3) == 3)
        target = mkv(P.arrival_sink_xz[2 * uid], P.arrival_sink_xz[2 * uid + 1]);
    if(state == NAVHIP_STATE_TURNING) {
        mode = AM_ZERO_VPREF;
    }else if(state == NAVHIP_STATE_SEEK_ENEMIES || state_uses_point_seek(state)) {
        vdes = load_vdes(P, uid, flock, me, status);
        if(state_uses_point_seek(state)) {
            const bool los = dest_los(P, uid, flock, status);
            arrive = arrive_force(me, vel, target, vdes, los, max_speed, hz, scaled_max_force);
            mode = AM_POINT_SEEK;
        }else{
            // arrive_force_enemies, movement.c:1593
            v2 desired = vscale(vdes, max_speed / (float)hz);
            arrive = vtrunc(vsub(desired, vel), scaled_max_force);
            mode = AM_ENEMY_SEEK;
        }
    }else if(P.form_ready && form) {
        if(!P.form_ready[uid]) {
            mode = AM_ZERO_VPREF;
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
                mode = AM_FORM_CELL;
            }else{
                const bool los = dest_los(P, uid, flock, status);
                // formation_seek: the flock target itself (movement.c:1985-2002)
                const v2 ftarget = (flock >= 0) ? mkv(P.flock_target_xz[2 * flock], P.flock_target_xz[2 * flock + 1]) : me;
                arrive = arrive_force(me, vel, ftarget, vdes, los, max_speed, hz, scaled_max_force);
                mode = AM_FORM_POINT;
            }
        }
    }else{
        mode = AM_UNSUPPORTED;
        status |= NAVHIP_ST_UNSUPPORTED;
    }
    R.mode = (uint8_t)mode;
    R.vdes[0] = vdes.x; R.vdes[1] = vdes.z;
    R.status = (uint8_t)status;
    if(mode == AM_UNSUPPORTED)
        return;
    R.arrive[0] = arrive.x; R.arrive[1] = arrive.z;
    R.probes = (uint16_t)((mode >= AM_POINT_SEEK && mode <= AM_FORM_POINT) ? probes_here : 0u);
}

NH_FN int mid_thread_b(const nh_step_params &P, int uid, const nh_nbr &NB, const float *coh_xz,
                       float scaled_max_force, double force_thresh, nh_mid_rec &R, v2 &out_vel)
{
    out_vel = mkv(0.0f, 0.0f);
    const int mode = R.mode;
    if(mode == AM_IDLE || mode == AM_UNSUPPORTED)
        return DISP_DONE;
    const uint32_t cnt = NB.cnt[uid];
    if((cnt >> 16) & NH_NB_IRREGULAR)
        return DISP_FULL;                                  // (arrive and probes are in the record)
    const float2 s2 = NB.sep[uid];
    v2 vpref = mkv(0.0f, 0.0f);
    if(mode != AM_ZERO_VPREF) {
        const v2 me = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
        const v2 vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]);
        vpref = vpref_from_forces(P, uid, mode, me, vel, P.flock[uid], mkv(R.arrive[0], R.arrive[1]), mkv(s2.x, s2.y),
                                  (uint32_t)R.probes, coh_xz, scaled_max_force, force_thresh);
    }
    R.vpref[0] = vpref.x; R.vpref[1] = vpref.z;

    const int n = (int)(cnt & 0xff) + (int)((cnt >> 8) & 0xff);
    if(n == 0) {                                       // inside_pcr of nothing is false (clearpath.c:604)
        out_vel = vpref;
        return DISP_DONE;
    }
    return n <= 2 ? DISP_ROW0 : n <= 4 ? DISP_ROW1 : n <= 8 ? DISP_ROW2 : n <= NH_ROW_MAX ? DISP_ROW3 : n <= 32 ? DISP_WAVE : DISP_HEAVY;
}

NH_FN int mid_thread(const nh_step_params &P, int uid, const nh_nbr &NB, const float *coh_xz,
                     float scaled_max_force, double force_thresh, nh_mid_rec &R, v2 &out_vel)
{
    mid_thread_a(P, uid, scaled_max_force, R);
    return mid_thread_b(P, uid, NB, coh_xz, scaled_max_force, force_thresh, R, out_vel);
}

