// state_kernels.hip -- more of entity_compute_update (movement.c:2303) on the device, SURVEY section 8(f4):
// the heading gate (:2319-2336), the arms of the state switch that k_state_update does not cover except
// STATE_SURROUND_ENTITY (:2423-2437, :2569-2668), adjacent_settled_count (:982) and the arrival overlay's settle rule
// (G_Arrival_ShouldSettle, arrival.c:946) with what it calls: N_SegmentWithinRegion (nav.c:4326) over
// M_Tile_LineSupercoverTilesSorted (tile.c:430), arrival_near_region / arrival_near_open_slot
// (arrival.c:158, :326).  Kernels AND their C entry points (include/navhip.h): this translation unit was
// added after the round's profiles were taken and touches no other (profiles/README.md, `files`); its
// device scratch is ONE staging slot of the context, carved up per call.
//
// Everything here is decision logic on a few hundred bytes per unit: HBM-latency bound, one row of 16
// lanes per unit where a loop can be shared (the slots of a zone, the pad ring of arrival_near_region), a
// thread per unit elsewhere.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include "navhip_internal.h"
#include "agent_internal.h"
#include "agent_math.h"

#define SK_SLOT      43          /* ctx->stage[] slot this file owns                                     */
#define SK_QUERY_R   30.0f       /* SEPARATION_NEIGHB_RADIUS, movement.c:428                            */
#define SK_QUERY_MAX 128         /* near_ents[128], movement.c:997                                      */

#define SKCHK(ctx, call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { \
    (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_); return NAVHIP_ERR_DEVICE; } } while(0)

// ---------------------------------------------------------------------------------------------
// the heading gate, a thread per unit
// ---------------------------------------------------------------------------------------------
// PFM_Quat_PitchDiff (pf_math.c:677) turns (1, 0, 0) by both quaternions and takes atan2(det, dot) of the
// x/z parts: the turned front of q is (1 - 2y^2 - 2z^2, 2xz + 2wy) (PFM_Mat4x4_RotFromQuat :324, column 0).
// dir_quat_from_velocity (movement.c:1411) is the rotation about Y by atan2(v.z, v.x) - pi/2, whose turned
// front is (cos a, sin a) = (v.z, -v.x) / |v|.  |angle| > tol  <=>  cos(angle) < cos(tol): no trigonometry,
// and the comparison in double with a margin that is a hundred times the float path's error.
__global__ __launch_bounds__(256) void k_heading_gate(nh_step_params P, int begin, int end, const float *pos_xz, const float *vel_xz,
                                                      const uint8_t *state, const float *radius, const uint32_t *flags,
                                                      navhip_gate_in in, float *out_vel, float *out_new_pos, uint8_t *out_gate)
{
    const int i = begin + blockIdx.x * 256 + threadIdx.x;
    if(i >= end) return;
    v2 nv = mkv(in.new_vel_xz[2 * i], in.new_vel_xz[2 * i + 1]);
    const int st = state[i];
    uint8_t gate = 0;
    const bool gated_state = st == NAVHIP_STATE_MOVING || st == NAVHIP_STATE_SEEK_ENEMIES
                          || st == NAVHIP_STATE_SURROUND_ENTITY || st == NAVHIP_STATE_ENTER_ENTITY_RANGE;
    if(gated_state && vlen(nv) > CP_EPS) {
        const v2 vdes = mkv(in.vdes_xz[2 * i], in.vdes_xz[2 * i + 1]);
        const v2 h = vlen(vdes) > CP_EPS ? vdes : nv;                       // intended_heading, :2286
        const double qx = in.next_rot[4 * i], qy = in.next_rot[4 * i + 1], qz = in.next_rot[4 * i + 2],
                     qw = in.next_rot[4 * i + 3];
        const double d1x = 1.0 - 2.0 * qy * qy - 2.0 * qz * qz, d1z = 2.0 * qx * qz + 2.0 * qw * qy;
        const double d2x = (double)h.z, d2z = -(double)h.x;
        const double l1 = sqrt(d1x * d1x + d1z * d1z), l2 = sqrt(d2x * d2x + d2z * d2z);
        const bool rolling = vlen(mkv(vel_xz[2 * i], vel_xz[2 * i + 1])) > CP_EPS;
        // cos(90 deg), cos(10 deg)
        const double cos_tol = rolling ? 0.0 : 0.98480775301220805936674302458952;
        if(!(l1 * l2 > 1e-9)) {
            gate = NAVHIP_GATE_HOST;                                        // not a yaw: the host's float path decides
        }else{
            const double c = (d1x * d2x + d1z * d2z) / (l1 * l2);
            if(fabs(c - cos_tol) < 1e-4) gate = NAVHIP_GATE_HOST;
            else if(c < cos_tol) { gate = NAVHIP_GATE_TURN; nv = mkv(0.0f, 0.0f); }
        }
    }
    out_vel[2 * i] = nv.x; out_vel[2 * i + 1] = nv.z;
    const v2 pos = mkv(pos_xz[2 * i], pos_xz[2 * i + 1]);
    v2 np = mkv(pos.x + nv.x, pos.z + nv.z);                                // new_pos_for_vel, :1820
    if(in.interp_from_xz && P.hz < 20 && !(flags[i] & NAVHIP_ENTITY_FLAG_GARRISONED)) {
        // a rate below 20 Hz: the position the rest of entity_compute_update tests is the first interpolated position
        // of an accepted move (:2356-2377), pos + vel of a refused one
        const int layer = nav_layer_for(flags[i], radius[i]);
        if(!P.map.layers[layer].cost) gate |= NAVHIP_GATE_HOST;
        else if(vlen(nv) > 0.0f && pos_pathable(P, layer, np.x, np.z)
             && (pos_blocked(P, layer, pos.x, pos.z) || !pos_blocked(P, layer, np.x, np.z))) {
            const float fr = in.interp_step[i];
            if(!(fabs(1.0 - (double)fr) < (double)(1.0f / 1024.0f))) {      // interpolate_positions, :2222
                const float fx = in.interp_from_xz[2 * i], fz = in.interp_from_xz[2 * i + 1];
                float dx = np.x - fx, dz = np.z - fz;
                dx = dx * fr; dz = dz * fr;
                np = mkv(fx + dx, fz + dz);
            }
        }
    }
    out_new_pos[2 * i] = np.x; out_new_pos[2 * i + 1] = np.z;
    out_gate[i] = gate;
}

// ---------------------------------------------------------------------------------------------
// the arms that flags, a counter, an angle or a distance decide (formation members on the move, ARRIVING_TO_CELL, the
// wait timer, the end of TURNING, ENTER_ENTITY_RANGE): a thread per unit, after k_state_update
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_state_aux(nh_step_params P, const float *pos_xz, const float *radius, const uint32_t *flags,
                                                   const uint8_t *state, navhip_state_aux_in in, uint8_t *io_state,
                                                   uint8_t *io_flags, int32_t *out_ticks)
{
    const int i = P.work_begin + blockIdx.x * 256 + threadIdx.x;
    if(i >= P.work_end) return;
    const int st = state[i];
    int ticks = in.wait_ticks_left[i];
    const uint32_t ef = flags[i];
    const int layer = nav_layer_for(ef, radius[i]);
    const uint8_t *cost = P.map.layers[layer].cost;
    const bool ours = st == NAVHIP_STATE_WAITING || st == NAVHIP_STATE_ARRIVING_TO_CELL || (st == NAVHIP_STATE_TURNING && in.ent_rot)
                   || (st == NAVHIP_STATE_ENTER_ENTITY_RANGE && in.range_target && in.range_target[i] >= -1)
                   || (st == NAVHIP_STATE_SURROUND_ENTITY && in.surround_target && in.surround_target[i] >= -1 && P.hz == 20)
                   || ((st == NAVHIP_STATE_MOVING || st == NAVHIP_STATE_MOVING_IN_FORMATION) && (in.fstate[i] & NAVHIP_FS_MEMBER));
    if(ours && !(ef & NAVHIP_ENTITY_FLAG_GARRISONED) && cost) {                 // (:2344 returns before everything)
        const uint8_t fs = in.fstate[i];
        tiledesc t;
        const bool pathable = tile_for_point(P, in.new_pos_xz[2 * i], in.new_pos_xz[2 * i + 1], t)
                           && cost[tile_index(P, t)] != NAVHIP_COST_IMPASSABLE;   // :2437
        uint8_t next = (uint8_t)st, fl = 0;
        bool decided = true;
        if(pathable) {
            if(st == NAVHIP_STATE_WAITING) {                                    // :2630-2644
                ticks--;
                if(ticks == 0) { next = in.wait_prev[i]; fl = NAVHIP_SU_SET_MOVING; }
            }else if(st == NAVHIP_STATE_ENTER_ENTITY_RANGE) {                   // :2569-2604
                const int tgt = in.range_target[i];
                if(tgt < 0) { next = NAVHIP_STATE_ARRIVED; fl = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK; }
                else{
                    const v2 np = mkv(in.new_pos_xz[2 * i], in.new_pos_xz[2 * i + 1]);
                    const v2 tp = mkv(pos_xz[2 * tgt], pos_xz[2 * tgt + 1]);
                    bool stop = vlen(vsub(np, tp)) <= in.target_range[i];
                    if(!stop) {
                        // N_IsAdjacentToImpassable (nav.c:4747): a 4-neighbour tile that is impassable or blocked ...
                        const nh_layer_view &L = P.map.layers[layer];
                        const int ar = t.chunk_r * 64 + t.tile_r, ac = t.chunk_c * 64 + t.tile_c;
                        const int dr[4] = {-1, 0, 0, 1}, dc[4] = {0, -1, 1, 0};
                        bool adj = false;
#pragma unroll
                        for(int k = 0; k < 4; k++) {
                            const int r = ar + dr[k], c = ac + dc[k];
                            if(r < 0 || c < 0 || r >= P.map.h * 64 || c >= P.map.w * 64) continue;
                            tiledesc a;
                            a.chunk_r = r >> 6; a.chunk_c = c >> 6; a.tile_r = r & 63; a.tile_c = c & 63;
                            const size_t idx = tile_index(P, a);
                            adj = adj || cost[idx] == NAVHIP_COST_IMPASSABLE || (L.blockers && L.blockers[idx] > 0);
                        }
                        // ... and N_IsMaximallyClose(new_pos, target, 0.0f) (nav.c:4707): the position IS the centre of one
                        // of the target's closest island tiles
                        if(adj) {
                            const int row = in.range_tiles_row[i];
                            for(int k = in.range_tiles_off[row]; k < in.range_tiles_off[row + 1] && !stop; k++) {
                                const float cx = P.map_x - (float)in.range_tiles[2 * k + 1] * 4.0f;
                                const float cz = P.map_z + (float)in.range_tiles[2 * k] * 4.0f;
                                stop = vlen(vsub(mkv(cx, cz), np)) <= 0.0f;
                            }
                        }
                    }
                    if(stop) { next = NAVHIP_STATE_WAITING; fl = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK; }
                    else if(vlen(vsub(tp, mkv(in.target_prev_xz[2 * i], in.target_prev_xz[2 * i + 1]))) > 5.0f)
                        fl = NAVHIP_SU_SET_DEST;
                }
            }else if(st == NAVHIP_STATE_SURROUND_ENTITY) {                      // :2509-2567
                const int tgt = in.surround_target[i];
                const uint8_t sq = in.surround_query[i];
                const int flock = P.flock[i];
                if(tgt < 0 || (sq & NAVHIP_SQ_ADJACENT)) { next = NAVHIP_STATE_ARRIVED; fl = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK; }
                else if(flock < 0) decided = false;                             // (the reference asserts a flock)
                else{
                    const v2 np = mkv(in.new_pos_xz[2 * i], in.new_pos_xz[2 * i + 1]);
                    const v2 me = mkv(pos_xz[2 * i], pos_xz[2 * i + 1]);
                    const v2 tp = mkv(pos_xz[2 * tgt], pos_xz[2 * tgt + 1]);
                    v2 dest = mkv(in.surround_nearest_prev_xz[2 * i], in.surround_nearest_prev_xz[2 * i + 1]);
                    const v2 delta = vsub(tp, mkv(in.surround_target_prev_xz[2 * i], in.surround_target_prev_xz[2 * i + 1]));
                    bool gone = false;
                    if(vlen(delta) > CP_EPS || vlen(mkv(P.vel_xz[2 * i], P.vel_xz[2 * i + 1])) < CP_EPS) {
                        // the host asked M_NavClosestReachableAdjacentPosFrom from both positions this tick can test:
                        // pos + new velocity [0] and pos [1] (the gate halted the unit; equal when the velocity is zero)
                        const int c = (np.x == me.x && np.z == me.z) ? 1 : 0;
                        if(!(sq & (NAVHIP_SQ_HAS_DEST_0 << c))) gone = true;
                        else dest = mkv(in.surround_dest_xz[4 * i + 2 * c], in.surround_dest_xz[4 * i + 2 * c + 1]);
                    }
                    if(gone) { next = NAVHIP_STATE_ARRIVED; fl = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK; }
                    else{
                        in.out_surround_dest_xz[2 * i] = dest.x; in.out_surround_dest_xz[2 * i + 1] = dest.z;
                        fl = NAVHIP_SU_SURROUND_PREV;
                        const v2 diff = vsub(mkv(P.flock_target_xz[2 * flock], P.flock_target_xz[2 * flock + 1]), dest);
                        if(vlen(diff) > CP_EPS) { fl |= NAVHIP_SU_SURROUND_DEST | NAVHIP_SU_SET_STATE; next = NAVHIP_STATE_SURROUND_ENTITY; }
                        else if(vlen(mkv(in.vdes_xz[2 * i], in.vdes_xz[2 * i + 1])) < CP_EPS) {
                            fl |= NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK; next = NAVHIP_STATE_WAITING;
                        }
                    }
                }
            }else if(st == NAVHIP_STATE_TURNING) {                              // :2606-2628
                // |PFM_Quat_PitchDiff(rot, target_dir)| <= 5 degrees, as the heading gate compares: the turned fronts of
                // both quaternions, the cosine in double, a margin around cos(5 deg)
                const float *a = in.ent_rot + 4 * i, *b = in.target_dir + 4 * i;
                const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
                const double d1x = 1.0 - 2.0 * ay * ay - 2.0 * az * az, d1z = 2.0 * ax * az + 2.0 * aw * ay;
                const double d2x = 1.0 - 2.0 * by * by - 2.0 * bz * bz, d2z = 2.0 * bx * bz + 2.0 * bw * by;
                const double l = sqrt(d1x * d1x + d1z * d1z) * sqrt(d2x * d2x + d2z * d2z);
                const double cos5 = 0.99619469809174553229501040247389;
                if(!(l > 1e-9)) decided = false;
                else{
                    const double c = (d1x * d2x + d1z * d2z) / l;
                    // (the cosine is flat at 5 degrees: a margin of 1e-5 is 0.007 degrees, and still 250 times the
                    // float path's error there)
                    if(fabs(c - cos5) < 1e-5) decided = false;                  // the host's float path decides
                    else if(c > cos5) { next = NAVHIP_STATE_ARRIVED; fl = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK; }
                }
            }else if(st == NAVHIP_STATE_ARRIVING_TO_CELL) {                     // :2645-2668
                if(!(fs & NAVHIP_FS_MEMBER)) { next = NAVHIP_STATE_MOVING; fl = NAVHIP_SU_SET_STATE; }
                else if(!(fs & NAVHIP_FS_READY)) { }
                else if(!(fs & NAVHIP_FS_IN_RANGE)) { next = NAVHIP_STATE_MOVING_IN_FORMATION; fl = NAVHIP_SU_SET_STATE; }
                else if(fs & NAVHIP_FS_ARRIVED) { next = NAVHIP_STATE_TURNING; fl = NAVHIP_SU_SET_STATE | NAVHIP_SU_TARGET_DIR; }
            }else{                                                              // a formation member on the move, :2423-2437
                if(!(fs & NAVHIP_FS_READY)) { }
                else if((fs & NAVHIP_FS_ASSIGNED) && (fs & NAVHIP_FS_IN_RANGE)) { next = NAVHIP_STATE_ARRIVING_TO_CELL; fl = NAVHIP_SU_SET_STATE; }
                else decided = false;                                           // falls through to the arrival arm: k_state_update's answer
            }
        }
        if(decided) { io_state[i] = next; io_flags[i] = fl; }
    }
    out_ticks[i] = ticks;
}

// the resident pass: move_work_out.ent_des_v of a unit = the direction its host gave the step, or -- where that was NaN:
// "sample on the device" -- the direction the step sampled (what move_hip.c's scatter leaves in the work item)
__global__ __launch_bounds__(256) void k_pick_vdes(int begin, int end, const float *given, const float *sampled, float *out)
{
    const int i = begin + blockIdx.x * 256 + threadIdx.x;
    if(i >= end) return;
    float x = given ? given[2 * i] : __int_as_float(0x7fc00000), z = given ? given[2 * i + 1] : 0.0f;
    if(x != x) { x = sampled ? sampled[2 * i] : 0.0f; z = sampled ? sampled[2 * i + 1] : 0.0f; }
    out[2 * i] = x; out[2 * i + 1] = z;
}

// the fused pass (navhip_state_pass): a unit whose gate is the host's to decide is the host's altogether
__global__ __launch_bounds__(256) void k_gate_host_rows(int begin, int end, const uint8_t *gate, const uint8_t *state,
                                                        const int32_t *in_ticks, uint8_t *io_state, uint8_t *io_flags,
                                                        int32_t *out_ticks)
{
    const int i = begin + blockIdx.x * 256 + threadIdx.x;
    if(i >= end || !(gate[i] & NAVHIP_GATE_HOST)) return;
    io_state[i] = state[i]; io_flags[i] = NAVHIP_SU_HOST;
    if(out_ticks) out_ticks[i] = in_ticks[i];
}

// sparse rows of the three arms with per-unit arrays (navhip_state_aux_in.sparse_units): row k of every given array
// goes to row units[k] of its dense twin, which the arm kernels read as before; a thread per listed unit
struct sk_sparse_rows {
    int n; const int32_t *units;
    const float *er, *td; float *d_er, *d_td;
    const int32_t *rt; const float *rr, *rp; const int32_t *row; int32_t *d_rt; float *d_rr, *d_rp; int32_t *d_row;
    const int32_t *stgt; const uint8_t *sq; const float *stp, *snp, *sd; int32_t *d_stgt; uint8_t *d_sq; float *d_stp, *d_snp, *d_sd;
};
__global__ __launch_bounds__(256) void k_sparse_rows(sk_sparse_rows S)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if(k >= S.n) return;
    const int i = S.units[k];
    if(S.er) {
        for(int c = 0; c < 4; c++) { S.d_er[4 * i + c] = S.er[4 * k + c]; S.d_td[4 * i + c] = S.td[4 * k + c]; }
    }
    if(S.rt) {
        S.d_rt[i] = S.rt[k]; S.d_rr[i] = S.rr[k]; S.d_row[i] = S.row[k];
        S.d_rp[2 * i] = S.rp[2 * k]; S.d_rp[2 * i + 1] = S.rp[2 * k + 1];
    }
    if(S.stgt) {
        S.d_stgt[i] = S.stgt[k]; S.d_sq[i] = S.sq[k];
        for(int c = 0; c < 2; c++) { S.d_stp[2 * i + c] = S.stp[2 * k + c]; S.d_snp[2 * i + c] = S.snp[2 * k + c]; }
        for(int c = 0; c < 4; c++) S.d_sd[4 * i + c] = S.sd[4 * k + c];
    }
}
__global__ __launch_bounds__(256) void k_sparse_gather2(int n, const int32_t *units, const float *dense, float *rows)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if(k >= n) return;
    rows[2 * k] = dense[2 * units[k]]; rows[2 * k + 1] = dense[2 * units[k] + 1];
}

// ---------------------------------------------------------------------------------------------
// adjacent_settled_count: the ids of the spatial query (k_spatial_query, the reference's visiting order,
// capped) -> the count, a thread per unit
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_query(int nq, const int32_t *uids, const float *pos_xz, float *query)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if(q >= nq) return;
    query[2 * q] = pos_xz[2 * uids[q]]; query[2 * q + 1] = pos_xz[2 * uids[q] + 1];
}

// a row of 16 lanes per unit: the candidates of a query are independent tests (a count, no order) -- a thread per unit
// walked its 128 ids one dependent gather after the other (167 us for 3 400 units)
__global__ __launch_bounds__(256) void k_settled_count(int nq, const int32_t *uids, const float *pos_xz,
                                                       const float *radius, const uint32_t *flags,
                                                       const uint8_t *state, const int32_t *q_counts,
                                                       const uint32_t *q_ids, int32_t *out)
{
    const int q = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15;
    const bool live = q < nq;
    const int uid = live ? uids[q] : 0;
    const float r_uid = radius[uid];
    const bool wide = 2.0f * r_uid + 5.0f > SK_QUERY_R;                     // search_radius, :995
    const v2 pos = mkv(pos_xz[2 * uid], pos_xz[2 * uid + 1]);
    const uint32_t my_air = flags[uid] & NAVHIP_ENTITY_FLAG_AIR;
    int count = 0;
    const uint32_t *ids = q_ids + (size_t)(live ? q : 0) * SK_QUERY_MAX;
    const int n = (live && !wide) ? q_counts[q] : 0;
    for(int k = gl; k < n; k += 16) {
        const int c = (int)ids[k];
        const uint32_t f = flags[c];
        if(f & NAVHIP_ENTITY_FLAG_GARRISONED) continue;                     // filter_garrisoned, position.c:384
        if(c == uid || !(f & NAVHIP_ENTITY_FLAG_MOVABLE) || (f & NAVHIP_ENTITY_FLAG_AIR) != my_air) continue;
        if(state[c] != NAVHIP_STATE_ARRIVED) continue;
        const v2 cp = mkv(pos_xz[2 * c], pos_xz[2 * c + 1]);
        if(vlen(vsub(pos, cp)) <= r_uid + radius[c] + 5.0f) count++;         // ADJACENCY_SEP_DIST
    }
#pragma unroll
    for(int d = 1; d < 16; d <<= 1) count += __shfl_xor(count, d);
    if(live && gl == 0) out[q] = wide ? -1 : count;
}

// ---------------------------------------------------------------------------------------------
// G_Arrival_ShouldSettle, a row of 16 lanes per unit
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sk_row_any(bool p)                          // over the 16 lanes of the unit's row
{
    // (wave64: the ballot is 64 bits, a row its 16-bit field at lane & 48 -- gfx950 only, like everything here)
    return ((__ballot(p) >> (threadIdx.x & 48)) & 0xffffull) != 0ull;
}

__device__ __forceinline__ uint64_t sk_key(const tiledesc &t)              // td_key, nav.c:207
{
    return ((uint64_t)t.chunk_r << 48) | ((uint64_t)t.chunk_c << 32) | ((uint64_t)t.tile_r << 16) | (uint64_t)t.tile_c;
}

__device__ bool sk_keys_contain(const uint64_t *keys, int num, uint64_t k)   // region_keys_contain, nav.c:4283
{
    int lo = 0, hi = num;
    while(lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        const uint64_t v = keys[mid];
        if(v == k) return true;
        if(v < k) lo = mid + 1; else hi = mid;
    }
    return false;
}

// N_SegmentWithinRegion (nav.c:4326): every tile of the supercover walk from a to b (tile.c:430, both ends
// inside the map, so the walk starts at a) is a key of the region.  The walk's floats are the reference's:
// PFM_Vec2_Normal of the direction, fabs() promoting the t_max quotients to double before they are stored in a
// float, M_Tile_Bounds as (map - chunk * 256) - tile * 4.  a == b gives NaN everywhere and one tile, as there.
__device__ bool sk_segment_within(const nh_step_params &P, v2 a, v2 b, const uint64_t *keys, int num)
{
    if(num == 0) return false;
    tiledesc ta, tb;
    if(!tile_for_point(P, a.x, a.z, ta) || !tile_for_point(P, b.x, b.z, tb)) return false;
    const int ar = ta.chunk_r * 64 + ta.tile_r, ac = ta.chunk_c * 64 + ta.tile_c;
    const int br = tb.chunk_r * 64 + tb.tile_r, bc = tb.chunk_c * 64 + tb.tile_c;
    const int maxout = abs(br - ar) + abs(bc - ac) + 2;
    v2 dir = mkv(b.x - a.x, b.z - a.z);
    const float len = vlen(dir);
    dir = mkv(dir.x / len, dir.z / len);
    const int step_c = dir.x <= 0.0f ? 1 : -1, step_r = dir.z >= 0.0f ? 1 : -1;
    const float t_delta_x = fabsf(4.0f / dir.x), t_delta_z = fabsf(4.0f / dir.z);
    const float bx = (P.map_x - (float)(ta.chunk_c * 256)) - (float)(ta.tile_c * 4);
    const float bz = (P.map_z + (float)(ta.chunk_r * 256)) + (float)(ta.tile_r * 4);
    float t_max_x = (float)((step_c > 0 ? fabs((double)(a.x - (bx - 4.0f))) : fabs((double)(a.x - bx))) / fabs((double)dir.x));
    float t_max_z = (float)((step_r > 0 ? fabs((double)(a.z - (bz + 4.0f))) : fabs((double)(a.z - bz))) / fabs((double)dir.z));
    int r = ar, c = ac;
    const int max_r = P.map.h * 64, max_c = P.map.w * 64;
    for(int n = 0; n < maxout; n++) {
        tiledesc t;
        t.chunk_r = r >> 6; t.chunk_c = c >> 6; t.tile_r = r & 63; t.tile_c = c & 63;
        if(!sk_keys_contain(keys, num, sk_key(t))) return false;
        int dc = 0, dr = 0;
        if(t_max_x < t_max_z) { t_max_x = t_max_x + t_delta_x; dc = step_c; }
        else                  { t_max_z = t_max_z + t_delta_z; dr = step_r; }
        if(r == br && c == bc) break;
        r += dr; c += dc;
        if(r < 0 || r >= max_r || c < 0 || c >= max_c) break;             // M_Tile_RelativeDesc
    }
    return true;
}

__device__ __forceinline__ bool sk_in_region(const nh_step_params &P, v2 p, const uint64_t *keys, int num)
{
    // arrival_in_region (arrival.c:153): the walk from p to p is the tile of p
    tiledesc t;
    if(num == 0 || !tile_for_point(P, p.x, p.z, t)) return false;
    return sk_keys_contain(keys, num, sk_key(t));
}

__global__ __launch_bounds__(256) void k_arrival_settle(nh_step_params P, const float *vel_xz, const float *radius_of,
                                                        navhip_settle_in in, navhip_settle_out out)
{
    const int q = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int gl = (int)(threadIdx.x & 15);
    if(q >= in.nq) return;
    const int uid = in.uid[q];
    const navhip_arrival_zone Z = in.zones[in.zone[q]];
    const uint64_t *keys = in.region_keys + Z.key_begin;
    const int num_region = Z.key_end - Z.key_begin;
    const v2 np = mkv(in.new_pos_xz[2 * q], in.new_pos_xz[2 * q + 1]);
    const v2 vel = mkv(vel_xz[2 * uid], vel_xz[2 * uid + 1]);
    const float radius = radius_of[uid];
    const int nsettled = in.nsettled[q];
    int substate = in.substate[q], stuck = in.stuck[q];
    bool anchored = in.progress_anchored[q] != 0;
    v2 anchor = mkv(in.progress_anchor_xz[2 * q], in.progress_anchor_xz[2 * q + 1]);
    const bool sink_valid = in.sink_valid[q] != 0;
    const v2 sink = mkv(in.sink_xz[2 * q], in.sink_xz[2 * q + 1]);

    const bool in_region = sk_in_region(P, np, keys, num_region);
    // arrival_near_region (arrival.c:158): the ring of pad tiles around the position, shared by the lanes
    bool near_region = in_region;
    if(!in_region) {
        int pad = (int)ceilf(Z.unit_radius / 4.0f);
        if(pad < 1) pad = 1;
        const int side = 2 * pad + 1;
        bool hit = false;
        for(int k = gl; k < side * side; k += 16) {
            const int dz = k / side - pad, dx = k % side - pad;
            if(dx == 0 && dz == 0) continue;
            hit = hit || sk_in_region(P, mkv(np.x + (float)dx * 4.0f, np.z + (float)dz * 4.0f), keys, num_region);
        }
        near_region = sk_row_any(hit);
    }
    // arrival_near_open_slot (arrival.c:326), tolerance radius * ARRIVAL_SINK_TOLERANCE: any slot of an active
    // row within it that no blocker stands on
    bool at_sink = false;
    if(in_region) {
        const float tol = radius * 1.5f, tol2 = tol * tol;
        bool hit = false;
        const nh_layer_view &L = P.map.layers[Z.layer];
        for(int k = Z.slot_begin + gl; k < Z.slot_end; k += 16) {
            if(in.slot_ring[k] > Z.active_row) continue;
            const v2 s = mkv(in.slots_xz[2 * k], in.slots_xz[2 * k + 1]);
            const v2 d = vsub(s, np);
            if(vdot(d, d) > tol2) continue;
            tiledesc t;
            bool blocked = false;
            if(L.blockers && tile_for_point(P, s.x, s.z, t)) blocked = L.blockers[tile_index(P, t)] > 0;
            hit = hit || !blocked;
        }
        at_sink = sk_row_any(hit);
    }
    int settle_contacts = 3;                                                // ARRIVAL_SETTLE_CONTACTS
    if(Z.fill_frac >= 0.90f) settle_contacts = 2;                          // (as the reference has it, :959-962)
    else if(Z.fill_frac >= 0.75f) settle_contacts = 3;
    const bool reachable_slot = sink_valid && sk_segment_within(P, np, sink, keys, num_region);
    const bool fill_done = Z.active_row >= Z.num_rows - 1;
    bool advancing = false;
    if(sink_valid) advancing = vdot(vel, vsub(sink, np)) > 0.0f;
    // unit_armed / unit_arm (arrival.c:96, :110): APPROACH 0, APPROACH_ARMED 1, SEEK 2, SEEK_ARMED 3
    bool armed = substate == 1 || substate == 3;
    if(!armed) {
        const v2 order = mkv(in.order_pos_xz[2 * q], in.order_pos_xz[2 * q + 1]);
        if(vlen(vsub(np, order)) > 4.0f) { substate += 1; armed = true; }    // ARRIVAL_ENGAGE_DIST
    }
    const bool by_prop = armed && !at_sink && in_region && nsettled >= settle_contacts
                      && (reachable_slot ? !advancing : fill_done);
    const float settle_range = ((float)Z.radius * 4.0f + radius) * 1.875f;  // ARRIVAL_SETTLE_RANGE
    const v2 to_centre = vsub(np, mkv(Z.centre_x, Z.centre_z));
    const bool within_settle_range = vdot(to_centre, to_centre) <= settle_range * settle_range;
    const bool stuck_eligible = nsettled >= 1 && (near_region || within_settle_range);
    if(!at_sink && stuck_eligible && armed) {
        if(!anchored) { anchor = np; anchored = true; stuck = 0; }
        if(vlen(vsub(np, anchor)) > 1.875f) { anchor = np; stuck = 0; }      // ARRIVAL_STUCK_DISP
        else stuck++;
    }
    const bool by_stuck = !at_sink && !by_prop && stuck_eligible && stuck >= 12;   // ARRIVAL_STUCK_LIMIT
    const bool by_contact = armed && !at_sink && near_region && !reachable_slot && nsettled >= 1
                         && Z.fill_frac >= 0.90f;
    if(gl == 0) {
        out.settle[q] = (at_sink || by_prop || by_stuck || by_contact) ? 1 : 0;
        out.substate[q] = (uint8_t)substate;
        out.progress_anchor_xz[2 * q] = anchor.x; out.progress_anchor_xz[2 * q + 1] = anchor.z;
        out.progress_anchored[q] = anchored ? 1 : 0;
        out.stuck[q] = stuck;
    }
}

// ---------------------------------------------------------------------------------------------
// C entry points
// ---------------------------------------------------------------------------------------------
namespace {

struct sk_arena {                     // one staging slot, carved up: offsets first, then the pointer
    size_t total = 0;
    size_t take(size_t bytes) { const size_t o = total; total += (bytes + 255) & ~(size_t)255; return o; }
};

void sk_map_view(const navhip_ctx *ctx, const navhip_world *w, nh_step_params *P)
{
    memset(P, 0, sizeof(*P));
    P->map.w = ctx->w; P->map.h = ctx->h;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        const navhip_layer &L = ctx->layers[l];
        P->map.layers[l] = nh_layer_view{L.cost, L.blockers, L.local_islands, L.factions,
                                         L.passmask, L.unit_cost, L.changed, L.islands, L.probemask};
    }
    P->map_x = w->map_pos_x; P->map_z = w->map_pos_z;
    P->n_ents = w->n_ents; P->hz = w->hz;
}

// the surround inputs come together, with the snapshot arrays the arm reads
bool sk_surround_inputs_ok(const navhip_world *w, const navhip_state_aux_in *in)
{
    const int given = (in->surround_target != nullptr) + (in->surround_query != nullptr) + (in->surround_target_prev_xz != nullptr)
                    + (in->surround_nearest_prev_xz != nullptr) + (in->surround_dest_xz != nullptr) + (in->vdes_xz != nullptr)
                    + (in->out_surround_dest_xz != nullptr);
    return given == 0 || (given == 7 && w->pos_xz && w->vel_xz && w->flock && (w->n_flocks == 0 || w->flock_target_xz));
}

// the six enter-range inputs come together, with the positions of the snapshot
bool sk_range_inputs_ok(const navhip_world *w, const navhip_state_aux_in *in)
{
    const int given = (in->range_target != nullptr) + (in->target_range != nullptr) + (in->target_prev_xz != nullptr)
                    + (in->range_tiles_row != nullptr) + (in->range_tiles_off != nullptr) + (in->range_tiles != nullptr);
    return given == 0 || (given == 6 && w->pos_xz && in->n_range_rows >= 0);
}

// the interpolated position of a rate below 20 Hz: both arrays, the movement rate, and what names the unit's nav layer
bool sk_interp_inputs_ok(const navhip_world *w, const navhip_gate_in *in)
{
    if((in->interp_from_xz != nullptr) != (in->interp_step != nullptr)) return false;
    if(w->hz != 20 && w->hz != 10 && w->hz != 5 && w->hz != 1 && w->hz != 0) return false;
    return !in->interp_from_xz || (w->radius && w->flags);
}

// sparse rows are navhip_state_pass_resident's
bool sk_no_sparse(const navhip_state_aux_in *in) { return in->sparse_units == nullptr && in->n_sparse == 0; }

bool sk_work_range(const navhip_world *w, int *b, int *e)
{
    *b = w->work_begin; *e = w->work_end;
    if(*b == 0 && *e == 0) *e = w->n_ents;
    return *b >= 0 && *e <= w->n_ents && *b <= *e;
}

}   // namespace

extern "C" {

int navhip_heading_gate_dev(navhip_ctx *ctx, const navhip_world *w, const navhip_gate_in *in, float *out_vel,
                            float *out_new_pos, uint8_t *out_gate, void *stream)
{
    if(!ctx || !w || !in || !out_vel || !out_new_pos || !out_gate || w->n_ents < 0) return NAVHIP_ERR_INVALID;
    if(w->n_ents == 0) return NAVHIP_OK;
    if(!w->pos_xz || !w->vel_xz || !w->state || !in->next_rot || !in->new_vel_xz || !in->vdes_xz) return NAVHIP_ERR_INVALID;
    if(!sk_interp_inputs_ok(w, in)) return NAVHIP_ERR_INVALID;
    int b, e;
    if(!sk_work_range(w, &b, &e)) return NAVHIP_ERR_INVALID;
    SKCHK(ctx, hipSetDevice(ctx->device));
    if(in->interp_from_xz) {
        // (the accept test of the interpolated position probes the derived row masks: rebuilt here if a plane changed)
        int rc = nh_refresh_derived(ctx, ctx->stream);
        if(rc) return rc;
    }
    nh_step_params P;
    sk_map_view(ctx, w, &P);
    if(e > b)
        hipLaunchKernelGGL(k_heading_gate, dim3((e - b + 255) / 256), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream,
                           P, b, e, w->pos_xz, w->vel_xz, w->state, w->radius, w->flags, *in, out_vel, out_new_pos, out_gate);
    SKCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_heading_gate(navhip_ctx *ctx, const navhip_world *w, const navhip_gate_in *in, float *out_vel,
                        float *out_new_pos, uint8_t *out_gate)
{
    if(!ctx || !w || !in || !out_vel || !out_new_pos || !out_gate || w->n_ents < 0) return NAVHIP_ERR_INVALID;
    if(w->n_ents == 0) return NAVHIP_OK;
    if(!w->pos_xz || !w->vel_xz || !w->state || !in->next_rot || !in->new_vel_xz || !in->vdes_xz) return NAVHIP_ERR_INVALID;
    int b, e;
    if(!sk_work_range(w, &b, &e)) return NAVHIP_ERR_INVALID;
    SKCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)w->n_ents;
    if(!sk_interp_inputs_ok(w, in)) return NAVHIP_ERR_INVALID;
    const bool ip = in->interp_from_xz != nullptr;
    sk_arena A;
    const size_t o_pos = A.take(n * 8), o_vel = A.take(n * 8), o_state = A.take(n), o_rot = A.take(n * 16),
                 o_nv = A.take(n * 8), o_vd = A.take(n * 8), o_ov = A.take(n * 8), o_op = A.take(n * 8), o_og = A.take(n),
                 o_if = A.take(ip ? n * 8 : 0), o_is = A.take(ip ? n * 4 : 0), o_rad = A.take(ip ? n * 4 : 0), o_flg = A.take(ip ? n * 4 : 0);
    char *base;
    int rc = navhip_stage_reserve(ctx, SK_SLOT, A.total, (void**)&base);
    if(rc) return rc;
    SKCHK(ctx, hipMemcpyAsync(base + o_pos, w->pos_xz, n * 8, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_vel, w->vel_xz, n * 8, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_state, w->state, n, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_rot, in->next_rot, n * 16, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_nv, in->new_vel_xz, n * 8, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_vd, in->vdes_xz, n * 8, hipMemcpyHostToDevice, s));
    navhip_world d = *w;
    d.pos_xz = (const float*)(base + o_pos); d.vel_xz = (const float*)(base + o_vel); d.state = (const uint8_t*)(base + o_state);
    navhip_gate_in di = {(const float*)(base + o_rot), (const float*)(base + o_nv), (const float*)(base + o_vd), nullptr, nullptr};
    if(ip) {
        SKCHK(ctx, hipMemcpyAsync(base + o_if, in->interp_from_xz, n * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_is, in->interp_step, n * 4, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_rad, w->radius, n * 4, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_flg, w->flags, n * 4, hipMemcpyHostToDevice, s));
        di.interp_from_xz = (const float*)(base + o_if); di.interp_step = (const float*)(base + o_is);
        d.radius = (const float*)(base + o_rad); d.flags = (const uint32_t*)(base + o_flg);
    }
    rc = navhip_heading_gate_dev(ctx, &d, &di, (float*)(base + o_ov), (float*)(base + o_op), (uint8_t*)(base + o_og), s);
    if(rc) return rc;
    if(e > b) {
        const size_t lo = (size_t)b, cnt = (size_t)(e - b);
        SKCHK(ctx, hipMemcpyAsync(out_vel + 2 * lo, base + o_ov + 8 * lo, cnt * 8, hipMemcpyDeviceToHost, s));
        SKCHK(ctx, hipMemcpyAsync(out_new_pos + 2 * lo, base + o_op + 8 * lo, cnt * 8, hipMemcpyDeviceToHost, s));
        SKCHK(ctx, hipMemcpyAsync(out_gate + lo, base + o_og + lo, cnt, hipMemcpyDeviceToHost, s));
    }
    SKCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

int navhip_state_update_aux_dev(navhip_ctx *ctx, const navhip_world *w, const navhip_state_aux_in *in, uint8_t *io_state,
                                uint8_t *io_flags, int32_t *out_ticks, void *stream)
{
    if(!ctx || !w || !in || !io_state || !io_flags || !out_ticks || w->n_ents < 0) return NAVHIP_ERR_INVALID;
    if(w->n_ents == 0) return NAVHIP_OK;
    if(!w->radius || !w->flags || !w->state || !in->fstate || !in->wait_ticks_left || !in->wait_prev || !in->new_pos_xz
    || (in->ent_rot != nullptr) != (in->target_dir != nullptr) || !sk_range_inputs_ok(w, in) || !sk_surround_inputs_ok(w, in) || !sk_no_sparse(in))
        return NAVHIP_ERR_INVALID;
    int b, e;
    if(!sk_work_range(w, &b, &e)) return NAVHIP_ERR_INVALID;
    SKCHK(ctx, hipSetDevice(ctx->device));
    nh_step_params P;
    sk_map_view(ctx, w, &P);
    P.work_begin = b; P.work_end = e;
    if(in->surround_target) { P.flock = w->flock; P.flock_target_xz = w->flock_target_xz; P.vel_xz = w->vel_xz; P.n_flocks = w->n_flocks; }
    if(e > b)
        hipLaunchKernelGGL(k_state_aux, dim3((e - b + 255) / 256), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream,
                           P, w->pos_xz, w->radius, w->flags, w->state, *in, io_state, io_flags, out_ticks);
    SKCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

int navhip_state_update_aux(navhip_ctx *ctx, const navhip_world *w, const navhip_state_aux_in *in, uint8_t *io_state,
                            uint8_t *io_flags, int32_t *out_ticks)
{
    if(!ctx || !w || !in || !io_state || !io_flags || !out_ticks || w->n_ents < 0) return NAVHIP_ERR_INVALID;
    if(w->n_ents == 0) return NAVHIP_OK;
    if(!w->radius || !w->flags || !w->state || !in->fstate || !in->wait_ticks_left || !in->wait_prev || !in->new_pos_xz
    || (in->ent_rot != nullptr) != (in->target_dir != nullptr) || !sk_range_inputs_ok(w, in) || !sk_surround_inputs_ok(w, in) || !sk_no_sparse(in))
        return NAVHIP_ERR_INVALID;
    int b, e;
    if(!sk_work_range(w, &b, &e)) return NAVHIP_ERR_INVALID;
    const size_t n = (size_t)w->n_ents;
    const bool su = in->surround_target != nullptr;
    if(su) {
        for(size_t i = (size_t)b; i < (size_t)e; i++) {
            if(in->surround_target[i] >= w->n_ents) return NAVHIP_ERR_INVALID;
            if(w->state[i] == NAVHIP_STATE_SURROUND_ENTITY && in->surround_target[i] >= -1 && w->flock[i] >= w->n_flocks) return NAVHIP_ERR_INVALID;
        }
    }
    size_t n_rt = 0;
    if(in->range_target) {
        const int rows = in->n_range_rows;
        for(int r = 0; r < rows; r++)
            if(in->range_tiles_off[r] < 0 || in->range_tiles_off[r + 1] < in->range_tiles_off[r]) return NAVHIP_ERR_INVALID;
        n_rt = rows ? (size_t)in->range_tiles_off[rows] : 0;
        for(size_t i = (size_t)b; i < (size_t)e; i++) {
            if(in->range_target[i] >= w->n_ents) return NAVHIP_ERR_INVALID;
            if(in->range_target[i] >= 0 && w->state[i] == NAVHIP_STATE_ENTER_ENTITY_RANGE
            && (in->range_tiles_row[i] < 0 || in->range_tiles_row[i] >= rows)) return NAVHIP_ERR_INVALID;
        }
    }
    SKCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    sk_arena A;
    const size_t o_rad = A.take(n * 4), o_fl = A.take(n * 4), o_st = A.take(n), o_fs = A.take(n), o_wt = A.take(n * 4),
                 o_wp = A.take(n), o_np = A.take(n * 8), o_ios = A.take(n), o_iof = A.take(n), o_ot = A.take(n * 4),
                 o_er = A.take(in->ent_rot ? n * 16 : 0), o_td = A.take(in->ent_rot ? n * 16 : 0);
    const bool rg = in->range_target != nullptr;
    const size_t rows = rg ? (size_t)in->n_range_rows : 0;
    const size_t o_pos = A.take(rg ? n * 8 : 0), o_rt = A.take(rg ? n * 4 : 0), o_rr = A.take(rg ? n * 4 : 0),
                 o_rp = A.take(rg ? n * 8 : 0), o_row = A.take(rg ? n * 4 : 0), o_off = A.take(rg ? (rows + 1) * 4 : 0),
                 o_til = A.take(rg ? n_rt * 4 + 4 : 0);
    const size_t F = (size_t)(w->n_flocks > 0 ? w->n_flocks : 0);
    const size_t o_spos = A.take(su && !rg ? n * 8 : 0), o_svel = A.take(su ? n * 8 : 0), o_sflock = A.take(su ? n * 4 : 0),
                 o_sftgt = A.take(su ? F * 8 + 8 : 0), o_stgt = A.take(su ? n * 4 : 0), o_sq = A.take(su ? n : 0),
                 o_stp = A.take(su ? n * 8 : 0), o_snp = A.take(su ? n * 8 : 0), o_sd = A.take(su ? n * 16 : 0),
                 o_svd = A.take(su ? n * 8 : 0), o_sout = A.take(su ? n * 8 : 0);
    char *base;
    int rc = navhip_stage_reserve(ctx, SK_SLOT, A.total, (void**)&base);
    if(rc) return rc;
    SKCHK(ctx, hipMemcpyAsync(base + o_rad, w->radius, n * 4, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_fl, w->flags, n * 4, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_st, w->state, n, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_fs, in->fstate, n, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_wt, in->wait_ticks_left, n * 4, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_wp, in->wait_prev, n, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_np, in->new_pos_xz, n * 8, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_ios, io_state, n, hipMemcpyHostToDevice, s));
    SKCHK(ctx, hipMemcpyAsync(base + o_iof, io_flags, n, hipMemcpyHostToDevice, s));
    navhip_world d = *w;
    d.radius = (const float*)(base + o_rad); d.flags = (const uint32_t*)(base + o_fl); d.state = (const uint8_t*)(base + o_st);
    d.pos_xz = nullptr;                          // (only the enter-range arm reads positions: staged below with its inputs)
    navhip_state_aux_in di;
    memset(&di, 0, sizeof(di));
    di.fstate = (const uint8_t*)(base + o_fs); di.wait_ticks_left = (const int32_t*)(base + o_wt);
    di.wait_prev = (const uint8_t*)(base + o_wp); di.new_pos_xz = (const float*)(base + o_np);
    if(rg) {
        SKCHK(ctx, hipMemcpyAsync(base + o_pos, w->pos_xz, n * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_rt, in->range_target, n * 4, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_rr, in->target_range, n * 4, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_rp, in->target_prev_xz, n * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_row, in->range_tiles_row, n * 4, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_off, in->range_tiles_off, (rows + 1) * 4, hipMemcpyHostToDevice, s));
        if(n_rt) SKCHK(ctx, hipMemcpyAsync(base + o_til, in->range_tiles, n_rt * 4, hipMemcpyHostToDevice, s));
        d.pos_xz = (const float*)(base + o_pos);
        di.range_target = (const int32_t*)(base + o_rt); di.target_range = (const float*)(base + o_rr);
        di.target_prev_xz = (const float*)(base + o_rp); di.range_tiles_row = (const int32_t*)(base + o_row);
        di.range_tiles_off = (const int32_t*)(base + o_off); di.range_tiles = (const int16_t*)(base + o_til);
        di.n_range_rows = in->n_range_rows;
    }
    if(in->ent_rot) {
        SKCHK(ctx, hipMemcpyAsync(base + o_er, in->ent_rot, n * 16, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_td, in->target_dir, n * 16, hipMemcpyHostToDevice, s));
        di.ent_rot = (const float*)(base + o_er); di.target_dir = (const float*)(base + o_td);
    }
    if(su) {
        if(!rg) { SKCHK(ctx, hipMemcpyAsync(base + o_spos, w->pos_xz, n * 8, hipMemcpyHostToDevice, s)); d.pos_xz = (const float*)(base + o_spos); }
        SKCHK(ctx, hipMemcpyAsync(base + o_svel, w->vel_xz, n * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_sflock, w->flock, n * 4, hipMemcpyHostToDevice, s));
        if(F) SKCHK(ctx, hipMemcpyAsync(base + o_sftgt, w->flock_target_xz, F * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_stgt, in->surround_target, n * 4, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_sq, in->surround_query, n, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_stp, in->surround_target_prev_xz, n * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_snp, in->surround_nearest_prev_xz, n * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_sd, in->surround_dest_xz, n * 16, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_svd, in->vdes_xz, n * 8, hipMemcpyHostToDevice, s));
        d.vel_xz = (const float*)(base + o_svel); d.flock = (const int32_t*)(base + o_sflock);
        d.flock_target_xz = (const float*)(base + o_sftgt);
        di.surround_target = (const int32_t*)(base + o_stgt); di.surround_query = (const uint8_t*)(base + o_sq);
        di.surround_target_prev_xz = (const float*)(base + o_stp); di.surround_nearest_prev_xz = (const float*)(base + o_snp);
        di.surround_dest_xz = (const float*)(base + o_sd); di.vdes_xz = (const float*)(base + o_svd);
        di.out_surround_dest_xz = (float*)(base + o_sout);
    }
    rc = navhip_state_update_aux_dev(ctx, &d, &di, (uint8_t*)(base + o_ios), (uint8_t*)(base + o_iof), (int32_t*)(base + o_ot), s);
    if(rc) return rc;
    if(e > b) {
        const size_t lo = (size_t)b, cnt = (size_t)(e - b);
        if(su) SKCHK(ctx, hipMemcpyAsync(in->out_surround_dest_xz + 2 * lo, base + o_sout + 8 * lo, cnt * 8, hipMemcpyDeviceToHost, s));
        SKCHK(ctx, hipMemcpyAsync(io_state + lo, base + o_ios + lo, cnt, hipMemcpyDeviceToHost, s));
        SKCHK(ctx, hipMemcpyAsync(io_flags + lo, base + o_iof + lo, cnt, hipMemcpyDeviceToHost, s));
        SKCHK(ctx, hipMemcpyAsync(out_ticks + lo, base + o_ot + 4 * lo, cnt * 4, hipMemcpyDeviceToHost, s));
    }
    SKCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

int navhip_state_pass(navhip_ctx *ctx, const navhip_world *w, const navhip_state_pass_in *in, const navhip_state_pass_out *out)
{
    if(!ctx || !w || !in || !out || w->n_ents < 0) return NAVHIP_ERR_INVALID;
    if(w->n_ents == 0) return NAVHIP_OK;
    const navhip_gate_in &G = in->gate;
    const navhip_state_in &T = in->state;
    const navhip_state_aux_in &X = in->aux;
    const bool aux = X.fstate != nullptr, turn = aux && X.ent_rot != nullptr, rg = aux && X.range_target != nullptr;
    const bool su = aux && X.surround_target != nullptr, ip = G.interp_from_xz != nullptr;
    const size_t n = (size_t)w->n_ents, F = (size_t)w->n_flocks;
    if(!w->pos_xz || !w->vel_xz || !w->radius || !w->flags || !w->state || !w->flock || !G.next_rot || !G.new_vel_xz || !G.vdes_xz
    || !out->state || !out->flags || !out->gate || !out->new_pos_xz
    || (F > 0 && (!w->flock_target_xz || !w->flock_offsets || !w->flock_members || !T.flock_layer || !T.flock_nearest_xz
                  || !T.flock_tiles_off || !T.flock_tiles))
    || (aux && (!X.wait_ticks_left || !X.wait_prev || !out->wait_ticks_left || (X.ent_rot != nullptr) != (X.target_dir != nullptr)
                || !sk_range_inputs_ok(w, &X) || !sk_no_sparse(&X)))
    || !sk_interp_inputs_ok(w, &G))
        return NAVHIP_ERR_INVALID;
    if(su) {
        // (the pass supplies vdes and the snapshot arrays itself: the caller gives the five query arrays and the output)
        if(!X.surround_query || !X.surround_target_prev_xz || !X.surround_nearest_prev_xz || !X.surround_dest_xz
        || !X.out_surround_dest_xz) return NAVHIP_ERR_INVALID;
    }
    int b, e;
    if(!sk_work_range(w, &b, &e)) return NAVHIP_ERR_INVALID;
    if(su) {
        for(size_t i = (size_t)b; i < (size_t)e; i++) {
            if(X.surround_target[i] >= w->n_ents) return NAVHIP_ERR_INVALID;
            if(w->state[i] == NAVHIP_STATE_SURROUND_ENTITY && X.surround_target[i] >= -1 && w->flock[i] >= w->n_flocks) return NAVHIP_ERR_INVALID;
        }
    }
    // (offsets that are not a CSR become a huge size_t below -- INVALID, not NOMEM)
    for(size_t f = 0; f < F; f++)
        if(w->flock_offsets[f] < 0 || w->flock_offsets[f + 1] < w->flock_offsets[f] || T.flock_tiles_off[f] < 0
        || T.flock_tiles_off[f + 1] < T.flock_tiles_off[f]) return NAVHIP_ERR_INVALID;
    if(F && (size_t)w->flock_offsets[F] > n) return NAVHIP_ERR_INVALID;
    const size_t nmembers = F ? (size_t)w->flock_offsets[F] : 0, ntiles = F ? (size_t)T.flock_tiles_off[F] : 0;
    size_t rows = 0, n_rt = 0;
    if(rg) {
        rows = (size_t)X.n_range_rows;
        for(size_t r = 0; r < rows; r++)
            if(X.range_tiles_off[r] < 0 || X.range_tiles_off[r + 1] < X.range_tiles_off[r]) return NAVHIP_ERR_INVALID;
        n_rt = rows ? (size_t)X.range_tiles_off[rows] : 0;
        for(size_t i = (size_t)b; i < (size_t)e; i++) {
            if(X.range_target[i] >= w->n_ents) return NAVHIP_ERR_INVALID;
            if(X.range_target[i] >= 0 && w->state[i] == NAVHIP_STATE_ENTER_ENTITY_RANGE
            && (X.range_tiles_row[i] < 0 || (size_t)X.range_tiles_row[i] >= rows)) return NAVHIP_ERR_INVALID;
        }
    }
    SKCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    sk_arena A;
    struct up { size_t off; const void *src; size_t bytes; };
    std::vector<up> ups;
    auto stage = [&](const void *src, size_t bytes) { const size_t o = A.take(bytes + 8); if(src && bytes) ups.push_back({o, src, bytes}); return o; };
    // the snapshot, once
    const size_t o_pos = stage(w->pos_xz, n * 8), o_vel = stage(w->vel_xz, n * 8), o_rad = stage(w->radius, n * 4),
                 o_flg = stage(w->flags, n * 4), o_st = stage(w->state, n), o_flock = stage(w->flock, n * 4),
                 o_ftgt = stage(w->flock_target_xz, F * 8), o_foff = stage(w->flock_offsets, (F + 1) * 4),
                 o_fmem = stage(w->flock_members, nmembers * 4);
    const size_t o_rot = stage(G.next_rot, n * 16), o_nv = stage(G.new_vel_xz, n * 8), o_vd = stage(G.vdes_xz, n * 8);
    const size_t o_skip = stage(T.skip, T.skip ? n : 0), o_flay = stage(T.flock_layer, F), o_fnear = stage(T.flock_nearest_xz, F * 8),
                 o_toff = stage(T.flock_tiles_off, (F + 1) * 4), o_tiles = stage(T.flock_tiles, ntiles * 4);
    const size_t o_fs = stage(X.fstate, aux ? n : 0), o_wt = stage(X.wait_ticks_left, aux ? n * 4 : 0), o_wp = stage(X.wait_prev, aux ? n : 0),
                 o_er = stage(X.ent_rot, turn ? n * 16 : 0), o_td = stage(X.target_dir, turn ? n * 16 : 0),
                 o_rt = stage(X.range_target, rg ? n * 4 : 0), o_rr = stage(X.target_range, rg ? n * 4 : 0),
                 o_rp = stage(X.target_prev_xz, rg ? n * 8 : 0), o_row = stage(X.range_tiles_row, rg ? n * 4 : 0),
                 o_roff = stage(X.range_tiles_off, rg ? (rows + 1) * 4 : 0), o_rtil = stage(X.range_tiles, rg ? n_rt * 4 : 0);
    const size_t o_if = stage(G.interp_from_xz, ip ? n * 8 : 0), o_is = stage(G.interp_step, ip ? n * 4 : 0);
    const size_t o_stgt = stage(X.surround_target, su ? n * 4 : 0), o_sq = stage(X.surround_query, su ? n : 0),
                 o_stp = stage(X.surround_target_prev_xz, su ? n * 8 : 0), o_snp = stage(X.surround_nearest_prev_xz, su ? n * 8 : 0),
                 o_sd = stage(X.surround_dest_xz, su ? n * 16 : 0);
    const size_t r_sd = A.take(su ? n * 8 : 0);
    // results
    const size_t r_vel = A.take(n * 8), r_np = A.take(n * 8), r_gate = A.take(n), r_st = A.take(n), r_fl = A.take(n), r_tk = A.take(n * 4);
    char *base;
    int rc = navhip_stage_reserve(ctx, SK_SLOT, A.total, (void**)&base);
    if(rc) return rc;
    for(const up &u : ups) SKCHK(ctx, hipMemcpyAsync(base + u.off, u.src, u.bytes, hipMemcpyHostToDevice, s));
    navhip_world d = *w;
    d.pos_xz = (const float*)(base + o_pos); d.vel_xz = (const float*)(base + o_vel); d.radius = (const float*)(base + o_rad);
    d.flags = (const uint32_t*)(base + o_flg); d.state = (const uint8_t*)(base + o_st); d.flock = (const int32_t*)(base + o_flock);
    d.flock_target_xz = (const float*)(base + o_ftgt); d.flock_offsets = (const int32_t*)(base + o_foff);
    d.flock_members = (const int32_t*)(base + o_fmem);
    // (nothing else of the world is read by the three passes; what is not staged must not travel as a host pointer)
    d.max_speed = d.speed = nullptr; d.has_dest_los = nullptr; d.vdes_xz = nullptr; d.flock_field_slot = nullptr; d.field_pool = nullptr;
    d.form_ready = nullptr; d.cell_pos_xz = d.form_cohesion_xz = d.form_align_xz = d.form_drag_xz = nullptr;
    d.arrival_sink_xz = nullptr; d.arrival_flags = nullptr; d.los_pool = nullptr; d.flock_los_slot = nullptr; d.los_pos_xz = nullptr;
    d.region_row = nullptr; d.region_field_slot = nullptr;
    navhip_gate_in dg = {(const float*)(base + o_rot), (const float*)(base + o_nv), (const float*)(base + o_vd),
                         ip ? (const float*)(base + o_if) : nullptr, ip ? (const float*)(base + o_is) : nullptr};
    rc = navhip_heading_gate_dev(ctx, &d, &dg, (float*)(base + r_vel), (float*)(base + r_np), (uint8_t*)(base + r_gate), s);
    if(rc) return rc;
    navhip_state_in ds = {(const float*)(base + r_np), (const float*)(base + o_vd), T.skip ? (const uint8_t*)(base + o_skip) : nullptr,
                          (const uint8_t*)(base + o_flay), (const float*)(base + o_fnear), (const int32_t*)(base + o_toff),
                          (const int16_t*)(base + o_tiles)};
    rc = navhip_state_update_dev(ctx, &d, &ds, (uint8_t*)(base + r_st), (uint8_t*)(base + r_fl), s);
    if(rc) return rc;
    if(aux) {
        navhip_state_aux_in da;
        memset(&da, 0, sizeof(da));
        da.fstate = (const uint8_t*)(base + o_fs); da.wait_ticks_left = (const int32_t*)(base + o_wt);
        da.wait_prev = (const uint8_t*)(base + o_wp); da.new_pos_xz = (const float*)(base + r_np);
        if(turn) { da.ent_rot = (const float*)(base + o_er); da.target_dir = (const float*)(base + o_td); }
        if(rg) {
            da.range_target = (const int32_t*)(base + o_rt); da.target_range = (const float*)(base + o_rr);
            da.target_prev_xz = (const float*)(base + o_rp); da.range_tiles_row = (const int32_t*)(base + o_row);
            da.range_tiles_off = (const int32_t*)(base + o_roff); da.range_tiles = (const int16_t*)(base + o_rtil);
            da.n_range_rows = X.n_range_rows;
        }
        if(su) {
            da.surround_target = (const int32_t*)(base + o_stgt); da.surround_query = (const uint8_t*)(base + o_sq);
            da.surround_target_prev_xz = (const float*)(base + o_stp); da.surround_nearest_prev_xz = (const float*)(base + o_snp);
            da.surround_dest_xz = (const float*)(base + o_sd); da.vdes_xz = (const float*)(base + o_vd);
            da.out_surround_dest_xz = (float*)(base + r_sd);
        }
        rc = navhip_state_update_aux_dev(ctx, &d, &da, (uint8_t*)(base + r_st), (uint8_t*)(base + r_fl), (int32_t*)(base + r_tk), s);
        if(rc) return rc;
    }
    if(e > b) {
        hipLaunchKernelGGL(k_gate_host_rows, dim3((e - b + 255) / 256), dim3(256), 0, s, b, e, (const uint8_t*)(base + r_gate),
                           (const uint8_t*)(base + o_st), aux ? (const int32_t*)(base + o_wt) : (const int32_t*)nullptr,
                           (uint8_t*)(base + r_st), (uint8_t*)(base + r_fl), aux ? (int32_t*)(base + r_tk) : (int32_t*)nullptr);
        SKCHK(ctx, hipGetLastError());
        const size_t lo = (size_t)b, cnt = (size_t)(e - b);
        SKCHK(ctx, hipMemcpyAsync(out->state + lo, base + r_st + lo, cnt, hipMemcpyDeviceToHost, s));
        SKCHK(ctx, hipMemcpyAsync(out->flags + lo, base + r_fl + lo, cnt, hipMemcpyDeviceToHost, s));
        SKCHK(ctx, hipMemcpyAsync(out->gate + lo, base + r_gate + lo, cnt, hipMemcpyDeviceToHost, s));
        SKCHK(ctx, hipMemcpyAsync(out->new_pos_xz + 2 * lo, base + r_np + 8 * lo, cnt * 8, hipMemcpyDeviceToHost, s));
        if(out->vel_xz) SKCHK(ctx, hipMemcpyAsync(out->vel_xz + 2 * lo, base + r_vel + 8 * lo, cnt * 8, hipMemcpyDeviceToHost, s));
        if(aux) SKCHK(ctx, hipMemcpyAsync(out->wait_ticks_left + lo, base + r_tk + 4 * lo, cnt * 4, hipMemcpyDeviceToHost, s));
        if(su) SKCHK(ctx, hipMemcpyAsync(X.out_surround_dest_xz + 2 * lo, base + r_sd + 8 * lo, cnt * 8, hipMemcpyDeviceToHost, s));
    }
    SKCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

#define SK_IN_SLOT  42          /* device slab of the resident pass's inputs  */
#define SK_OUT_SLOT 31          /* ... and of its results                      */

int navhip_state_pass_resident(navhip_ctx *ctx, const navhip_state_pass_in *in, const navhip_state_pass_out *out)
{
    if(!ctx || !in || !out) return NAVHIP_ERR_INVALID;
    navhip_world d; navhip_step_out so;
    if(!nh_async_resident(ctx, &d, &so)) {
        ctx->last_error = "navhip_state_pass_resident: no completed host-buffer step is resident on the device";
        return NAVHIP_ERR_INVALID;
    }
    const navhip_gate_in &G = in->gate;
    const navhip_state_in &T = in->state;
    const navhip_state_aux_in &X = in->aux;
    const bool aux = X.fstate != nullptr, turn = aux && X.ent_rot != nullptr, rg = aux && X.range_target != nullptr;
    const bool su = aux && X.surround_target != nullptr, ip = G.interp_from_xz != nullptr;
    const size_t n = (size_t)d.n_ents, F = (size_t)(d.n_flocks > 0 ? d.n_flocks : 0);
    if(n == 0) return NAVHIP_OK;
    if(!d.pos_xz || !d.vel_xz || !d.radius || !d.flags || !d.state || !d.flock || !so.vel_xz || (!so.vdes_xz && !d.vdes_xz) || !G.next_rot
    || !out->state || !out->flags || !out->gate || !out->new_pos_xz
    || (F > 0 && (!d.flock_target_xz || !d.flock_offsets || !d.flock_members || !T.flock_layer || !T.flock_nearest_xz
                  || !T.flock_tiles_off || !T.flock_tiles))
    || (aux && (!X.wait_ticks_left || !X.wait_prev || !out->wait_ticks_left || (X.ent_rot != nullptr) != (X.target_dir != nullptr)))
    || (ip && !G.interp_step) || (!ip && G.interp_step)
    || (su && (!X.surround_query || !X.surround_target_prev_xz || !X.surround_nearest_prev_xz || !X.surround_dest_xz || !X.out_surround_dest_xz)))
        return NAVHIP_ERR_INVALID;
    if(rg) {
        const int given = (X.target_range != nullptr) + (X.target_prev_xz != nullptr) + (X.range_tiles_row != nullptr)
                        + (X.range_tiles_off != nullptr) + (X.range_tiles != nullptr);
        if(given != 5 || X.n_range_rows < 0) return NAVHIP_ERR_INVALID;
    }
    int b, e;
    if(!sk_work_range(&d, &b, &e)) return NAVHIP_ERR_INVALID;
    // sparse rows: the arrays of the three arms hold a row per LISTED unit (ns of them), the device lays them out
    const bool sp = aux && X.sparse_units != nullptr;
    if((X.sparse_units == nullptr && X.n_sparse != 0) || X.n_sparse < 0 || (X.sparse_units && !aux)) return NAVHIP_ERR_INVALID;
    const size_t ns = sp ? (size_t)X.n_sparse : 0;
    const size_t nr = sp ? ns : n;                              // rows of an arm array
    const size_t v0 = sp ? 0 : (size_t)b, v1 = sp ? ns : (size_t)e;   // ... and those the checks below walk
    for(size_t k = 0; k < ns; k++)
        if(X.sparse_units[k] < b || X.sparse_units[k] >= e) return NAVHIP_ERR_INVALID;
    for(size_t f = 0; f < F; f++)
        if(T.flock_tiles_off[f] < 0 || T.flock_tiles_off[f + 1] < T.flock_tiles_off[f]) return NAVHIP_ERR_INVALID;
    const size_t ntiles = F ? (size_t)T.flock_tiles_off[F] : 0;
    size_t rows = 0, n_rt = 0;
    if(rg) {
        rows = (size_t)X.n_range_rows;
        for(size_t r = 0; r < rows; r++)
            if(X.range_tiles_off[r] < 0 || X.range_tiles_off[r + 1] < X.range_tiles_off[r]) return NAVHIP_ERR_INVALID;
        n_rt = rows ? (size_t)X.range_tiles_off[rows] : 0;
        for(size_t i = v0; i < v1; i++)
            if(X.range_target[i] >= d.n_ents
            || (X.range_target[i] >= 0 && (X.range_tiles_row[i] < 0 || (size_t)X.range_tiles_row[i] >= (rows ? rows : 1)))) return NAVHIP_ERR_INVALID;
    }
    if(su)
        for(size_t i = v0; i < v1; i++)
            if(X.surround_target[i] >= d.n_ents) return NAVHIP_ERR_INVALID;
    SKCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // ---- inputs: one device slab; pageable arrays are packed into the pinned slab and cross the bus as ONE transfer,
    // pinned ones (navhip_host_alloc) are transferred in place
    struct item { const void *src; size_t bytes; size_t off; bool pinned; };
    std::vector<item> items;
    sk_arena A;
    auto stage = [&](const void *src, size_t bytes) { const size_t o = A.take(bytes + 8); if(src && bytes) items.push_back({src, bytes, o, nh_is_pinned(src)}); return o; };
    const size_t o_rot = stage(G.next_rot, n * 16), o_if = stage(G.interp_from_xz, ip ? n * 8 : 0), o_is = stage(G.interp_step, ip ? n * 4 : 0);
    const size_t o_skip = stage(T.skip, T.skip ? n : 0), o_flay = stage(T.flock_layer, F), o_fnear = stage(T.flock_nearest_xz, F * 8),
                 o_toff = stage(T.flock_tiles_off, (F + 1) * 4), o_tiles = stage(T.flock_tiles, ntiles * 4);
    const size_t o_fs = stage(X.fstate, aux ? n : 0), o_wt = stage(X.wait_ticks_left, aux ? n * 4 : 0), o_wp = stage(X.wait_prev, aux ? n : 0),
                 o_er = stage(X.ent_rot, turn ? nr * 16 : 0), o_td = stage(X.target_dir, turn ? nr * 16 : 0),
                 o_rt = stage(X.range_target, rg ? nr * 4 : 0), o_rr = stage(X.target_range, rg ? nr * 4 : 0),
                 o_rp = stage(X.target_prev_xz, rg ? nr * 8 : 0), o_row = stage(X.range_tiles_row, rg ? nr * 4 : 0),
                 o_roff = stage(X.range_tiles_off, rg ? (rows + 1) * 4 : 0), o_rtil = stage(X.range_tiles, rg ? n_rt * 4 : 0),
                 o_stgt = stage(X.surround_target, su ? nr * 4 : 0), o_sq = stage(X.surround_query, su ? nr : 0),
                 o_stp = stage(X.surround_target_prev_xz, su ? nr * 8 : 0), o_snp = stage(X.surround_nearest_prev_xz, su ? nr * 8 : 0),
                 o_sd = stage(X.surround_dest_xz, su ? nr * 16 : 0), o_units = stage(X.sparse_units, ns * 4);
    // (sparse: the dense twins the arm kernels read, filled on the device)
    const size_t e_er = A.take(sp && turn ? n * 16 : 0), e_td = A.take(sp && turn ? n * 16 : 0),
                 e_rt = A.take(sp && rg ? n * 4 : 0), e_rr = A.take(sp && rg ? n * 4 : 0), e_rp = A.take(sp && rg ? n * 8 : 0),
                 e_row = A.take(sp && rg ? n * 4 : 0), e_stgt = A.take(sp && su ? n * 4 : 0), e_sq = A.take(sp && su ? n : 0),
                 e_stp = A.take(sp && su ? n * 8 : 0), e_snp = A.take(sp && su ? n * 8 : 0), e_sd = A.take(sp && su ? n * 16 : 0);
    const size_t in_total = A.total;
    // ---- results: rows [b, e) of each, one device slab, one transfer for the pageable destinations
    const size_t cnt = (size_t)(e - b), lo = (size_t)b;
    sk_arena R;
    const size_t r_vel = R.take(n * 8), r_np = R.take(n * 8), r_gate = R.take(n), r_st = R.take(n), r_fl = R.take(n),
                 r_tk = R.take(aux ? n * 4 : 0), r_sd = R.take(su ? n * 8 : 0), r_vd = R.take(n * 8),
                 r_sdk = R.take(sp && su ? ns * 8 : 0);           // (sparse: the surround positions of the listed units)
    size_t pack_in = 0;
    for(auto &it : items) if(!it.pinned) pack_in += (it.bytes + 255) & ~(size_t)255;
    struct oitem { void *dst; size_t dev_off, row; size_t h_off; bool pinned; size_t lo, cnt; };     // rows [lo, lo + cnt) travel
    std::vector<oitem> outs = {{out->state, r_st, 1, 0, false, lo, cnt}, {out->flags, r_fl, 1, 0, false, lo, cnt},
                               {out->gate, r_gate, 1, 0, false, lo, cnt}, {out->new_pos_xz, r_np, 8, 0, false, lo, cnt}};
    if(out->vel_xz) outs.push_back({out->vel_xz, r_vel, 8, 0, false, lo, cnt});
    if(aux) outs.push_back({out->wait_ticks_left, r_tk, 4, 0, false, lo, cnt});
    if(su && !sp) outs.push_back({X.out_surround_dest_xz, r_sd, 8, 0, false, lo, cnt});
    if(su && sp && ns) outs.push_back({X.out_surround_dest_xz, r_sdk, 8, 0, false, 0, ns});
    size_t pack_out = 0;
    for(auto &o : outs) { o.pinned = nh_is_pinned(o.dst); if(!o.pinned) { o.h_off = pack_out; pack_out += (o.cnt * o.row + 255) & ~(size_t)255; } }
    char *h_in = nullptr, *h_out = nullptr, *base = nullptr, *res = nullptr;
    int rc = nh_async_slabs(ctx, pack_in, pack_out, &h_in, &h_out);
    if(!rc) rc = navhip_stage_reserve(ctx, SK_IN_SLOT, in_total + pack_in + 256, (void**)&base);
    if(!rc) rc = navhip_stage_reserve(ctx, SK_OUT_SLOT, R.total + pack_out + 256, (void**)&res);
    if(rc) return rc;
    // (the pageable inputs: packed, sent as one block behind the arrays' own region, then laid out by device-to-device
    // copies?  No: the layout IS the packing order -- every pageable item gets its device address inside the block)
    char *d_block = base + in_total;
    size_t poff = 0;
    std::vector<const char*> dev_of(items.size());
    for(size_t k = 0; k < items.size(); k++) {
        item &it = items[k];
        if(it.pinned) {
            SKCHK(ctx, hipMemcpyAsync(base + it.off, it.src, it.bytes, hipMemcpyHostToDevice, s));
            dev_of[k] = base + it.off;
        }else{
            memcpy(h_in + poff, it.src, it.bytes);
            dev_of[k] = d_block + poff;
            poff += (it.bytes + 255) & ~(size_t)255;
        }
    }
    if(poff) SKCHK(ctx, hipMemcpyAsync(d_block, h_in, poff, hipMemcpyHostToDevice, s));
    auto dev = [&](size_t off) -> const char* {                // the device address of the item staged at arena offset `off`
        for(size_t k = 0; k < items.size(); k++) if(items[k].off == off) return dev_of[k];
        return base + off;                                     // (nothing staged there: an address nobody reads)
    };
    // ---- sparse rows -> the dense arrays of the arms (targets default to "the host's", rotations and queries to zero)
    const char *a_er = dev(o_er), *a_td = dev(o_td), *a_rt = dev(o_rt), *a_rr = dev(o_rr), *a_rp = dev(o_rp), *a_row = dev(o_row),
               *a_stgt = dev(o_stgt), *a_sq = dev(o_sq), *a_stp = dev(o_stp), *a_snp = dev(o_snp), *a_sd = dev(o_sd);
    if(sp) {
        sk_sparse_rows SR;
        memset(&SR, 0, sizeof(SR));
        SR.n = (int)ns; SR.units = (const int32_t*)dev(o_units);
        if(turn) {
            SKCHK(ctx, hipMemsetAsync(base + e_er, 0, n * 16, s)); SKCHK(ctx, hipMemsetAsync(base + e_td, 0, n * 16, s));
            SR.er = (const float*)a_er; SR.td = (const float*)a_td; SR.d_er = (float*)(base + e_er); SR.d_td = (float*)(base + e_td);
            a_er = base + e_er; a_td = base + e_td;
        }
        if(rg) {
            SKCHK(ctx, hipMemsetAsync(base + e_rt, 0xfe, n * 4, s)); SKCHK(ctx, hipMemsetAsync(base + e_row, 0, n * 4, s));
            SR.rt = (const int32_t*)a_rt; SR.rr = (const float*)a_rr; SR.rp = (const float*)a_rp; SR.row = (const int32_t*)a_row;
            SR.d_rt = (int32_t*)(base + e_rt); SR.d_rr = (float*)(base + e_rr); SR.d_rp = (float*)(base + e_rp); SR.d_row = (int32_t*)(base + e_row);
            a_rt = base + e_rt; a_rr = base + e_rr; a_rp = base + e_rp; a_row = base + e_row;
        }
        if(su) {
            SKCHK(ctx, hipMemsetAsync(base + e_stgt, 0xfe, n * 4, s)); SKCHK(ctx, hipMemsetAsync(base + e_sq, 0, n, s));
            SKCHK(ctx, hipMemsetAsync(res + r_sd, 0, n * 8, s));
            SR.stgt = (const int32_t*)a_stgt; SR.sq = (const uint8_t*)a_sq; SR.stp = (const float*)a_stp; SR.snp = (const float*)a_snp;
            SR.sd = (const float*)a_sd;
            SR.d_stgt = (int32_t*)(base + e_stgt); SR.d_sq = (uint8_t*)(base + e_sq); SR.d_stp = (float*)(base + e_stp);
            SR.d_snp = (float*)(base + e_snp); SR.d_sd = (float*)(base + e_sd);
            a_stgt = base + e_stgt; a_sq = base + e_sq; a_stp = base + e_stp; a_snp = base + e_snp; a_sd = base + e_sd;
        }
        if(ns) {
            hipLaunchKernelGGL(k_sparse_rows, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, SR);
            SKCHK(ctx, hipGetLastError());
        }
    }
    // ---- the three passes on the resident snapshot
    const float *d_vdes = (const float*)(res + r_vd);
    if(e > b) hipLaunchKernelGGL(k_pick_vdes, dim3((e - b + 255) / 256), dim3(256), 0, s, b, e, d.vdes_xz, (const float*)so.vdes_xz, (float*)(res + r_vd));
    navhip_gate_in dg = {(const float*)dev(o_rot), so.vel_xz, d_vdes, ip ? (const float*)dev(o_if) : nullptr, ip ? (const float*)dev(o_is) : nullptr};
    rc = navhip_heading_gate_dev(ctx, &d, &dg, (float*)(res + r_vel), (float*)(res + r_np), (uint8_t*)(res + r_gate), s);
    if(rc) return rc;
    navhip_state_in ds = {(const float*)(res + r_np), d_vdes, T.skip ? (const uint8_t*)dev(o_skip) : nullptr, (const uint8_t*)dev(o_flay),
                          (const float*)dev(o_fnear), (const int32_t*)dev(o_toff), (const int16_t*)dev(o_tiles)};
    rc = navhip_state_update_dev(ctx, &d, &ds, (uint8_t*)(res + r_st), (uint8_t*)(res + r_fl), s);
    if(rc) return rc;
    if(aux) {
        navhip_state_aux_in da;
        memset(&da, 0, sizeof(da));
        da.fstate = (const uint8_t*)dev(o_fs); da.wait_ticks_left = (const int32_t*)dev(o_wt);
        da.wait_prev = (const uint8_t*)dev(o_wp); da.new_pos_xz = (const float*)(res + r_np);
        if(turn) { da.ent_rot = (const float*)a_er; da.target_dir = (const float*)a_td; }
        if(rg) {
            da.range_target = (const int32_t*)a_rt; da.target_range = (const float*)a_rr;
            da.target_prev_xz = (const float*)a_rp; da.range_tiles_row = (const int32_t*)a_row;
            da.range_tiles_off = (const int32_t*)dev(o_roff); da.range_tiles = (const int16_t*)dev(o_rtil);
            da.n_range_rows = X.n_range_rows;
        }
        if(su) {
            da.surround_target = (const int32_t*)a_stgt; da.surround_query = (const uint8_t*)a_sq;
            da.surround_target_prev_xz = (const float*)a_stp; da.surround_nearest_prev_xz = (const float*)a_snp;
            da.surround_dest_xz = (const float*)a_sd; da.vdes_xz = d_vdes; da.out_surround_dest_xz = (float*)(res + r_sd);
        }
        rc = navhip_state_update_aux_dev(ctx, &d, &da, (uint8_t*)(res + r_st), (uint8_t*)(res + r_fl), (int32_t*)(res + r_tk), s);
        if(rc) return rc;
    }
    if(e > b) {
        hipLaunchKernelGGL(k_gate_host_rows, dim3((e - b + 255) / 256), dim3(256), 0, s, b, e, (const uint8_t*)(res + r_gate),
                           d.state, aux ? (const int32_t*)dev(o_wt) : (const int32_t*)nullptr,
                           (uint8_t*)(res + r_st), (uint8_t*)(res + r_fl), aux ? (int32_t*)(res + r_tk) : (int32_t*)nullptr);
        SKCHK(ctx, hipGetLastError());
        if(su && sp && ns) {
            hipLaunchKernelGGL(k_sparse_gather2, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, (int)ns, (const int32_t*)dev(o_units),
                               (const float*)(res + r_sd), (float*)(res + r_sdk));
            SKCHK(ctx, hipGetLastError());
        }
        // results: the pageable destinations' rows are gathered into one block on the device and cross the bus once
        char *d_oblock = res + R.total;
        for(auto &o : outs) {
            if(o.pinned) SKCHK(ctx, hipMemcpyAsync((char*)o.dst + o.lo * o.row, res + o.dev_off + o.lo * o.row, o.cnt * o.row, hipMemcpyDeviceToHost, s));
            else         SKCHK(ctx, hipMemcpyAsync(d_oblock + o.h_off, res + o.dev_off + o.lo * o.row, o.cnt * o.row, hipMemcpyDeviceToDevice, s));
        }
        if(pack_out) SKCHK(ctx, hipMemcpyAsync(h_out, d_oblock, pack_out, hipMemcpyDeviceToHost, s));
    }
    SKCHK(ctx, hipStreamSynchronize(s));
    if(e > b)
        for(auto &o : outs) if(!o.pinned) memcpy((char*)o.dst + o.lo * o.row, h_out + o.h_off, o.cnt * o.row);
    return NAVHIP_OK;
}

static int sk_settled_count(navhip_ctx *ctx, const navhip_world *w, int nq, const int32_t *uids, int32_t *out_counts, bool use_resident);

int navhip_settled_count(navhip_ctx *ctx, const navhip_world *w, int nq, const int32_t *uids, int32_t *out_counts)
{
    return sk_settled_count(ctx, w, nq, uids, out_counts, false);
}

int navhip_settled_count_resident(navhip_ctx *ctx, const navhip_world *w, int nq, const int32_t *uids, int32_t *out_counts)
{
    return sk_settled_count(ctx, w, nq, uids, out_counts, true);
}

static int sk_settled_count(navhip_ctx *ctx, const navhip_world *w, int nq, const int32_t *uids, int32_t *out_counts, bool use_resident)
{
    if(!ctx || !w || nq < 0 || w->n_ents < 0 || (nq > 0 && (!uids || !out_counts))) return NAVHIP_ERR_INVALID;
    if(nq == 0) return NAVHIP_OK;
    if(!use_resident && (!w->pos_xz || !w->radius || !w->flags || !w->state)) return NAVHIP_ERR_INVALID;
    for(int q = 0; q < nq; q++)
        if(uids[q] < 0 || uids[q] >= w->n_ents) return NAVHIP_ERR_INVALID;
    // the spatial index over the snapshot, the circle queries and the count, all on the device: the ids of the queries
    // (the reference's visiting order, capped) never leave HBM -- only the uids go up and the counts come back
    const size_t n = (size_t)w->n_ents;
    SKCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    navhip_world d; navhip_step_out so;
    const bool resident = use_resident && nh_async_resident(ctx, &d, &so) && d.n_ents == w->n_ents && d.pos_xz && d.radius && d.flags && d.state;
    if(use_resident && !resident) {
        ctx->last_error = "navhip_settled_count_resident: no completed host-buffer step of this size is resident on the device";
        return NAVHIP_ERR_INVALID;
    }
    sk_arena A;
    const size_t o_pos = A.take(resident ? 0 : n * 8), o_rad = A.take(resident ? 0 : n * 4), o_fl = A.take(resident ? 0 : n * 4),
                 o_st = A.take(resident ? 0 : n), o_uid = A.take((size_t)nq * 4), o_q = A.take((size_t)nq * 8),
                 o_cnt = A.take((size_t)nq * 4), o_ids = A.take((size_t)nq * SK_QUERY_MAX * 4), o_out = A.take((size_t)nq * 4);
    char *base;
    int rc = navhip_stage_reserve(ctx, SK_SLOT, A.total, (void**)&base);
    if(rc) return rc;
    if(!resident) {
        // (a snapshot the velocity pass left on the device is read in place: the tick's tables are the same)
        SKCHK(ctx, hipMemcpyAsync(base + o_pos, w->pos_xz, n * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_rad, w->radius, n * 4, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_fl, w->flags, n * 4, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_st, w->state, n, hipMemcpyHostToDevice, s));
        d = *w;
        d.pos_xz = (const float*)(base + o_pos); d.radius = (const float*)(base + o_rad);
        d.flags = (const uint32_t*)(base + o_fl); d.state = (const uint8_t*)(base + o_st);
    }
    d.grid_xmin = w->grid_xmin; d.grid_xmax = w->grid_xmax; d.grid_zmin = w->grid_zmin; d.grid_zmax = w->grid_zmax;
    SKCHK(ctx, hipMemcpyAsync(base + o_uid, uids, (size_t)nq * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_gather_query, dim3((nq + 255) / 256), dim3(256), 0, s, nq, (const int32_t*)(base + o_uid), d.pos_xz, (float*)(base + o_q));
    rc = nh_spatial_query_dev(ctx, &d, (const float*)(base + o_q), nq, SK_QUERY_R, SK_QUERY_MAX, (int32_t*)(base + o_cnt),
                              (uint32_t*)(base + o_ids), s);
    if(rc) return rc;
    hipLaunchKernelGGL(k_settled_count, dim3((nq + 15) / 16), dim3(256), 0, s, nq, (const int32_t*)(base + o_uid),
                       d.pos_xz, d.radius, d.flags, d.state, (const int32_t*)(base + o_cnt), (const uint32_t*)(base + o_ids),
                       (int32_t*)(base + o_out));
    SKCHK(ctx, hipGetLastError());
    SKCHK(ctx, hipMemcpyAsync(out_counts, base + o_out, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    SKCHK(ctx, hipStreamSynchronize(s));
    return NAVHIP_OK;
}

int navhip_arrival_settle_dev(navhip_ctx *ctx, const navhip_world *w, const navhip_settle_in *in,
                              const navhip_settle_out *out, void *stream)
{
    if(!ctx || !w || !in || !out || in->nq < 0 || in->n_zones < 0 || w->n_ents < 0) return NAVHIP_ERR_INVALID;
    if(in->nq == 0) return NAVHIP_OK;
    if(!w->vel_xz || !w->radius || !in->zones || in->n_zones < 1 || !in->uid || !in->zone || !in->new_pos_xz
    || !in->nsettled || !in->substate || !in->sink_valid || !in->sink_xz || !in->order_pos_xz || !in->progress_anchor_xz
    || !in->progress_anchored || !in->stuck || !out->settle || !out->substate || !out->progress_anchor_xz
    || !out->progress_anchored || !out->stuck)
        return NAVHIP_ERR_INVALID;
    SKCHK(ctx, hipSetDevice(ctx->device));
    nh_step_params P;
    sk_map_view(ctx, w, &P);
    hipLaunchKernelGGL(k_arrival_settle, dim3((in->nq + 15) / 16), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream,
                       P, w->vel_xz, w->radius, *in, *out);
    SKCHK(ctx, hipGetLastError());
    return NAVHIP_OK;
}

static int sk_arrival_settle(navhip_ctx *ctx, const navhip_world *w, const navhip_settle_in *in, const navhip_settle_out *out, bool use_resident);

int navhip_arrival_settle(navhip_ctx *ctx, const navhip_world *w, const navhip_settle_in *in, const navhip_settle_out *out)
{
    return sk_arrival_settle(ctx, w, in, out, false);
}

int navhip_arrival_settle_resident(navhip_ctx *ctx, const navhip_world *w, const navhip_settle_in *in, const navhip_settle_out *out)
{
    return sk_arrival_settle(ctx, w, in, out, true);
}

static int sk_arrival_settle(navhip_ctx *ctx, const navhip_world *w, const navhip_settle_in *in, const navhip_settle_out *out, bool use_resident)
{
    if(!ctx || !w || !in || !out || in->nq < 0 || in->n_zones < 0 || w->n_ents < 0) return NAVHIP_ERR_INVALID;
    if(in->nq == 0) return NAVHIP_OK;
    const bool count_here = in->nsettled == nullptr;          // (adjacent_settled_count on the resident snapshot too)
    if(count_here && !use_resident) return NAVHIP_ERR_INVALID;
    navhip_world rw; navhip_step_out rso;
    const bool resident = use_resident && nh_async_resident(ctx, &rw, &rso) && rw.n_ents == w->n_ents && rw.vel_xz && rw.radius
                       && (!count_here || (rw.pos_xz && rw.flags && rw.state));
    if(use_resident && !resident) {
        ctx->last_error = "navhip_arrival_settle_resident: no completed host-buffer step of this size is resident on the device";
        return NAVHIP_ERR_INVALID;
    }
    if((!resident && (!w->vel_xz || !w->radius)) || !in->zones || in->n_zones < 1 || !in->uid || !in->zone || !in->new_pos_xz
    || !in->substate || !in->sink_valid || !in->sink_xz || !in->order_pos_xz || !in->progress_anchor_xz
    || !in->progress_anchored || !in->stuck || !out->settle || !out->substate || !out->progress_anchor_xz
    || !out->progress_anchored || !out->stuck)
        return NAVHIP_ERR_INVALID;
    const size_t n = (size_t)w->n_ents, nq = (size_t)in->nq, nz = (size_t)in->n_zones;
    size_t n_slots = 0, n_keys = 0;
    for(size_t z = 0; z < nz; z++) {
        const navhip_arrival_zone &Z = in->zones[z];
        if(Z.slot_begin < 0 || Z.slot_end < Z.slot_begin || Z.key_begin < 0 || Z.key_end < Z.key_begin
        || Z.layer < 0 || Z.layer >= NAVHIP_NAV_LAYER_MAX) return NAVHIP_ERR_INVALID;
        if((size_t)Z.slot_end > n_slots) n_slots = (size_t)Z.slot_end;
        if((size_t)Z.key_end > n_keys) n_keys = (size_t)Z.key_end;
    }
    if((n_slots > 0 && (!in->slots_xz || !in->slot_ring)) || (n_keys > 0 && !in->region_keys)) return NAVHIP_ERR_INVALID;
    for(size_t q = 0; q < nq; q++)
        if(in->uid[q] < 0 || in->uid[q] >= w->n_ents || in->zone[q] < 0 || in->zone[q] >= in->n_zones) return NAVHIP_ERR_INVALID;
    SKCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    sk_arena A;
    const size_t o_vel = A.take(resident ? 0 : n * 8), o_rad = A.take(resident ? 0 : n * 4);
    // (the inputs of the arm from here to in_end, the results from r_set to the end: each region crosses the bus once)
    const size_t o_z = A.take(nz * sizeof(navhip_arrival_zone)),
                 o_sl = A.take(n_slots * 8 + 8), o_ring = A.take(n_slots * 4 + 4), o_keys = A.take(n_keys * 8 + 8),
                 o_uid = A.take(nq * 4), o_zone = A.take(nq * 4), o_np = A.take(nq * 8), o_ns = A.take(count_here ? 0 : nq * 4),
                 o_sub = A.take(nq), o_sv = A.take(nq), o_sink = A.take(nq * 8), o_ord = A.take(nq * 8),
                 o_anc = A.take(nq * 8), o_and = A.take(nq), o_stk = A.take(nq * 4), in_end = A.total,
                 o_q = A.take(count_here ? nq * 8 : 0), o_cnt = A.take(count_here ? nq * 4 : 0),
                 o_ids = A.take(count_here ? nq * SK_QUERY_MAX * 4 : 0),
                 r_set = A.take(nq), r_sub = A.take(nq), r_anc = A.take(nq * 8), r_and = A.take(nq), r_stk = A.take(nq * 4),
                 r_ns = A.take(count_here ? nq * 4 : 0);
    char *base;
    int rc = navhip_stage_reserve(ctx, SK_SLOT, A.total, (void**)&base);
    if(rc) return rc;
    struct up { size_t off; const void *src; size_t bytes; };
    const up ups[] = {{o_z, in->zones, nz * sizeof(navhip_arrival_zone)}, {o_sl, in->slots_xz, n_slots * 8}, {o_ring, in->slot_ring, n_slots * 4},
                      {o_keys, in->region_keys, n_keys * 8}, {o_uid, in->uid, nq * 4}, {o_zone, in->zone, nq * 4}, {o_np, in->new_pos_xz, nq * 8},
                      {o_ns, in->nsettled, count_here ? 0 : nq * 4}, {o_sub, in->substate, nq}, {o_sv, in->sink_valid, nq},
                      {o_sink, in->sink_xz, nq * 8}, {o_ord, in->order_pos_xz, nq * 8}, {o_anc, in->progress_anchor_xz, nq * 8},
                      {o_and, in->progress_anchored, nq}, {o_stk, in->stuck, nq * 4}};
    struct down { size_t off; void *dst; size_t bytes; };
    const down downs[] = {{r_set, out->settle, nq}, {r_sub, out->substate, nq}, {r_anc, out->progress_anchor_xz, nq * 8},
                          {r_and, out->progress_anchored, nq}, {r_stk, out->stuck, nq * 4},
                          {r_ns, out->nsettled, count_here && out->nsettled ? nq * 4 : 0}};
    // a context with an asynchronous step path owns page-locked slabs: the small arrays are packed into them
    char *h_in = nullptr, *h_out = nullptr;
    const bool packed = nh_async_slabs(ctx, in_end - o_z, A.total - r_set, &h_in, &h_out) == NAVHIP_OK;
    if(!resident) {
        SKCHK(ctx, hipMemcpyAsync(base + o_vel, w->vel_xz, n * 8, hipMemcpyHostToDevice, s));
        SKCHK(ctx, hipMemcpyAsync(base + o_rad, w->radius, n * 4, hipMemcpyHostToDevice, s));
    }
    if(packed) {
        for(const up &u : ups) if(u.bytes) memcpy(h_in + (u.off - o_z), u.src, u.bytes);
        SKCHK(ctx, hipMemcpyAsync(base + o_z, h_in, in_end - o_z, hipMemcpyHostToDevice, s));
    }else
        for(const up &u : ups) if(u.bytes) SKCHK(ctx, hipMemcpyAsync(base + u.off, u.src, u.bytes, hipMemcpyHostToDevice, s));
    navhip_world d = *w;
    // (the snapshot the velocity half left on the device is read in place: movestate.velocity and the radii are the tick's)
    d.vel_xz = resident ? rw.vel_xz : (const float*)(base + o_vel); d.radius = resident ? rw.radius : (const float*)(base + o_rad);
    if(count_here) {
        navhip_world dq = rw;
        dq.grid_xmin = w->grid_xmin; dq.grid_xmax = w->grid_xmax; dq.grid_zmin = w->grid_zmin; dq.grid_zmax = w->grid_zmax;
        hipLaunchKernelGGL(k_gather_query, dim3((in->nq + 255) / 256), dim3(256), 0, s, in->nq, (const int32_t*)(base + o_uid), rw.pos_xz, (float*)(base + o_q));
        rc = nh_spatial_query_dev(ctx, &dq, (const float*)(base + o_q), in->nq, SK_QUERY_R, SK_QUERY_MAX, (int32_t*)(base + o_cnt),
                                  (uint32_t*)(base + o_ids), s);
        if(rc) return rc;
        hipLaunchKernelGGL(k_settled_count, dim3((in->nq + 15) / 16), dim3(256), 0, s, in->nq, (const int32_t*)(base + o_uid),
                           rw.pos_xz, rw.radius, rw.flags, rw.state, (const int32_t*)(base + o_cnt), (const uint32_t*)(base + o_ids),
                           (int32_t*)(base + r_ns));
        SKCHK(ctx, hipGetLastError());
    }
    navhip_settle_in di = *in;
    di.zones = (const navhip_arrival_zone*)(base + o_z); di.slots_xz = (const float*)(base + o_sl);
    di.slot_ring = (const int32_t*)(base + o_ring); di.region_keys = (const uint64_t*)(base + o_keys);
    di.uid = (const int32_t*)(base + o_uid); di.zone = (const int32_t*)(base + o_zone);
    di.new_pos_xz = (const float*)(base + o_np); di.nsettled = (const int32_t*)(base + (count_here ? r_ns : o_ns));
    di.substate = (const uint8_t*)(base + o_sub); di.sink_valid = (const uint8_t*)(base + o_sv);
    di.sink_xz = (const float*)(base + o_sink); di.order_pos_xz = (const float*)(base + o_ord);
    di.progress_anchor_xz = (const float*)(base + o_anc); di.progress_anchored = (const uint8_t*)(base + o_and);
    di.stuck = (const int32_t*)(base + o_stk);
    navhip_settle_out dout = {(uint8_t*)(base + r_set), (uint8_t*)(base + r_sub), (float*)(base + r_anc),
                              (uint8_t*)(base + r_and), (int32_t*)(base + r_stk), nullptr};
    rc = navhip_arrival_settle_dev(ctx, &d, &di, &dout, s);
    if(rc) return rc;
    if(packed) SKCHK(ctx, hipMemcpyAsync(h_out, base + r_set, A.total - r_set, hipMemcpyDeviceToHost, s));
    else for(const down &o : downs) if(o.bytes) SKCHK(ctx, hipMemcpyAsync(o.dst, base + o.off, o.bytes, hipMemcpyDeviceToHost, s));
    SKCHK(ctx, hipStreamSynchronize(s));
    if(packed) for(const down &o : downs) if(o.bytes) memcpy(o.dst, h_out + (o.off - r_set), o.bytes);
    return NAVHIP_OK;
}

}   // extern "C"
