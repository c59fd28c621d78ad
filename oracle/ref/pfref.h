/* oracle/ref/pfref.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Flat C API of oracle/_ref/libpfref.so: the reference engine's own
 * navigation / ClearPath / movement translation units, compiled unmodified
 * from where they lie under /root/reference/src and driven through a thin
 * harness.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call this.
 *
 * Every entry point names the reference function it drives.
 */
#ifndef PFREF_H
#define PFREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfref_nav pfref_nav;

/* Flattened `struct field_target` + call arguments of N_FlowFieldUpdate
 * (field.h:85-109,146-152). */
typedef struct pfref_field_req {
    int32_t  layer;
    int32_t  type;          /* TARGET_PORTAL = 0, TARGET_TILE = 1 (field.h:86-87) */
    int32_t  faction_id;
    int32_t  inout;         /* 1: N_FlowFieldUpdate ran on an existing field (nav.c:1998-2001) */
    int32_t  chunk_r, chunk_c;
    int32_t  tile_r, tile_c;
    /* struct portal_desc (field.h:67-72) */
    int32_t  port_r0, port_c0, port_r1, port_c1;
    int32_t  next_chunk_r, next_chunk_c;
    int32_t  next_r0, next_c0, next_r1, next_c1;
    int32_t  port_iid, next_iid;
} pfref_field_req;

typedef struct pfref_portal {
    int32_t chunk_r, chunk_c;
    int32_t r0, c0, r1, c1;
    int32_t conn_chunk_r, conn_chunk_c;
    int32_t conn_r0, conn_c0, conn_r1, conn_c1;
    int32_t component_id;
} pfref_portal;

/* --- context ------------------------------------------------------------ */

/* Build a `struct nav_private` (nav_private.h:52) for the layers in
 * `layer_mask` from explicit cost_base planes ([h][w][64][64] u8, the
 * N_CopyCostBasePacked layout nav.c:2432), then run the reference's own
 * n_update_portals / n_update_island_field / n_update_local_island_field
 * (nav.c:1710,1731,986) exactly as N_NewCtxForMapData does (nav.c:2340-2343). */
pfref_nav *pfref_nav_create(int w, int h, const uint8_t *cost_base, unsigned layer_mask);
void       pfref_nav_destroy(pfref_nav *nav);
void      *pfref_nav_private(pfref_nav *nav);   /* the struct nav_private* */
int        pfref_nav_width(const pfref_nav *nav);
int        pfref_nav_height(const pfref_nav *nav);

/* Replace the blockers plane of one layer ([h][w][64][64] u16) and relabel
 * local islands with the reference's n_update_local_island_field. */
void pfref_nav_set_blockers(pfref_nav *nav, int layer, const uint16_t *blockers);
/* N_BlockersIncref / N_BlockersDecref (nav.c:4663,4685) + N_Update-equivalent
 * relabel of dirty local islands (nav.c:996). */
void pfref_nav_blockers_circle(pfref_nav *nav, float x, float z, float range, int faction_id,
                               uint32_t flags, int incref);
void pfref_nav_flush_dirty(pfref_nav *nav);

/* plane: 0 cost_base(u8) 1 blockers(u16) 2 islands(u16) 3 local_islands(u16)
 *        4 factions (u8, [chunk][15][64][64]).  Layout [h][w][64][64]. */
size_t pfref_nav_copy_plane(const pfref_nav *nav, int layer, int plane, void *out);

int  pfref_nav_num_portals(const pfref_nav *nav, int layer, int chunk_r, int chunk_c);
void pfref_nav_get_portal(const pfref_nav *nav, int layer, int chunk_r, int chunk_c, int idx,
                          pfref_portal *out);

/* G_GetEnemyFactions(faction_id) as the harness answers it (game.c:2744): bit f = at war with f */
void pfref_set_enemy_factions(int faction_id, unsigned mask);

/* --- flow fields -------------------------------------------------------- */

/* N_FlowFieldInit (unless req->inout) + N_FlowFieldUpdate (field.c:2020,2030).
 * inout_dirs: 4096 bytes, one dir_idx (0..8) per cell, row-major [64][64].
 * out_integ (optional): the float integration field the reference computed
 * for this call, captured by replaying N_FlowFieldUpdate's own sequence of
 * static calls (field.c:2055-2077); INFINITY for unreached cells. */
int pfref_field_update(pfref_nav *nav, const pfref_field_req *req,
                       uint8_t *inout_dirs, float *out_integ);

/* N_FlowFieldID (field.c:1952): TILE / PORTAL requests, and the region targets
 * (kind = field_target.type: 2 ENEMIES a = faction_id; 4 ENTITY a = target uid;
 *  5 ZONE a, b = centre in absolute nav tiles (row, column), c = radius) */
uint64_t pfref_flow_field_id(pfref_nav *nav, const pfref_field_req *req);
uint64_t pfref_region_field_id(int kind, int layer, int chunk_r, int chunk_c, uint32_t a, int b, int c);

/* the repair builds the sampler runs on an existing field (nav.c:3527-3547) */
int pfref_field_nearest_pathable(pfref_nav *nav, int layer, int chunk_r, int chunk_c, int start_r,
                                 int start_c, int faction_id, uint8_t *inout_dirs);
int pfref_field_island_to_nearest(pfref_nav *nav, const pfref_field_req *req, int local_iid,
                                  uint8_t *inout_dirs);

/* region fields: N_CellArrivalFieldCreate :2445, N_GroupArrivalFieldCreate :2525, TARGET_ZONE */
int pfref_cell_arrival_field(pfref_nav *nav, int dim, int layer, int enemies,
                             int tgt_abs_r, int tgt_abs_c, int cen_abs_r, int cen_abs_c,
                             const int16_t *blocked, int n_blocked, uint8_t *out);
int pfref_group_arrival_field(pfref_nav *nav, int dim, int layer, int enemies, const float *targets_xz,
                              int ntargets, float center_x, float center_z,
                              const int16_t *blocked, int n_blocked, uint8_t *out);
int pfref_zone_field(pfref_nav *nav, int layer, int chunk_r, int chunk_c, int cen_abs_r, int cen_abs_c,
                     int radius, uint8_t *inout_dirs, int16_t *out_seeds, int max_seeds, int *out_geom);

/* N_LOSFieldCreate (field.c:2085); fields are 4096 bytes, bit 0 visible, bit 1 wavefront_blocked */
int pfref_los_field(pfref_nav *nav, int layer, int faction_id, int chunk_r, int chunk_c,
                    int tgt_chunk_r, int tgt_chunk_c, int tgt_tile_r, int tgt_tile_c,
                    int prev_dr, int prev_dc, const uint8_t *prev, uint8_t *out);

/* Time `reps` passes of N_FlowFieldInit+N_FlowFieldUpdate over `n` requests on
 * `nthreads` pthreads (requests are independent; nav_private is read-only).
 * Returns seconds of wall time (CLOCK_MONOTONIC). */
double pfref_field_bench(pfref_nav *nav, const pfref_field_req *reqs, int n, int reps,
                         int nthreads);
/* The same build with the results kept (no in-place requests): out_dirs[n][4096]. */
double pfref_field_update_many(pfref_nav *nav, const pfref_field_req *reqs, int n, int nthreads,
                               uint8_t *out_dirs);

/* --- planner trace ------------------------------------------------------ */

/* n_request_path (nav.c:1774) from xz_src to xz_dst with the field cache
 * cleared first when `clear_cache`; every N_FlowFieldUpdate call the planner
 * makes is recorded.  Returns 1 when a path exists. */
int    pfref_request_path(pfref_nav *nav, int layer, int faction_id,
                          float src_x, float src_z, float dst_x, float dst_z,
                          int clear_cache, uint32_t *out_dest_id);
int    pfref_trace_count(void);
void   pfref_trace_get(int idx, pfref_field_req *out_req, uint8_t *out_before, uint8_t *out_after);
void   pfref_trace_clear(void);
double pfref_move_bench_hip(const float *vdes, int begin, int end, int reps, double out_times[4]);
int    pfref_cache_put_fields(pfref_nav *nav, int n, const pfref_field_req *reqs, const uint32_t *dest_ids,
                              const uint8_t *dirs);
/* the planner's N_LOSFieldCreate calls: {dest id, chunk r, c, has_prev, prev chunk r, c} */
int    pfref_los_trace_count(void);
void   pfref_los_trace_get(int idx, int32_t out[6]);
void   pfref_los_trace_clear(void);

/* N_DesiredPointSeekVelocity (nav.c:3468): may call n_request_path on a miss. */
void   pfref_desired_point_seek_velocity(pfref_nav *nav, uint32_t dest_id, float x, float z,
                                         float dst_x, float dst_z, float out[2]);
/* (dest,chunk) -> cached field (fieldcache.c): returns 1 and copies 4096 dirs */
uint64_t pfref_cached_ffid(pfref_nav *nav, uint32_t dest_id, int chunk_r, int chunk_c);
int    pfref_cached_los(pfref_nav *nav, uint32_t dest_id, int chunk_r, int chunk_c, uint8_t *out);
int    pfref_cached_field(pfref_nav *nav, uint32_t dest_id, int chunk_r, int chunk_c,
                          uint8_t *out_dirs);
/* N_HasDestLOS (nav.c:4026) */
int    pfref_has_dest_los(pfref_nav *nav, uint32_t dest_id, float x, float z, float dst_x, float dst_z);
int    pfref_position_pathable(pfref_nav *nav, int layer, float x, float z);
int    pfref_position_blocked(pfref_nav *nav, int layer, float x, float z);
void   pfref_map_pos(const pfref_nav *nav, float out[3]);

/* --- the reference-side binding of libnavhip.so (bindings/permafrost/nav_hip.c, move_hip.c) ------------- */

/* N_HIP_Init: create the device context for this map and upload its planes.  0 = no GPU. */
int  pfref_hip_init(pfref_nav *nav);
void pfref_hip_shutdown(void);
/* use_binding != 0: every N_FlowFieldUpdate / ...ToNearestPathable / ...IslandToNearest call nav.c makes
 * goes through the binding; backend 0 = the reference's CPU builders behind it, 1 = libnavhip */
void pfref_hip_mode(int use_binding, int backend);
int  pfref_hip_sync_layer(pfref_nav *nav, int layer);        /* after blocker changes */
void pfref_hip_stats(long out[3]);                           /* device builds, batches, requests */
/* N_DesiredPointSeekVelocity for n agents: batched = 0 serial calls in order, 1 = the
 * miss-collecting batched form (N_HIP_DesiredPointSeekVelocities) */
void pfref_desired_velocities(pfref_nav *nav, int n, const uint32_t *dest_ids, const float *pos_xz,
                              const float *dest_xz, int batched, float *out_xz);
uint32_t pfref_dest_id(pfref_nav *nav, int layer, int faction_id, float dst_x, float dst_z);
void pfref_cache_clear(pfref_nav *nav);
/* move_velocity_work through the WORK_TYPE_HIP arm; 0 = the arm declined (no device) */
int  pfref_move_velocity_hip(const float *vdes, int begin, int end, float *out_vel);

/* --- the asynchronous field batch / LOS / blockers seams of the binding --------------------------- */

/* explicit game state for the reference's enemy / entity frontier extraction (field.c:1209-1370):
 * every entity uid == index; faction 0..14; flags ENTITY_FLAG_* */
void pfref_game_load(float xmin, float xmax, float zmin, float zmax, int n, const float *pos_xz,
                     const float *radius, const int32_t *faction, const uint32_t *flags);
void pfref_game_unload(void);

typedef struct pfref_async_req {
    int32_t  kind;          /* 0 enemy seek, 1 surround entity, 2 group arrival (zone) */
    int32_t  layer;
    int32_t  faction_id;
    float    x, z;          /* curr_pos (0, 1) / centre_pos (2) */
    uint32_t ent;           /* 1: the surrounded entity */
    int32_t  radius;        /* 2: zone radius in tiles */
} pfref_async_req;

/* compute_async_fields of one tick (movement.c:4149-4164); returns the number of jobs */
int  pfref_async_batch(pfref_nav *nav, int n, const pfref_async_req *reqs, uint64_t *out_ids, int max_ids);
int  pfref_cached_field_by_id(pfref_nav *nav, uint64_t ffid, uint8_t *out_dirs);
/* N_DesiredEnemySeekVelocity / N_DesiredSurroundVelocity / N_DesiredGroupArrivalVelocity (nav.c:3603,3687,3561)
 * per agent (kind 2: the zone centre in the bits of ent (x) and faction_id (z)); out_flags bit 0: the
 * group-arrival lookup returned true, bit 1: at_slot */
void pfref_desired_region_velocities(pfref_nav *nav, int n, const pfref_async_req *reqs, float *out_xz, uint8_t *out_flags);
void pfref_hip_async_stats(long out[3]);      /* device jobs, device batches, jobs left to the CPU builders */
void pfref_hip_los_stats(long out[2]);        /* device LOS fields, device batches */
void pfref_hip_blockers_stats(long out[2]);   /* circles flushed, device batches */
int  pfref_hip_blockers_flush(void);          /* N_HIP_BlockersFlush */
void *pfref_hip_ctx(void);                    /* the binding's navhip_ctx* (tests read the device planes) */
void pfref_nav_dirty_chunks(pfref_nav *nav, int layer, uint8_t *flags);

/* the field cache's device image: N_HIP_PoolEnable (nav_hip.c) + device sampling in the WORK_TYPE_HIP arm */
int  pfref_hip_pool_enable(int n_slots, int n_rows);
void pfref_hip_pool_disable(void);
void pfref_hip_pool_stats(long out[3]);       /* host puts mirrored, mappings mirrored, fields built resident */
void pfref_move_hip_sampling(int on);
void pfref_move_hip_dry_run(int on);
int  pfref_move_hip_snapshot(int cap, float *pos, float *vel, float *radius, float *max_speed, uint32_t *flags,
                             uint8_t *state, int32_t *flock, int32_t *flock_offsets, int32_t *flock_members, int32_t *n_flocks);
void pfref_move_hip_threads(int nthreads, int min_items);   /* fork-join width of the binding's host loops (1 = serial) */
void pfref_move_hip_stats(long out[3]);       /* agents sampled on the device, host fallbacks, steps */

/* --- ClearPath ---------------------------------------------------------- */

/* G_ClearPath_NewVelocity (clearpath.c:694).  ent/neighbours are
 * `struct cp_ent` = {pos.x,pos.z,vel.x,vel.z,radius} (5 floats). */
void pfref_clearpath_new_velocity(const float ent[5], const float des_v[2],
                                  const float *dyn, int n_dyn,
                                  const float *stat, int n_stat, float out[2]);

/* --- spatial index ------------------------------------------------------ */

/* bg_ent_init + inserts in index order + bg_ent_cleanup (position.c:283,
 * bitmap_grid.h:1102,1477), then G_Pos_EntsInCircleFrom (position.c:379) for
 * each query.  out_counts[q], out_ids[q*maxout + k]. */
void pfref_spatial_query(float xmin, float xmax, float zmin, float zmax,
                         const float *pos_xz, int n,
                         const float *query_xz, int nq, float range, int maxout,
                         int32_t *out_counts, uint32_t *out_ids);

/* --- movement tick (velocity half + position accept) -------------------- */

typedef struct pfref_move_world {
    int            n;             /* agents; uid == index */
    const float   *pos_xz;        /* [n][2]  */
    const float   *vel_xz;        /* [n][2]  movestate.velocity */
    const float   *radius;        /* [n]     selection radius   */
    const float   *max_speed;     /* [n]     movestate.max_speed */
    const float   *speed;         /* [n]     move_work_in.speed  */
    const uint32_t*flags;         /* [n]     ENTITY_FLAG_*       */
    const int32_t *state;         /* [n]     enum arrive_state   */
    const int32_t *flock;         /* [n]     flock index or -1   */
    const uint8_t *has_dest_los;  /* [n] */
    int            n_flocks;
    const float   *flock_target_xz; /* [n_flocks][2] */
    const uint32_t*flock_dest_id;   /* [n_flocks]    */
    int            hz;            /* 20/10/5/1 */
} pfref_move_world;

/* Drives movement.c's own static move_velocity_work (movement.c:3395) over
 * agents [begin,end) after loading `world` into the movement module's
 * snapshot tables, with ent_des_v taken from `vdes` (or, when vdes == NULL,
 * from N_DesiredPointSeekVelocity).  out_vel[n][2], out_vpref optional. */
int  pfref_move_load(pfref_nav *nav, const pfref_move_world *world);
void pfref_move_velocity(const float *vdes, int begin, int end, float *out_vel);
/* formation inputs of every work item: fstate.assignment_ready, cell_pos, fstate.normal_*_force */
void pfref_move_set_formation(const uint8_t *ready, const float *cell_pos, const float *cohesion,
                              const float *align, const float *drag);
/* entity_compute_update with movestate.next_rot as an input (the heading gate, movement.c:2319-2336);
 * dir_quat_from_velocity (:1411); adjacent_settled_count (:982); G_Arrival_ShouldSettle (arrival.c:946) against one
 * zone given as plain arrays (see ref_move.c) */
void pfref_move_heading_gate(const float *new_vel, const float *vdes, const float *next_rot, int begin, int end,
                             uint8_t *out_turn, float *out_vel);
void pfref_move_dir_quat(const float *heading_xz, int n, float *out_quat);
void pfref_move_settled_count(const int32_t *uids, int nq, int32_t *out);
int pfref_arrival_should_settle(pfref_nav *nav, int layer, const float *centre_xz, int radius, float unit_radius,
                                float fill_frac, int active_row, int num_rows,
                                const float *slots_xz, const int32_t *slot_ring, int num_slots,
                                const float *region_xz, int num_region_pos, uint64_t *out_keys,
                                int nq, const float *new_pos_xz, const float *vel_xz, const float *radius_of,
                                const int32_t *nsettled, uint8_t *substate, const uint8_t *sink_valid,
                                const float *sink_xz, const float *order_pos_xz, float *progress_anchor_xz,
                                uint8_t *progress_anchored, int32_t *stuck, uint8_t *out_settle);
/* a real arrival zone for (flock, layer) from plain arrays; every unit's struct arrival_unit_state set / read back;
 * counters of the binding's settle pass (move_hip_settle_stats) */
int pfref_move_set_arrival_zone(int flock, int layer, const float *centre_xz, int radius, float unit_radius,
                                float fill_frac, int active_row, int num_rows, const float *slots_xz,
                                const int32_t *slot_ring, int num_slots, const float *region_xz, int num_region_pos);
void pfref_move_set_arrival_units(const uint8_t *substate, const uint8_t *sink_valid, const float *sink_xz,
                                  const float *order_pos_xz, const float *progress_anchor_xz,
                                  const uint8_t *progress_anchored, const int32_t *stuck);
void pfref_move_get_arrival_units(uint8_t *substate, float *progress_anchor_xz, uint8_t *progress_anchored, int32_t *stuck);
void pfref_move_hip_settle_stats(long out[4]);
long pfref_move_hip_wait_differ(void);
double pfref_move_hip_state_work_seconds(void);   /* wall time of the last move_hip_state_work */
void pfref_move_hip_state_times(double out[6]);   /* snapshot, per-unit inputs, flock queries, navhip_state_pass, settle pass, scatter */
/* inputs of the flag / counter arms of the state switch (see ref_move.c); out_flags of pfref_move_state_update(_hip)
 * carry bit 2 = UPDATE_SET_MOVING (out_state = next_state then too) and bit 3 = UPDATE_SET_TARGET_DIR */
void pfref_move_set_state_aux(const uint8_t *fstate, const int32_t *wait_ticks_left, const uint8_t *wait_prev);
void pfref_move_get_wait_ticks(int32_t *out);
void pfref_move_set_turning(const float *ent_rot, const float *target_dir);
void pfref_move_hip_resident_state_pass(int on);   /* the state pass on the velocity pass's device-resident snapshot */
long pfref_move_hip_resident_passes(void);
void pfref_move_get_out(float *vel, float *vdes);  /* s_move_work.out[].ent_vel / .ent_des_v */
long pfref_move_hip_surround_differ(void);   /* surround positions of the device's pass that differ from the reference's store */
void pfref_move_set_surround(const int32_t *target_uid, const float *target_prev_xz, const float *nearest_prev_xz);
void pfref_move_get_surround(float *target_prev_xz, float *nearest_prev_xz, float *next_dest_xz);
void pfref_move_surround_queries(const float *new_vel, uint8_t *out_query, float *out_dest_xz);
void pfref_move_set_next_rot(const float *next_rot);      /* NULL: facing on the heading */
void pfref_move_set_interp(const float *next_pos_xz, const float *step);
void pfref_move_set_range_targets(const int32_t *target_uid, const float *target_range, const float *target_prev_xz);
/* fine-arrival inputs: sink [n][2], flags [n] (bit 0 unit committed to a valid slot, bit 1 the
 * flock's arrival_state for the unit's layer is in ARRIVAL_PHASE_FILLING) */
void pfref_move_set_arrival(const float *sink_xz, const uint8_t *flags);
void pfref_move_unload(void);
/* individual steering terms for unit tests (movement.c:1546,1653,1690,1870,2768) */
void pfref_move_vpref(int uid, const float vdes[2], float out[2]);
void pfref_move_forces(int uid, const float vdes[2], float out_arrive[2], float out_cohesion[2],
                       float out_separation[2]);
int  pfref_move_neighbours(int uid, float *out_dyn, int *n_dyn, float *out_stat, int *n_stat);
void pfref_move_get_vdes(float *out_vdes);
int  pfref_move_flock_order(int flock, uint32_t *out_uids);
/* entity_compute_update (movement.c:2303) per work item; see ref_move.c */
void pfref_move_state_update(const float *new_vel, const float *vdes, int begin, int end, uint8_t *out_state,
                             uint8_t *out_flags);
int  pfref_closest_pathable(pfref_nav *nav, int layer, float x, float z, float out[2]);
int  pfref_dest_island_tiles(pfref_nav *nav, int layer, float x, float z, int16_t *out_abs, int max_tiles);
double pfref_move_bench(const float *vdes, int begin, int end, int reps, int nthreads, float *out_vel);

#ifdef __cplusplus
}
#endif
#endif
