// agent_types.h -- plain-data views used by the agent-step kernels.  No HIP types beyond float2 /
// float4, so the per-thread bodies (agent_math.h, agent_thread.h) also compile with g++ for the
// host-side unit tests under tests/hostsim (test infrastructure; the product has no CPU path).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "map_view.h"

#ifdef NH_HOSTSIM
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
#endif

// Compact per-entity bits of the pool record (recA.w): the ENTITY_FLAG_* bits the movement tick
// reads (entity.h:56-82) + what find_neighbours needs to know about a neighbour, + the uid.
#define NH_PB_MOVABLE     0x01u    /* ENTITY_FLAG_MOVABLE                                          */
#define NH_PB_AIR         0x02u    /* ENTITY_FLAG_AIR                                              */
#define NH_PB_GARRISONED  0x04u    /* ENTITY_FLAG_GARRISONED                                       */
#define NH_PB_STATIC      0x08u    /* find_neighbours (movement.c:2817): ent_still || |vel| < 0.3 || at_slot */
#define NH_PB_WATER       0x10u    /* ENTITY_FLAG_WATER                                            */
#define NH_PB_IDLE        0x20u    /* no work item: still state or ENTITY_FLAG_COMBAT_HELD         */
#define NH_PB_UID_SHIFT   8        /* uid in bits 8..31: n_ents <= 2^24                            */

// bg_<name>_t geometry (bitmap_grid.h:959-990) + the cell-sorted element pool.  Pool slot k holds
// one inserted entity; the slots of a cell are contiguous and in the order bg_insert(uid 0..n-1) +
// bg_cleanup leave behind (descending uid, bitmap_grid.h:1102-1121,1515-1521).
struct nh_grid {
    int32_t origin_x, origin_y;      // BG_SCALE_F(xmin), BG_SCALE_F(ymin)
    int     grid_w, grid_h;          // ceil(span / 16 wu)
    int     n;                       // entities in the snapshot (pool slots <= n)
    const int32_t *cell_start;       // [ncells+1]
    const float4  *recA;             // [npool] {pos.x, pos.z, radius, NH_PB_* | uid << 8}
    const float2  *recV;             // [npool] movestate.velocity
    const int32_t *pool_of;          // [n] uid -> pool slot, -1 = not inserted (outside the slab filter)
    const int32_t *active, *n_active;    // a slab step only: the pool slots of the entities with a work item, any order
};

// structure-of-arrays sources of the pool records (all null: positions only)
struct nh_pack_src {
    const float    *vel_xz, *radius;
    const uint32_t *flags;
    const uint8_t  *state;
    const float    *arrival_sink_xz;
    const uint8_t  *arrival_flags;
};

struct nh_step_params {
    nh_map_view map;
    nh_grid     grid;
    float       map_x, map_z;
    int         n_ents, n_flocks, hz;
    int         n_members;           // upper bound of flock_offsets[n_flocks] (launch size)
    int         work_begin, work_end;
    int         members_key;         // what a lane grouping of the flocks' members is valid for (see k_coh_bin)
    const float    *pos_xz, *vel_xz, *radius, *max_speed, *speed;
    const uint32_t *flags;
    const uint8_t  *state, *has_dest_los;
    const int32_t  *flock;
    const float    *vdes_xz;
    const float    *flock_target_xz;
    const int32_t  *flock_offsets, *flock_members, *flock_field_slot;
    const uint8_t  *field_pool;
    const uint8_t  *form_ready;
    const float    *cell_pos_xz, *form_cohesion_xz, *form_align_xz, *form_drag_xz;
    const float    *arrival_sink_xz;
    const uint8_t  *arrival_flags;
    const uint8_t  *los_pool;        // per-agent has_dest_los from the device LOS pool (optional)
    const int32_t  *flock_los_slot;
    const float    *los_pos_xz;
    const int32_t  *region_row;      // per-entity mapping row of a region field (enemy seek / surround), -1 = none
    const int32_t  *region_field_slot;
};

struct nh_step_outs {
    float   *vel_xz, *new_pos_xz, *vdes_xz, *vpref_xz;
    uint8_t *status;
};

// What k_agent_nbr (pool order, needs only the spatial hash) leaves for the rest of the step,
// indexed by uid:
//   sep[uid]      separation_force (movement.c:1690), already truncated
//   cnt[uid]      n_dyn | n_stat << 8 | NH_NB_* << 16
//   rec[uid][j]   the ClearPath neighbours THEMSELVES -- struct cp_ent {pos.x, pos.z, vel.x, vel.z, radius},
//                 five floats -- j = 0..31 dynamic, 32..63 static (velocity 0, movement.c:2820), in
//                 find_neighbours order.  The walk has the pool record of every hit in registers anyway;
//                 the ClearPath kernels read an entity's row back as ONE contiguous run (entity-major)
//                 instead of chasing pool slots: a 16-byte record + an 8-byte velocity per neighbour, a
//                 64-byte sector each, were three quarters of the bytes those kernels fetched
#define NH_NB_IRREGULAR 0x1u   /* garrisoned hit / wide query: the wave-per-agent path redoes the gather */
#define NH_NB_DONE      0x2u   /* the walk ran for this entity this tick                                  */
struct nh_nbr {
    float2   *sep;
    uint32_t *cnt;
    float    *rec;
    int       stride;          // floats per entity (= 64 * 5)
};

// Work lists filled on the device (counters[NH_WL_*] + ids), consumed by fixed-size launches that
// stride over them: agents that still need a ClearPath search after k_agent_mid.
enum { NH_WL_ROW0 = 0, NH_WL_ROW1, NH_WL_ROW2, NH_WL_ROW3,   // a row of 16 lanes per agent: 1-2, 3-4, 5-8, 9-16 neighbours
       NH_WL_WAVE,          // one workgroup per agent: 17-32 neighbours
       NH_WL_HEAVY,         // one workgroup per agent: 33-64 neighbours (started first)
       NH_WL_FULL,          // one wave per agent, whole step (irregular gather)
       NH_WL_RETRY,         // filled by k_cp_small: 1-4 neighbours and no admissible candidate (the retry logic)
       NH_WL_LISTS };       // (number of lists)
// Every list is kept as NH_WL_SUB sub-lists, one per group of producer waves (wave index mod
// NH_WL_SUB): an append is one atomic per wave and list, and a few thousand atomics on ONE address
// serialise at ~5 ns each (measured: they were most of k_agent_mid's time).
#define NH_WL_SUB 64
struct nh_worklists {
    int32_t *count;            // [NH_WL_LISTS][NH_WL_SUB] entries + the ticket counters of the ClearPath
                               // kernels, one per 128-byte line: [0] workgroup problems, [1 + s] stripe s
                               // of the row units, [1 + NH_CP_STRIPES + s] of the retry launch (+ the same again:
                               // other parity)
    int32_t *ids;              // [NH_WL_LISTS][NH_WL_SUB][cap] uids
    int      cap;              // entries a sub-list can hold (its producers cannot append more)
};
#define NH_CP_STRIPES 32
#define NH_WL_COUNTERS (NH_WL_LISTS * NH_WL_SUB + 32 * (2 + 2 * NH_CP_STRIPES))

// per-entity record k_agent_mid leaves for the list consumers (32 bytes, indexed by uid)
struct nh_mid_rec {
    float    vpref[2];     // preferred velocity (ClearPath's des_v)
    float    vdes[2];
    float    arrive[2];    // NH_WL_FULL only: the arrive term of the steering force
    uint16_t probes;       // NH_WL_FULL only: probe_tiles_bits
    uint8_t  status;
    uint8_t  mode;
    float    vel_cap;      // max_speed / hz  (movement.c:3464)
};
