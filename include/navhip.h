/* include/navhip.h -- C ABI of libnavhip.so
 *
 * MI355X (gfx950) implementation of permafrost-engine's per-tick navigation
 * hot path: chunk flow-field build (integration sweep + 8-neighbour bake) and
 * the per-agent steering + ClearPath velocity step, behind plain-C entry
 * points that the reference's C host code binds directly (see INTEGRATION.md
 * for the reference-side call sites).  No torch / C++ types cross this
 * boundary: plain pointers, sizes and PODs only.
 *
 * Every entry point cites the reference interface (file:line under the
 * reference's src/) that it replaces or is fed by.
 *
 * Conventions
 *   - all functions return NAVHIP_OK (0) or a negative NAVHIP_ERR_* code; on a
 *     non-zero return the caller falls back to the reference CPU path
 *     (SURVEY.md §8b "Errors");
 *   - "host" pointers are ordinary CPU memory, "dev" pointers are HIP device
 *     memory on the context's GPU; `stream` is a hipStream_t passed as void*
 *     (NULL = the context's own stream);
 *   - plane layouts are the reference's packed upload layouts
 *     (N_CopyCostBasePacked / N_CopyBlockersPacked, nav.c:2432,2470): for one
 *     layer, chunk-row-major, each chunk a raw [64][64] array.
 */
#ifndef NAVHIP_H
#define NAVHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAVHIP_OK                 0
#define NAVHIP_ERR_INVALID       -1   /* bad argument                         */
#define NAVHIP_ERR_DEVICE        -2   /* HIP runtime error (see last_error)   */
#define NAVHIP_ERR_NOMEM         -3
#define NAVHIP_ERR_NOT_UPLOADED  -4   /* a plane the request needs is missing */

#define NAVHIP_FIELD_RES         64          /* FIELD_RES_R/C      nav_data.h:45-46 */
#define NAVHIP_FIELD_CELLS       4096
#define NAVHIP_COST_IMPASSABLE   0xff        /* COST_IMPASSABLE    nav_data.h:47 */
#define NAVHIP_ISLAND_NONE       0xffff      /* ISLAND_NONE        nav_data.h:48 */
#define NAVHIP_FACTION_ID_NONE   0xf         /* FACTION_ID_NONE    nav_data.h:49 */
#define NAVHIP_MAX_FACTIONS      15          /* MAX_FACTIONS       game.h:48     */
#define NAVHIP_NAV_LAYER_MAX     12          /* enum nav_layer     nav.h:78-92   */

/* enum flow_dir, nav.h:94-104 */
enum { NAVHIP_FD_NONE = 0, NAVHIP_FD_NW, NAVHIP_FD_N, NAVHIP_FD_NE, NAVHIP_FD_W,
       NAVHIP_FD_E, NAVHIP_FD_SW, NAVHIP_FD_S, NAVHIP_FD_SE };

/* field_target.type values, field.h:85-98 (only the chunk-aligned targets), plus the repair build
 * N_FlowFieldUpdateToNearestPathable (field.c:2247), which ignores the field's own target */
enum { NAVHIP_TARGET_PORTAL = 0, NAVHIP_TARGET_TILE = 1, NAVHIP_TARGET_NEAREST_PATHABLE = 2 };

/* per-layer planes held on the device (struct nav_chunk members, nav_data.h:118-158) */
enum {
    NAVHIP_PLANE_COST_BASE     = 0,   /* uint8_t  [64][64] per chunk              */
    NAVHIP_PLANE_BLOCKERS      = 1,   /* uint16_t [64][64] per chunk              */
    NAVHIP_PLANE_LOCAL_ISLANDS = 2,   /* uint16_t [64][64] per chunk              */
    NAVHIP_PLANE_FACTIONS      = 3,   /* uint8_t  [15][64][64] per chunk          */
    NAVHIP_PLANE_ISLANDS       = 4,   /* uint16_t [64][64] per chunk (global island ids,
                                         nav_chunk.islands nav_data.h:150; only read by
                                         NAVHIP_REQ_ISLAND_NEAREST)                */
    NAVHIP_PLANE_COUNT
};

/* navhip_field_req.flags */
#define NAVHIP_REQ_INOUT   0x1   /* update an existing field in place: unreachable cells keep
                                    the bytes already in the output slot (field.c:737-751,
                                    nav.c:1998-2008) instead of starting from N_FlowFieldInit */

#define NAVHIP_REQ_IF_CHANGED 0x2 /* incremental repair: build only when the request's chunk (or, for a
                                    portal target, the next chunk) is flagged changed by the blocker
                                    updates since the last navhip_clear_changed; otherwise the slot
                                    is left untouched (the cached field stays valid, the device image
                                    of N_ApplyDeferredInvalidations, nav.c:2208) */
#define NAVHIP_REQ_LIVE_IIDS  0x4 /* portal targets: re-read port_iid / next_iid on the device from the
                                    CURRENT local_islands plane -- the label of the first tile of the
                                    port / next portal that has one (a blocker leaves ISLAND_NONE); a
                                    portal blocked from end to end leads nowhere: the slot is left
                                    untouched (for request lists that outlive a relabel, as in the
                                    device-resident incremental-repair benchmark) */

#define NAVHIP_REQ_ISLAND_NEAREST 0x8 /* N_FlowFieldUpdateIslandToNearest(aux_iid, ...) (field.c:2307) on
                                    an existing TILE/PORTAL field: the frontier is moved to the tiles
                                    of local island aux_iid nearest (Manhattan) to the target's own
                                    frontier; always in place.  The repair the sampler runs when an
                                    agent's tile is orphaned from its goal by blockers (nav.c:3540) */

/* One chunk-field build: the arguments of
 *   N_FlowFieldUpdate(chunk, priv, faction_id, layer, target, ctx, inout)   field.c:2030
 * with `struct field_target` / `struct portal_desc` (field.h:67-72,85-101) flattened so the
 * record is position independent (portal pointers become endpoint coordinates).  32 bytes. */
typedef struct navhip_field_req {
    uint8_t  layer;          /* enum nav_layer                                         */
    uint8_t  type;           /* NAVHIP_TARGET_TILE / NAVHIP_TARGET_PORTAL              */
    uint8_t  faction_id;     /* NAVHIP_FACTION_ID_NONE, or the pathing faction         */
    uint8_t  flags;          /* NAVHIP_REQ_*                                           */
    uint16_t enemies;        /* enemies_for_faction(faction_id) bitmask, field.c:166   */
    uint16_t chunk_r, chunk_c;
    uint8_t  tile_r, tile_c;                       /* TARGET_TILE: target.tile;
                                                      NEAREST_PATHABLE: the start tile  */
    uint8_t  port_r0, port_c0, port_r1, port_c1;   /* TARGET_PORTAL: pd.port->endpoints */
    uint8_t  next_r0, next_c0, next_r1, next_c1;   /*                pd.next->endpoints */
    uint16_t next_chunk_r, next_chunk_c;           /*                pd.next->chunk     */
    uint16_t port_iid, next_iid;                   /*                pd.port_iid/next_iid */
    uint16_t aux_iid;                              /* NAVHIP_REQ_ISLAND_NEAREST: local_iid */
    uint16_t _pad;
} navhip_field_req;

typedef struct navhip_ctx navhip_ctx;

/* ---- context / map state (SURVEY.md §8b "Map/state upload") ------------------------------ */

/* Create a context for a map of chunk_w x chunk_h chunks (struct nav_private width/height,
 * nav_private.h:53) on HIP device `device`.  Fails loudly (NAVHIP_ERR_DEVICE) when no GPU is
 * present: there is no CPU fallback inside this library. */
int  navhip_ctx_create(navhip_ctx **out, int chunk_w, int chunk_h, int device);
void navhip_ctx_destroy(navhip_ctx *ctx);
const char *navhip_last_error(const navhip_ctx *ctx);
int  navhip_device(const navhip_ctx *ctx);
/* the context's HIP stream, as void* (hipStream_t) */
void *navhip_stream(const navhip_ctx *ctx);
/* A stream (hipStream_t as void*; the library's, never to be destroyed by the caller) for wide, throughput-bound work --
 * the field builds -- that runs BESIDE the agent step enqueued on `main_stream`.  cu_count > 0: it may only use the
 * compute units [cu_begin, cu_begin + cu_count) (hipExtStreamCreateWithCUMask; the MI355X has 256 CUs in 8 XCDs of 32)
 * and leaves the rest of the chip to the narrow, latency-bound front of the step; cu_count <= 0: every compute unit.
 * The stream has a hardware queue of its own, on a pipe of the command processor that neither `main_stream` nor the
 * step's side streams use (a hand-over between two queues on one pipe costs 100-200 us instead of 12: measured once per
 * process and caller stream, csrc/navhip_api.hip).  One set of such streams exists per process and device: calls with
 * the same arguments return the same stream. */
int  navhip_stream_beside(navhip_ctx *ctx, void *main_stream, int cu_begin, int cu_count, void **out_stream);
/* The library's own stream for the agent chain (what navhip_tick_create runs a tick on when its description names no
 * stream): a hardware queue of its own on the fourth pipe, beside the three of navhip_stream_beside(ctx, <this stream>, ..)
 * and of the step's side streams.  A host that has no reason to run the step on a stream of its own should use this
 * one: the set is the same for every context of the process, so a tick costs the same whatever the process created before. */
int  navhip_stream_main(navhip_ctx *ctx, void **out_stream);
/* navhip_stream_beside(ctx, navhip_stream(ctx), cu_begin, cu_count, out_stream) with cu_count > 0 */
int  navhip_stream_create_partial(navhip_ctx *ctx, int cu_begin, int cu_count, void **out_stream);
/* waits for the context's own stream and for the side streams of navhip_agent_prefetch_dev */
int  navhip_sync(navhip_ctx *ctx);

/* Upload one whole plane of one layer (all chunks).  Replaces N_CopyCostBasePacked /
 * N_CopyBlockersPacked consumers (nav.c:2408-2490): same layout, same sizes. */
int  navhip_upload_plane(navhip_ctx *ctx, int layer, int plane, const void *host, size_t bytes);
/* Upload one chunk of one plane (dirty-chunk update after N_Update, nav.c:2119). */
int  navhip_upload_chunk(navhip_ctx *ctx, int layer, int plane, int chunk_r, int chunk_c,
                         const void *host, size_t bytes);
/* Device pointer of a resident plane (NULL when never uploaded); for zero-copy producers. */
void *navhip_plane_dev(navhip_ctx *ctx, int layer, int plane);

/* Read a resident plane back (tests; host mirrors after device-side blocker updates). */
int  navhip_download_plane(navhip_ctx *ctx, int layer, int plane, void *host, size_t bytes);

/* ---- dynamic obstacles -------------------------------------------- */

/* One N_BlockersIncref (delta = +1) / N_BlockersDecref (delta = -1) call, nav.c:4663,4685:
 * (xz_pos, range, faction_id, flags).  24 bytes. */
typedef struct navhip_circle {
    float    x, z;          /* xz_pos                                                      */
    float    radius;        /* range; the device path takes ceil(radius/4) <= 28 tiles     */
    int32_t  faction_id;    /* 0..14                                                       */
    uint32_t flags;         /* ENTITY_FLAG_AIR selects the air layers, else water + ground */
    int32_t  delta;         /* +1 incref, -1 decref                                        */
} navhip_circle;

/* Apply n incref/decref calls to the resident blockers (+ factions) planes of every resident
 * layer of the affected domains, with the reference's tile sets: tiles under the circle
 * (M_Tile_AllUnderCircle, tile.c:687) on the 1x1 layer, plus 1/2/3 successive contours
 * (M_Tile_Contour, tile.c:759) on the 3x3/5x5/7x7 layers (n_update_blockers_circle_*,
 * nav.c:1051-1133).  Afterwards, on the device and without a host round trip: the derived
 * passability of every touched chunk is rebuilt, chunks whose passability CHANGED are flagged
 * (the dirty_chunks set of n_update_blockers, nav.c:1033-1046, restricted to real changes) and
 * their local-island labels recomputed (n_update_local_islands, nav.c:967).
 * map_pos_x/z: the vec3 map_pos .x/.z handed to every N_* call. */
int  navhip_blockers_circles(navhip_ctx *ctx, const navhip_circle *circles, int n,
                             float map_pos_x, float map_pos_z);
int  navhip_blockers_circles_dev(navhip_ctx *ctx, const navhip_circle *dev_circles, int n,
                                 float map_pos_x, float map_pos_z, void *stream);
/* n_update_local_island_field (nav.c:986) for one layer on the device, from the resident
 * cost_base + blockers planes (allocates the local_islands plane when it was never uploaded). */
int  navhip_relabel_local_islands(navhip_ctx *ctx, int layer);
/* The changed-chunk flags of one layer ([chunk_h*chunk_w] bytes, 1 = passability changed);
 * clear != 0 resets them afterwards. */
int  navhip_changed_chunks(navhip_ctx *ctx, int layer, uint8_t *host_flags, int clear);
int  navhip_clear_changed(navhip_ctx *ctx, void *stream);

/* ---- chunk flow fields ---------------------------------------- */

/* Build n chunk fields.  Replaces the fiber fan-out of field_task (nav.c:2049-2060) joined by
 * N_AwaitAsyncFields (nav.c:3958), and the serial N_FlowFieldInit+N_FlowFieldUpdate pairs inside
 * n_request_path (nav.c:1830-1831,2016-2017).
 *   reqs        host, n records
 *   inout_dirs  host, n * 4096 bytes; slot i is `struct flow_field.field` of request i
 *               (one dir_idx 0..8 per byte, row-major [64][64]).  Read only for requests with
 *               NAVHIP_REQ_INOUT, always written.
 *   out_integ   host, n * 4096 floats or NULL: the integration field (field.c:2059-2077),
 *               INFINITY where unreached.  Debug/parity output; the reference never exposes it. */
int  navhip_build_fields(navhip_ctx *ctx, const navhip_field_req *reqs, int n,
                         uint8_t *inout_dirs, float *out_integ);
/* Same with every buffer resident on the device (no PCIe in the call). */
int  navhip_build_fields_dev(navhip_ctx *ctx, const navhip_field_req *dev_reqs, int n,
                             uint8_t *dev_inout_dirs, float *dev_out_integ, void *stream);

/* ---- region flow fields ---------------------------------------------------- */

/* One flow field over a square region of nav tiles that may straddle chunks (struct region,
 * field.c; field_build_integration_region field.c:582).  Seeds and overlay-blocked tiles are
 * (abs_r, abs_c) int16 pairs in absolute nav-tile coordinates (chunk * 64 + tile), handed over in
 * two shared arrays.  32 bytes.
 *   out_mode 0  whole region, two 4-bit directions per byte (field_build_flow_unaligned :800):
 *               N_CellArrivalFieldCreate :2445 / N_GroupArrivalFieldCreate :2525; the caller
 *               computes base_abs_* exactly as those functions do (:2475-2488 / :2558-2559)
 *   out_mode 1  the 64x64 window at (roff, coff) of the region written IN PLACE into a chunk field
 *               of one direction per byte (field_build_flow_region :763): field_update_enemies
 *               :1537 / _entity :1615 / _zone :1822, whose frontiers the game side provides */
typedef struct navhip_region_req {
    uint8_t  layer;
    uint8_t  out_mode;
    uint16_t enemies;                 /* 0: field_tile_passable, else ..._no_enemies(enemies)   */
    int16_t  base_abs_r, base_abs_c;  /* region.base; may lie outside the map                   */
    uint16_t rdim, cdim;              /* even, rdim == cdim <= 128                              */
    uint16_t roff, coff;              /* out_mode 1                                             */
    uint32_t seed_begin, seed_count;
    uint32_t overlay_begin, overlay_count;
} navhip_region_req;

/* out: n slots of out_stride bytes (>= rdim*cdim/2 for mode 0, >= 4096 for mode 1).  Mode 0
 * slots are fully written; mode 1 slots are read-modify-write (unreached tiles keep their byte). */
int  navhip_build_region_fields(navhip_ctx *ctx, const navhip_region_req *reqs, int n,
                                const int16_t *seeds, size_t n_seeds,
                                const int16_t *overlay, size_t n_overlay,
                                uint8_t *inout, size_t out_stride);
int  navhip_build_region_fields_dev(navhip_ctx *ctx, const navhip_region_req *dev_reqs, int n,
                                    int max_dim, const int16_t *dev_seeds,
                                    const int16_t *dev_overlay, uint8_t *dev_inout,
                                    size_t out_stride, void *stream);

/* ---- line-of-sight fields ------------------------------------------------- */

/* One N_LOSFieldCreate(id, chunk_coord, target, priv, map_pos, ctx, out_los, prev_los) call
 * (field.c:2085).  16 bytes.  layer / faction_id are N_DestLayer(id) / N_DestFactionID(id)
 * (nav.c:5052,5057); prev_dr/prev_dc = prev_los->chunk - chunk_coord (0,0: the destination
 * chunk, no previous field). */
typedef struct navhip_los_req {
    uint8_t  layer, faction_id;
    uint16_t enemies;                       /* enemy_faction_from(faction_id), field.c:153 */
    uint16_t chunk_r, chunk_c;
    uint16_t target_chunk_r, target_chunk_c;
    uint8_t  target_tile_r, target_tile_c;
    int8_t   prev_dr, prev_dc;
} navhip_los_req;

/* Build n LOS fields.  A field is 4096 bytes, one per tile, row-major [64][64]:
 * bit 0 = visible, bit 1 = wavefront_blocked (struct LOS_field, field.h:48-54).
 *   prev_fields  host, n * 4096 bytes (slot i = prev_los of request i; ignored for requests
 *                without a previous field) or NULL when no request has one
 *   out_fields   host, n * 4096 bytes
 * Bit-exact with the reference, including the pop order of its binary heap (pqueue.h). */
int  navhip_build_los(navhip_ctx *ctx, const navhip_los_req *reqs, int n,
                      const uint8_t *prev_fields, uint8_t *out_fields,
                      float map_pos_x, float map_pos_z);
int  navhip_build_los_dev(navhip_ctx *ctx, const navhip_los_req *dev_reqs, int n,
                          const uint8_t *dev_prev_fields, uint8_t *dev_out_fields,
                          float map_pos_x, float map_pos_z, void *stream);

/* N_FlowFieldID (field.c:1952) for TILE / PORTAL targets: the 64-bit cache key the reference's
 * fieldcache uses; pure bit packing, host side. */
uint64_t navhip_flow_field_id(const navhip_field_req *req);
/* ... and for the region targets (field.h:85-101), whose fields navhip_build_region_fields writes
 * (out_mode 1): kind = the reference's field_target.type,
 *   NAVHIP_FFID_ENEMIES  a = enemies.faction_id
 *   NAVHIP_FFID_ENTITY   a = ent.target (uid)
 *   NAVHIP_FFID_ZONE     a, b = zone.centre in absolute nav tiles (chunk * 64 + tile: row, column),
 *                        c = zone.radius
 * 0 for any other kind. */
enum { NAVHIP_FFID_ENEMIES = 2, NAVHIP_FFID_ENTITY = 4, NAVHIP_FFID_ZONE = 5 };
uint64_t navhip_region_field_id(int kind, int layer, int chunk_r, int chunk_c, uint32_t a, int b, int c);

/* kernel selection override for tests/bench: 0 = auto (bit-parallel BFS when every passable
 * cell of the chunk has cost 1, generic relaxation otherwise), 1 = force generic. */
int  navhip_set_field_kernel(navhip_ctx *ctx, int mode);

/* ---- resident flow-field pool (SURVEY.md §8b "Ownership" / the reference's field cache) ------- */

/* A C host has no device pointers: without a device-resident pool every host-buffer call would move
 * the field cache over PCIe again.  The pool keeps chunk fields in HBM under the reference's own
 * 64-bit flow-field ids (N_FlowFieldID, field.c:1952) and mirrors the operations of the reference's
 * cache (fieldcache.c): put / contains / (dest, chunk) -> field mapping; least-recently-used slots
 * are recycled like lru_flow_put (fieldcache.c:411), and mappings that point at a recycled slot drop
 * to "no field".  n_dests = rows of the mapping table; the agent step uses the FLOCK INDEX as row.
 * Called from the nav-tick task only, like the reference's cache (FC_ASSERT_NAV_TASK). */
int  navhip_pool_create(navhip_ctx *ctx, int n_slots, int n_dests);
void navhip_pool_destroy(navhip_ctx *ctx);
int  navhip_pool_clear(navhip_ctx *ctx);                                  /* N_FC_ClearAll              */
int  navhip_pool_contains(navhip_ctx *ctx, uint64_t ff_id);               /* N_FC_ContainsFlowField     */
int  navhip_pool_put(navhip_ctx *ctx, uint64_t ff_id, const uint8_t *dirs);   /* N_FC_PutFlowField: 4096
                                                                              direction bytes from the host */
int  navhip_pool_get(navhip_ctx *ctx, uint64_t ff_id, uint8_t *out_dirs); /* N_FC_FlowFieldAt (a copy)  */
/* lru_flow_remove (the N_FC_Invalidate* family, fieldcache.c:253,520,580) and the eviction of the
 * host cache: the field is dropped and every mapping that points at it becomes "no field" */
int  navhip_pool_invalidate(navhip_ctx *ctx, uint64_t ff_id);
/* Batched N_FlowFieldInit + N_FlowFieldUpdate + N_FC_PutFlowField: request i is built INTO the pool
 * slot of ff_ids[i].  For a NAVHIP_REQ_INOUT request base_ids[i] names the resident field the update
 * starts from (the memcpy of nav.c:1998, the repairs of nav.c:3527-3547); 0 or ff_ids[i] itself = the
 * slot's own content; base_ids may be NULL.  Requests that depend on an earlier request of the same
 * call are ordered behind it.  out_dirs: n * 4096 bytes copied back, or NULL. */
int  navhip_pool_build(navhip_ctx *ctx, const navhip_field_req *reqs, const uint64_t *ff_ids,
                       const uint64_t *base_ids, int n, uint8_t *out_dirs);
/* N_FC_PutDestFFMapping, batched: (dest row, chunk) -> ff_id; an id that is not resident maps to
 * "no field" (the step then reports NAVHIP_ST_FIELD_MISS for agents on that chunk). */
int  navhip_pool_map(navhip_ctx *ctx, int n, const int32_t *dest, const uint16_t *chunk_r,
                     const uint16_t *chunk_c, const uint64_t *ff_ids);
/* navhip_world.n_field_slots value that makes the agent step sample the context's resident pool
 * (field_pool / flock_field_slot are then ignored) */
#define NAVHIP_POOL_RESIDENT (-1)

/* Page-locked host memory: arrays allocated here are transferred by the host-buffer entry points
 * without a staging copy (any other host memory works too, through the library's own pinned slab). */
void *navhip_host_alloc(size_t bytes);
void  navhip_host_free(void *p);

/* ---- per-agent movement step ---------------------------------- */

/* enum move_state, movement.c:113-143 */
enum { NAVHIP_STATE_MOVING = 0, NAVHIP_STATE_MOVING_IN_FORMATION, NAVHIP_STATE_ARRIVED,
       NAVHIP_STATE_SEEK_ENEMIES, NAVHIP_STATE_WAITING, NAVHIP_STATE_SURROUND_ENTITY,
       NAVHIP_STATE_ENTER_ENTITY_RANGE, NAVHIP_STATE_TURNING, NAVHIP_STATE_ARRIVING_TO_CELL };

/* ENTITY_FLAG_* bits the step reads, entity.h:56-82 */
#define NAVHIP_ENTITY_FLAG_MOVABLE      (1u << 3)
#define NAVHIP_ENTITY_FLAG_WATER        (1u << 14)
#define NAVHIP_ENTITY_FLAG_AIR          (1u << 15)
#define NAVHIP_ENTITY_FLAG_GARRISONED   (1u << 18)
#define NAVHIP_ENTITY_FLAG_COMBAT_HELD  (1u << 21)

/* navhip_step_out.status bits */
#define NAVHIP_ST_MOVED        0x01  /* pos+vel accepted by the position test (movement.c:2356-2358) */
#define NAVHIP_ST_FIELD_MISS   0x02  /* no flow field cached for the agent's chunk: the host must run
                                        the planner (n_request_path, nav.c:3483-3492) and re-step    */
#define NAVHIP_ST_FIELD_NONE   0x04  /* the field has FD_NONE under the agent (nav.c:3495-3554 path)  */
#define NAVHIP_ST_LOS_MISS     0x08  /* NAVHIP_LOS_LOOKUP, but no LOS field is mapped for the agent's chunk   */
#define NAVHIP_ST_UNSUPPORTED  0x80  /* formation state without formation inputs: velocity = 0        */

/* The snapshot the movement tick works on: `struct move_gamestate` + `struct move_work_in` +
 * flock table (movement.c:207-213,264-276,296-312) as structure-of-arrays; uid == array index.
 * All pointers are HOST pointers for navhip_agent_step and DEVICE pointers for
 * navhip_agent_step_dev. */
typedef struct navhip_world {
    int32_t  n_ents;                 /* every entity in the position snapshot                  */
    int32_t  n_flocks;
    int32_t  hz;                     /* movement tick rate: 20/10/5/1 (hz_count, movement.c:2210) */
    int32_t  n_field_slots;          /* slots of field_pool, or NAVHIP_POOL_RESIDENT                */
    const float    *pos_xz;          /* [n][2]  G_Pos_GetXZFrom                                */
    const float    *vel_xz;          /* [n][2]  movestate.velocity (world units per tick)      */
    const float    *radius;          /* [n]     G_GetSelectionRadiusFrom                       */
    const float    *max_speed;       /* [n]     movestate.max_speed                            */
    const float    *speed;           /* [n]     move_work_in.speed                             */
    const uint32_t *flags;           /* [n]     G_FlagsGetFrom                                 */
    const uint8_t  *state;           /* [n]     movestate.state                                */
    const uint8_t  *has_dest_los;    /* [n]     move_work_in.has_dest_los                      */
    const int32_t  *flock;           /* [n]     index into the flock table, -1 = none          */
    const float    *vdes_xz;         /* [n][2]  move_work_in.ent_des_v, or NULL: sample the flow
                                                fields on the device (N_DesiredPointSeekVelocity).
                                                With a non-NULL array, an entry whose x is NaN is
                                                sampled on the device as well                       */
    const float    *flock_target_xz; /* [F][2]  flock.target_xz                                */
    const int32_t  *flock_offsets;   /* [F+1]   CSR offsets into flock_members                 */
    const int32_t  *flock_members;   /* member uids, in kh_foreach order of flock.ents         */
    const int32_t  *flock_field_slot;/* [F][chunks] slot of the (dest,chunk) flow field in
                                                field_pool, -1 = not cached (N_FC_GetDestFFMapping) */
    const uint8_t  *field_pool;      /* [slots][4096] flow fields (one dir_idx per byte)       */
    float    map_pos_x, map_pos_z;   /* vec3 map_pos .x/.z handed to every N_* call            */
    float    grid_xmin, grid_xmax, grid_zmin, grid_zmax;  /* bg_ent_init bounds, position.c:276-283 */
    int32_t  work_begin, work_end;   /* the slab [begin,end) of uids this call computes (the index
                                        slabs of move_submit_cpu_work, movement.c:3759-3762); every
                                        entity still acts as a neighbour (internally only entities
                                        within reach of the slab's queries are indexed, which does
                                        not change any result).  0,0 = all entities               */
    /* Formation inputs, computed by the host's formation module (struct formation_state,
     * movement.c:215-225; move_work_in.cell_pos, :268).  form_ready == NULL: entities in
     * STATE_MOVING_IN_FORMATION / STATE_ARRIVING_TO_CELL are reported NAVHIP_ST_UNSUPPORTED. */
    const uint8_t  *form_ready;      /* [n]     fstate.assignment_ready                          */
    const float    *cell_pos_xz;     /* [n][2]  move_work_in.cell_pos                            */
    const float    *form_cohesion_xz;/* [n][2]  fstate.normal_cohesion_force                     */
    const float    *form_align_xz;   /* [n][2]  fstate.normal_align_force                        */
    const float    *form_drag_xz;    /* [n][2]  fstate.normal_drag_force                         */
    /* Fine-arrival inputs (struct arrival_unit_state, arrival.h:105-128, kept per unit in
     * movestate.arrival; struct arrival_state per flock and nav layer, arrival.h:66-91).  NULL (both):
     * the arrival overlay is inactive for everyone.  The step reads them exactly where the reference
     * does: the seek target of the arrive force is the unit's slot once bits 0 and 1 are set
     * (G_Arrival_SeekTarget, arrival.c:1034; movement.c:1751-1753,1887-1889), and a NEIGHBOUR with bit 0
     * set that stands within 1.5 radii of its slot is a static obstacle for ClearPath
     * (G_Arrival_NeighbourSettling, arrival.c:1042; find_neighbours, movement.c:2815-2817). */
    const float    *arrival_sink_xz; /* [n][2]  movestate.arrival.sink                           */
    const uint8_t  *arrival_flags;   /* [n]     bit 0: substate is SEEK / SEEK_ARMED (unit_committed,
                                                arrival.c:90) and sink_valid; bit 1: the flock's
                                                arrival_state for the unit's nav layer exists and is in
                                                ARRIVAL_PHASE_FILLING                                  */
    /* Per-agent line of sight on the device (compute_los_state movement.c:4129 -> N_HasDestLOS
     * nav.c:4026): an entry of has_dest_los equal to NAVHIP_LOS_LOOKUP is answered from the LOS fields
     * of navhip_build_los -- `visible` of the agent's tile in the (destination, chunk) field.  No field
     * mapped for the chunk: false, and NAVHIP_ST_LOS_MISS is reported so that the host can request the
     * path (nav.c:4041-4047).  All three NULL / 0: every entry of has_dest_los is taken as given. */
    const uint8_t  *los_pool;        /* [slots][4096] LOS fields (bit 0 visible, bit 1 wavefront_blocked) */
    const int32_t  *flock_los_slot;  /* [F][chunks] slot of the (dest, chunk) LOS field, -1 = none       */
    const float    *los_pos_xz;      /* [n][2]  the position the lookup uses: movestate.prev_pos
                                                (movement.c:4137), or NULL = pos_xz                    */
    int32_t  n_los_slots;
    /* Host-buffer calls (navhip_agent_step, _submit) only: a caller that knows that its FLOCK tables --
     * flock, flock_target_xz, flock_offsets, flock_members -- are the ones of its previous call on this
     * context (no entity was added or removed, no flock made, disbanded or re-targeted: G_Move_AddEntity
     * movement.c:4591, G_Move_RemoveEntity :4615, make_flock :789) passes the same nonzero epoch again and
     * those tables are not transferred a second time; 0 = transfer everything (the default).  Everything
     * else always travels: the per-tick state (pos, vel, speed, state, has_dest_los, vdes, formation and
     * arrival inputs) and the per-entity attributes that change without any of those events -- radius,
     * max_speed (MOVE_CMD_SET_MAX_SPEED, movement.c:3226) and flags (ENTITY_FLAG_GARRISONED is toggled
     * outside movement.c).
     * Device-resident calls (navhip_agent_step_dev, _prefetch_dev) read it for one purpose: a caller that steps a
     * uid SLAB (work_begin / work_end) and passes the same nonzero epoch again promises that flock_members is the
     * one of its previous call -- the cohesion term then reuses the lane grouping of the slab's members it made
     * last tick.  0 on a slab call = unknown: the grouping is rebuilt from the flock tables (slower, never wrong).
     * Steps over the whole snapshot do not need it. */
    uint32_t static_epoch;
    /* Region fields sampled on the device: N_DesiredEnemySeekVelocity (nav.c:3603) for STATE_SEEK_ENEMIES and
     * N_DesiredSurroundVelocity (nav.c:3687) for STATE_SURROUND_ENTITY entities that use the surround field
     * (ent_desired_velocity, movement.c:1478-1500).  Those return N_FlowDir of the ONE tile under the entity
     * in the chunk field of an ENEMIES / ENTITY target (navhip_build_region_fields, out_mode 1) -- no blend.
     * region_row[i] >= 0 names the entity's mapping row: a row of region_field_slot ([rows][chunks] slots of
     * field_pool, -1 = not cached), or -- with n_field_slots = NAVHIP_POOL_RESIDENT -- a row of the resident
     * pool's mapping table (navhip_pool_map; rows n_flocks.. are free for this); -1 = the entity samples its
     * flock's point-seek fields as above.  Like there, only entries whose vdes_xz is absent (NULL array or
     * NaN x) are sampled; a missing field / FD_NONE under the entity (the repair builds of nav.c:3652-3683,
     * :3733-3760) is reported as NAVHIP_ST_FIELD_MISS / _NONE.  Both NULL: no region sampling. */
    const int32_t  *region_row;        /* [n] */
    const int32_t  *region_field_slot; /* [n_region_rows][chunks] */
    int32_t  n_region_rows;
} navhip_world;
#define NAVHIP_LOS_LOOKUP 0xff

typedef struct navhip_step_out {
    float   *vel_xz;        /* [n][2]  move_work_out.ent_vel (movement.c:3462-3464); 0 for still ents */
    float   *new_pos_xz;    /* [n][2]  or NULL: pos+vel when accepted, else pos (movement.c:2336-2358) */
    float   *vdes_xz;       /* [n][2]  or NULL: the desired direction used (debug / parity)    */
    float   *vpref_xz;      /* [n][2]  or NULL: preferred velocity before ClearPath (debug / parity) */
    uint8_t *status;        /* [n]     or NULL: NAVHIP_ST_*                                     */
} navhip_step_out;

/* One velocity step for every non-still entity: replaces the WORK_TYPE_CPU / WORK_TYPE_GPU arms of
 * fork_join_velocity_computations (movement.c:4182-4194), i.e. move_velocity_work (:3395) for all
 * work items, plus the position accept test of entity_compute_update (:2336-2358).
 * Host-buffer form: uploads the snapshot, runs, downloads the outputs. */
int  navhip_agent_step(navhip_ctx *ctx, const navhip_world *world, const navhip_step_out *out);
/* Device-resident form (no PCIe in the call), asynchronous on `stream`. */
int  navhip_agent_step_dev(navhip_ctx *ctx, const navhip_world *dev_world,
                           const navhip_step_out *dev_out, void *stream);

/* Asynchronous host-buffer form (SURVEY.md §8b "Threading": the nav task must not block for a tick;
 * the GL path polls a fence once per frame, movement.c:4212-4233): submit stages the inputs through
 * pinned memory and returns at once; poll returns 1 while the step is running and 0 once the outputs
 * are in the caller's arrays (wait blocks until then).  One step in flight per context; the caller's
 * arrays must stay untouched between submit and completion. */
int  navhip_agent_step_submit(navhip_ctx *ctx, const navhip_world *world, const navhip_step_out *out);
int  navhip_agent_step_poll(navhip_ctx *ctx);
int  navhip_agent_step_wait(navhip_ctx *ctx);

/* Optional overlap: start the parts of the step that depend only on the snapshot (the spatial hash
 * and the O(N*F) cohesion term) on the context's own side streams, forked from `stream`, and return
 * at once.  Work enqueued on `stream` afterwards (e.g. the tick's field builds) then runs
 * concurrently with them; the next navhip_agent_step_dev on the same snapshot arrays joins the side
 * streams instead of recomputing.  Purely a scheduling hint: results are identical.
 * The side streams read the snapshot arrays until that step has completed on ITS stream (work the
 * caller enqueues there afterwards is ordered behind the last read); a prefetch that is never
 * followed by a step is drained by navhip_sync. */
int  navhip_agent_prefetch_dev(navhip_ctx *ctx, const navhip_world *dev_world, void *stream);
/* flags: NAVHIP_PREFETCH_FRONT_INLINE  the spatial hash and the neighbour walk stay on `stream` itself
 * (only the cohesion term forks): for a caller that enqueues nothing wide on `stream` between this call
 * and the step -- the chain then follows the previous step on the same stream without a hand-over. */
#define NAVHIP_PREFETCH_FRONT_INLINE 0x1u
/*        NAVHIP_PREFETCH_SNAPSHOT_HELD  the caller leaves the snapshot arrays of this call untouched until
 * its NEXT step on this context has been enqueued (it ping-pongs its position / velocity buffers): the
 * step then does not make the caller's stream wait for the lane regrouping the cohesion stream runs
 * for the next tick -- that work is ordered in front of the next step's cohesion term anyway. */
#define NAVHIP_PREFETCH_SNAPSHOT_HELD 0x2u
/*        NAVHIP_PREFETCH_FOLLOWS_STEP   every array of the snapshot was final when the last navhip_agent_step_dev
 * on `stream` ended, and nothing was enqueued on `stream` since that this call has to follow (a tick loop that
 * ping-pongs the outputs of one step into the snapshot of the next): the side streams then start behind a word the
 * step stored in device memory when it ended, instead of behind an event recorded on `stream` now -- one packet less
 * in front of the spatial hash.  Ignored when the last step did not run on `stream`. */
#define NAVHIP_PREFETCH_FOLLOWS_STEP 0x4u
int  navhip_agent_prefetch_dev_ex(navhip_ctx *ctx, const navhip_world *dev_world, void *stream, uint32_t flags);
/* Scheduling hint for a caller that runs other wide work (the field builds of the NEXT tick, say)
 * beside the agent step: make `stream` wait until the given stage of the step in flight is done, so
 * that the narrow, serial front of the step is not slowed down by it.
 *   NAVHIP_STAGE_START       the moment the last navhip_agent_prefetch_dev was forked from its stream
 *                            (= the previous step on that stream has finished)
 *   NAVHIP_STAGE_NEIGHBOURS  spatial hash + neighbour walk of the last navhip_agent_prefetch_dev
 *   NAVHIP_STAGE_LISTS       preferred velocities + work lists of the last navhip_agent_step_dev
 *   NAVHIP_STAGE_END         the last navhip_agent_step_dev has written its outputs (NAVHIP_ERR_INVALID when that step
 *                            ran on one stream, without side streams: its stream is its end -- order behind that)
 * (NAVHIP_ERR_INVALID when that call has not been made).  The wait is a one-lane kernel on `stream` that ends when the
 * stage's word in device memory has been stored -- no event, no packet on the step's own streams; 2-3 us from the
 * store to the next kernel on `stream` --, or an event wait where the step itself hands over
 * through events: in a jam, under rocprofv3 --pmc, with NAVHIP_HANDOVER=events in the environment. */
#define NAVHIP_STAGE_NEIGHBOURS 0
#define NAVHIP_STAGE_LISTS      1
#define NAVHIP_STAGE_START      2
#define NAVHIP_STAGE_END        3
int  navhip_stream_wait_stage(navhip_ctx *ctx, void *stream, int stage);

/* Per-kernel-group timing of the agent step with HIP events on the launch stream (bench.py's
 * roofline line).  A profiled navhip_agent_step[_dev] runs every kernel group back to back on the
 * caller's stream (no side streams); navhip_last_step_ms then returns, in milliseconds,
 * {spatial-hash build, neighbour walk (k_agent_nbr), k_cohesion, cohesion lane regrouping,
 *  k_agent_mid + the ClearPath searches} (it waits for the step). */
#define NAVHIP_STEP_PHASES 5
int  navhip_set_profiling(navhip_ctx *ctx, int on);
int  navhip_last_step_ms(navhip_ctx *ctx, float out_ms[NAVHIP_STEP_PHASES]);

/* Work counters of the context since creation (or the last reset): what SURVEY.md section 5 asks the shim
 * to expose so that a host can report cells/s and agent-steps/s with its own clock (one chunk field
 * = 4096 cells; host counters, no device synchronisation). */
typedef struct navhip_counters {
    uint64_t field_calls;     /* navhip_build_fields / _dev / pool builds                          */
    uint64_t chunk_fields;    /* chunk-field requests in them                                      */
    uint64_t step_calls;      /* navhip_agent_step / _dev / _submit                                */
    uint64_t agent_steps;     /* work items (work_end - work_begin) in them                        */
    uint64_t los_fields;      /* navhip_build_los requests                                         */
    uint64_t region_fields;   /* navhip_build_region_fields requests                               */
    uint64_t blocker_circles; /* navhip_blockers_circles entries                                   */
} navhip_counters;
int  navhip_get_counters(navhip_ctx *ctx, navhip_counters *out, int reset);
/* How the last agent step split its agents (waits for it): the number of agents whose ClearPath ran
 * on a row of 16 lanes ([0..3]: 1-2, 3-4, 5-8, 9-16 neighbours), on a wave ([4]: 17-64 neighbours),
 * and agents whose whole step ran on a wave ([5]: garrisoned neighbours / wide queries).  Everyone
 * else has no ClearPath neighbour and finished in the thread-per-agent pass. */
int  navhip_last_step_lists(navhip_ctx *ctx, int32_t out_counts[6]);
/* The same without waiting: the counts of the most recent step whose copy has ARRIVED (the library sends them to
 * pinned host memory behind every step, on its side stream) -- one or two steps old, zeros before the first.
 * For a host that adapts its schedule to the crowd: tick.py starts the next tick's field builds with the tick
 * instead of behind the neighbour walk once the workgroup searches ([4]) would starve them of registers. */
int  navhip_step_lists_peek(navhip_ctx *ctx, int32_t out_counts[6]);

/* ClearPath retry statistics of this process since the last reset (diagnostics: bench.py's status
 * histogram): out[k], k = 1..7 = searches that returned in attempt k + 1 of G_ClearPath_NewVelocity's
 * remove_furthest loop (clearpath.c:704-713; [7]: eight or more), out[0] = searches that gave up with a
 * list empty, out[8] = attempts of all retried searches.  Waits for the device. */
int  navhip_debug_cp_attempts(unsigned long long out[9], int reset);

/* ---- more than one GPU: the tick's exchange step for a C host --------- */

/* One process per GPU, one context per process.  The path shards without any data-path collective
 * (requests by destination, uid slabs of the velocity step: movement.c:3759-3762) up to ONE exchange per
 * tick: all ranks need all rows of [new position | velocity] before the next tick's neighbour queries.
 * These entry points call librccl directly (loaded on first use; RCCL over xGMI on an MI355X node), so
 * that the reference's C host does not need torch.distributed.
 *   rank 0 makes the id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by whatever means
 *   the host has (a file, a socket, an environment variable); every rank then calls navhip_comm_init. */
#define NAVHIP_COMM_ID_BYTES 128
int  navhip_comm_unique_id(uint8_t out_id[NAVHIP_COMM_ID_BYTES]);
int  navhip_comm_init(navhip_ctx *ctx, int rank, int world, const uint8_t id[NAVHIP_COMM_ID_BYTES]);
/* The same exchange without RCCL, for bringing the step up on one device or behind a transport of the host's
 * own: the "network" is dev_mailbox, a device buffer laid out like the gathered array ([n][4] floats for
 * navhip_comm_allgather_step_dev, the rows themselves for _rows_dev).  A call deposits this rank's rows there and
 * takes every other rank's rows from there -- the caller has put them in (or moves the mailbox between ranks by
 * whatever means it has).  Packing, slab bounds, the ragged grouping and unpacking are the code RCCL runs behind;
 * the GPU tests drive every rank of a 2-, 3- and 4-rank job through it on one device. */
int  navhip_comm_init_mailbox(navhip_ctx *ctx, int rank, int world, void *dev_mailbox, size_t mailbox_bytes);
void navhip_comm_destroy(navhip_ctx *ctx);
int  navhip_comm_rank(const navhip_ctx *ctx);     /* -1 without a communicator */
int  navhip_comm_world(const navhip_ctx *ctx);    /*  0 without a communicator */
/* The slab results of navhip_agent_step_dev -> every rank, in place, asynchronous on `stream`:
 * bounds[r] .. bounds[r + 1] = the uid slab rank r stepped (bounds[0] = 0, bounds[world] = n_ents).  ONE
 * collective: the rank's rows are packed into [n][4] floats (16 B per agent), all-gathered
 * (ncclAllGather for equal slabs, a group of ncclBroadcast for ragged ones), and the other ranks' rows
 * unpacked into dev_new_pos_xz / dev_vel_xz. */
int  navhip_comm_allgather_step_dev(navhip_ctx *ctx, float *dev_new_pos_xz, float *dev_vel_xz,
                                    const int32_t *bounds, void *stream);
/* The same for any row array (baked 4 KB flow tiles: row_bytes = 4096, bounds over the request stream). */
int  navhip_comm_allgather_rows_dev(navhip_ctx *ctx, void *dev_rows, size_t row_bytes, const int32_t *bounds,
                                    void *stream);

/* ---- the whole tick behind ONE call (host code stays in C: the per-tick orchestration is the library's) ------- */

/* The reference runs its navigation tick from one place: move_do_tick (movement.c:4312) -> navigation_tick_task
 * (:4263-4280): field work, the velocity fork-join, the state fork-join, the snapshot for the next tick.  A navhip_tick
 * is that loop for a device-resident world: every call of navhip_tick_run enqueues, per tick,
 *   - the chunk-field builds of this context's share of the requests (into the pool slot of each request; with a second
 *     pool the fields tick t+1 samples are built DURING tick t on a stream of their own, behind the narrow front of
 *     the step -- the schedule DESIGN.md section 4 describes),
 *   - [dynamic obstacles] the tick's N_BlockersIncref/Decref batch in front of them (incremental repair),
 *   - the velocity step + position accept of the uid slab (navhip_agent_prefetch_dev + navhip_agent_step_dev),
 *   - [a communicator on the context] the slab exchange (navhip_comm_allgather_step_dev) on a stream of its own, which
 *     only the snapshot consumers of the next tick wait for,
 *   - the advance of the snapshot: position / velocity buffers and the two pools ping-pong.
 * Nothing is waited for; the host returns after a few dozen launches.  Results are those of the separate calls, bit for
 * bit: the same kernels on the same buffers in the same order per stream. */
typedef struct navhip_tick navhip_tick;
#define NAVHIP_TICK_SERIAL 0x2u   /* the whole tick on ONE stream: no side streams, no events, the fields in front of
                                     the step of their own tick (field_pool_1 unused).  For small worlds, whose tick is a
                                     chain of short dependent launches: every cross-stream edge costs a barrier packet
                                     (10-20 us once the host runs ahead of the device) and buys no overlap there --
                                     configs[0] 0.27 -> ... ms per tick (profiles/r05_host_overhead_*.txt)              */
#define NAVHIP_TICK_TIME_FIELDS 0x8u /* time the field builds of every fourth tick with two HIP events on the FIELD stream (no
                                     packet on the agent stream): navhip_tick_info.fields_ms / fields_samples -- how long
                                     the builds take INSIDE the tick, beside the agent step they overlap with           */
#define NAVHIP_TICK_OWNS_SNAPSHOT 0x10u /* between two ticks the caller enqueues nothing on the tick's stream that the next
                                     tick has to follow, and writes the snapshot buffers only with the device idle: the side
                                     streams of tick t+1 then start behind a word tick t stored in device memory when it
                                     ended (NAVHIP_PREFETCH_FOLLOWS_STEP) -- inside one navhip_tick_run that is so anyway  */
typedef struct navhip_tick_desc {
    navhip_world world;             /* DEVICE arrays of the snapshot (buffer set 0: pos_xz, vel_xz, field_pool);
                                       work_begin/work_end = this rank's uid slab                                      */
    float    *pos_xz_1, *vel_xz_1;  /* [n][2] buffer set 1: tick t reads set (t & 1) and writes the other               */
    uint8_t  *status;               /* [n] NAVHIP_ST_* or NULL                                                          */
    float    *vdes_xz, *vpref_xz;   /* [n][2] or NULL (debug / parity outputs of navhip_step_out)                       */
    const navhip_field_req *dev_reqs;   /* this rank's chunk-field requests, rebuilt every tick (device)                */
    int32_t   n_reqs;
    int32_t   req_slot0;            /* request i is built into slot req_slot0 + i of the pool                           */
    uint8_t  *field_pool_1;         /* [slots][4096] second pool: the fields of tick t+1 are built during tick t into the
                                       pool tick t does not sample; NULL = built in front of the step of their own tick */
    int32_t   field_cus;            /* compute units the ahead builds may use (the last field_cus of the device); 0 = all */
    int32_t   fields_stage;         /* NAVHIP_STAGE_NEIGHBOURS / NAVHIP_STAGE_START: where in tick t they start         */
    const navhip_circle *dev_moves; /* [n_move_ticks][n_moves] the N_BlockersIncref/Decref batch of every tick, or NULL */
    int32_t   n_moves, n_move_ticks;
    int32_t   move_tick0;           /* the batch of this object's first tick (tick k applies row (move_tick0 + k) % n_move_ticks) */
    const int32_t *bounds;          /* HOST [world + 1] uid slabs of the ranks: the exchange of navhip_comm_allgather_step_dev
                                       through the context's communicator every tick; NULL = no exchange                */
    void     *stream, *field_stream, *comm_stream;   /* hipStream_t or NULL = streams of the library's own              */
    uint32_t  flags;                /* NAVHIP_TICK_*                                                                    */
} navhip_tick_desc;
typedef struct navhip_tick_info {
    int64_t  ticks;                 /* ticks enqueued so far: the current snapshot is buffer set (ticks & 1)            */
    double   host_enqueue_ms;       /* host time spent inside navhip_tick_run since creation                            */
    void    *stream, *field_stream, *comm_stream;
    double   fields_ms;             /* NAVHIP_TICK_TIME_FIELDS: mean duration of the timed field builds that have finished */
    int32_t  fields_samples, _pad;
} navhip_tick_info;
int  navhip_tick_create(navhip_ctx *ctx, const navhip_tick_desc *desc, navhip_tick **out);
/* Enqueue n ticks; asynchronous. */
int  navhip_tick_run(navhip_tick *tick, int n);
/* The two halves for a host with a transport of its own between them (tick.py's torch.distributed exchange):
 * compute = everything up to the exchange of ONE tick, advance = the ping-pong behind the host's exchange. */
int  navhip_tick_compute(navhip_tick *tick);
int  navhip_tick_advance(navhip_tick *tick);
int  navhip_tick_sync(navhip_tick *tick);
int  navhip_tick_get_info(const navhip_tick *tick, navhip_tick_info *out);
void navhip_tick_destroy(navhip_tick *tick);

/* ---- the arrival arm of the movement state machine --------------------------- */

/* entity_compute_update (movement.c:2303) decides, per unit and tick, the next movement state.  Most of it is
 * game logic over host state (orientation and the heading gate, formations, the arrival overlay, surround
 * targets, wait timers) and stays with the host.  The part every point-seeking unit runs every tick is the
 * STATE_MOVING / STATE_MOVING_IN_FORMATION case without a formation and without an active arrival group
 * (:2441-2520) -- and that part is data parallel:
 *   1. garrisoned and not still          -> STATE_ARRIVED, no blocker                       (:2344-2351)
 *   2. new position not pathable         -> no transition                                   (:2437-2438)
 *   3. arrived(uid, new_pos) (:2170): within 1.5 radii of the flock target; or adjacent to an impassable
 *      tile (N_IsAdjacentToImpassable, nav.c:4745) and maximally close to the target (N_IsMaximallyClose,
 *      nav.c:4707: within that distance of one of the destination's closest island tiles); or within it of
 *      N_ClosestPathable(target) (nav.c:4126)                 -> STATE_ARRIVED, blocker
 *   4. a flock mate within radius + radius + 5 has arrived (adjacent_flock_members :953, :2480-2497)
 *                                                             -> STATE_ARRIVED, blocker
 *   5. no guidance: |vdes| < 1/1024                           -> STATE_WAITING, blocker     (:2508-2515)
 * The two nav queries of step 3 depend on the destination only; the host makes them once per flock
 * (flock_nearest_xz, flock_tiles) and the device runs the per-unit tests.  Everything else is reported
 * back as NAVHIP_SU_HOST (the host runs entity_compute_update for that unit as before). */
#define NAVHIP_SU_SET_STATE  0x01   /* UPDATE_SET_STATE: next_state holds the new state                       */
#define NAVHIP_SU_BLOCK      0x02   /* movestate_patch.next_block                                            */
#define NAVHIP_SU_HOST       0x80   /* not decided here: another state, skip[i] != 0 (an active arrival zone:
                                       navhip_arrival_settle; a rate below 20 Hz), or a unit whose nav layer is
                                       not its flock's.  Formation members need not be skipped: see
                                       navhip_state_update_aux                                                 */
typedef struct navhip_state_in {
    const float    *new_pos_xz;       /* [n][2] the position entity_compute_update tests: new_pos_for_vel(uid,
                                                new_vel) (movement.c:2338) -- position + the velocity of the tick
                                                AFTER the heading gate (:2321-2334) -- at 20 Hz; at lower movement
                                                rates the interpolated intermediate position that replaces it
                                                (:2368-2377) -- or mark those units in `skip` (the binding does) */
    const float    *vdes_xz;          /* [n][2] move_work_out.ent_des_v                                        */
    const uint8_t  *skip;             /* [n] or NULL                                                           */
    const uint8_t  *flock_layer;      /* [F]    the nav layer flock_nearest_xz / flock_tiles were made for     */
    const float    *flock_nearest_xz; /* [F][2] N_ClosestPathable(layer, flock.target_xz); x = NaN: none       */
    const int32_t  *flock_tiles_off;  /* [F+1]  CSR offsets into flock_tiles                                   */
    const int16_t  *flock_tiles;      /* [..][2] absolute nav tiles (row, column) of n_closest_island_tiles(
                                                 dest tile, its global island, false), nav.c:4725, in order    */
} navhip_state_in;
/* world: the snapshot of the tick (pos_xz, radius, flags, state, flock, flock tables; uid slab
 * work_begin/work_end as in the step).  out_state / out_flags: [n] bytes, rows of the slab written.
 * Host buffers. */
int  navhip_state_update(navhip_ctx *ctx, const navhip_world *world, const navhip_state_in *in,
                         uint8_t *out_state, uint8_t *out_flags);
/* Everything resident on the device, asynchronous on `stream`. */
int  navhip_state_update_dev(navhip_ctx *ctx, const navhip_world *dev_world, const navhip_state_in *dev_in,
                             uint8_t *dev_out_state, uint8_t *dev_out_flags, void *stream);

/* ---- more of entity_compute_update: the heading gate, the arms of the state switch that flags, a
 *      counter, an angle or a distance decide, the arrival overlay's settle rule, and all of it but the last in one call ----
 *
 * The heading gate (movement.c:2319-2336): a unit in STATE_MOVING / SEEK_ENEMIES / SURROUND_ENTITY /
 * ENTER_ENTITY_RANGE (move_gated_by_heading, :2273) whose new velocity is longer than EPSILON does not translate
 * while its facing (movestate.next_rot) is more than MOVE_HEADING_HALT (90 degrees, a unit that is rolling:
 * |movestate.velocity| > EPSILON) or MOVE_HEADING_RESUME (10 degrees, a halted one) off its intended heading
 * (vdes when that is longer than EPSILON, else the new velocity, :2286): its velocity becomes zero and it turns in
 * place (turn_to_move).  The reference measures the angle with float quaternions and atan2
 * (PFM_Quat_PitchDiff, pf_math.c:677; dir_quat_from_velocity, movement.c:1411); the device compares the cosine
 * in double and leaves a unit whose cosine is within 1e-4 of the tolerance's to the host (NAVHIP_GATE_HOST;
 * the float path's error is below 1e-6).  world: n_ents, pos_xz, vel_xz (movestate.velocity), state, work range.
 * out_vel_xz[i] = the velocity after the gate, out_new_pos_xz[i] = new_pos_for_vel (:1820) of it -- what
 * navhip_state_in.new_pos_xz wants at 20 Hz -- out_gate[i] = NAVHIP_GATE_*.  Rows of the slab are written. */
typedef struct navhip_gate_in {
    const float *next_rot;     /* [n][4] movestate.next_rot (x, y, z, w)                            */
    const float *new_vel_xz;   /* [n][2] move_work_out.ent_vel: the velocity the step produced      */
    const float *vdes_xz;      /* [n][2] move_work_out.ent_des_v                                    */
    /* Movement rates below 20 Hz (world->hz = 10 / 5 / 1), both NULL at 20 Hz: entity_compute_update then tests -- in
     * the state switch, in G_Arrival_ShouldSettle -- not pos + vel but the first INTERPOLATED position of the move
     * when the move is accepted (movement.c:2356-2377: |vel| > 0, new position pathable, and not blocked unless the
     * unit already stands on a blocker): interpolate_positions(movestate.next_pos, new_pos, movestate.step) (:2222,
     * float arithmetic; `to` itself when |1 - step| < 1/1024).  out_new_pos_xz receives that position.  Needs
     * world->radius and ->flags (the unit's nav layer); a unit whose layer has no cost plane comes back
     * NAVHIP_GATE_HOST. */
    const float *interp_from_xz;   /* [n][2] movestate.next_pos (x, z)                              */
    const float *interp_step;      /* [n]    movestate.step                                         */
} navhip_gate_in;
#define NAVHIP_GATE_TURN  0x01   /* turn_to_move: the velocity was zeroed, the unit pivots (UPDATE_TURNING_IN_PLACE) */
#define NAVHIP_GATE_HOST  0x80   /* within the margin: not decided, out_vel / out_new_pos hold the UNGATED step     */
int  navhip_heading_gate(navhip_ctx *ctx, const navhip_world *world, const navhip_gate_in *in,
                         float *out_vel_xz, float *out_new_pos_xz, uint8_t *out_gate);
/* Everything resident on the device, asynchronous on `stream`. */
int  navhip_heading_gate_dev(navhip_ctx *ctx, const navhip_world *dev_world, const navhip_gate_in *dev_in,
                             float *dev_out_vel_xz, float *dev_out_new_pos_xz, uint8_t *dev_out_gate, void *stream);

/* The arms of the state switch that flags and a counter decide (movement.c:2423-2437, :2630-2644, :2645-2668): a
 * formation member in STATE_MOVING / MOVING_IN_FORMATION that waits for its assignment or has come within range of its
 * cell (-> ARRIVING_TO_CELL), STATE_ARRIVING_TO_CELL (-> MOVING / MOVING_IN_FORMATION / TURNING), the timer of
 * STATE_WAITING (-> movestate.wait_prev once it runs out), the end of STATE_TURNING (-> ARRIVED) and
 * STATE_ENTER_ENTITY_RANGE.  Called AFTER navhip_state_update on the same slab, with
 * formation members NOT skipped there: a member the flags do not decide keeps the answer of the arrival arm, exactly
 * as the reference falls through to it (:2439).  Rows this call decides are overwritten in inout_state / inout_flags
 * (NAVHIP_SU_HOST cleared); garrisoned units and units whose new position is not pathable (:2437) are left alone /
 * left unchanged as there.  out_wait_ticks_left[i] = movestate.wait_ticks_left after the tick (written for every row
 * of the slab).  world: n_ents, radius, flags, state, map_pos, work range (pos_xz too with the
 * enter-range inputs). */
#define NAVHIP_SU_SET_MOVING  0x04   /* UPDATE_SET_MOVING: the state is movestate.wait_prev (the wait ran out, :2641)     */
#define NAVHIP_SU_TARGET_DIR  0x08   /* UPDATE_SET_TARGET_DIR rides along: next_target_dir = fstate.target_orientation   */
#define NAVHIP_SU_SET_DEST    0x10   /* UPDATE_SET_DEST | UPDATE_SET_TARGET_PREV: next_dest = next_target_prev = the target's
                                        position, next_attack = false (:2597-2602); the state stays                      */
#define NAVHIP_SU_SURROUND_DEST 0x20 /* STATE_SURROUND_ENTITY: UPDATE_SET_DEST | UPDATE_SET_STATE, next_dest =
                                        out_surround_dest_xz[i], next_attack = false, the state stays (:2555-2560)         */
#define NAVHIP_SU_SURROUND_PREV 0x40 /* movestate.surround_target_prev = the target's position, .surround_nearest_prev =
                                        out_surround_dest_xz[i] (the reference writes them inside the switch, :2551-2552)  */
#define NAVHIP_FS_MEMBER      0x01   /* fstate.fid != NULL_FID                                                           */
#define NAVHIP_FS_READY       0x02   /* fstate.assignment_ready                                                          */
#define NAVHIP_FS_ASSIGNED    0x04   /* fstate.assigned_to_cell                                                          */
#define NAVHIP_FS_IN_RANGE    0x08   /* fstate.in_range_of_cell                                                          */
#define NAVHIP_FS_ARRIVED     0x10   /* fstate.arrived_at_cell                                                           */
typedef struct navhip_state_aux_in {
    const uint8_t  *fstate;           /* [n] NAVHIP_FS_* of move_work_in.fstate (struct formation_state, movement.c:215) */
    const int32_t  *wait_ticks_left;  /* [n] movestate.wait_ticks_left                                                   */
    const uint8_t  *wait_prev;        /* [n] movestate.wait_prev                                                         */
    const float    *new_pos_xz;       /* [n][2] as navhip_state_in.new_pos_xz                                            */
    /* STATE_TURNING (:2606-2628), both NULL = its units stay NAVHIP_SU_HOST: the unit has arrived once its rotation is
     * within 5 degrees of movestate.target_dir (cosine compared in double, a unit within 1e-5 of cos 5 deg -- 0.007
     * degrees -- stays the host's); one that keeps turning is decided too (no transition) -- its rotation patch (turn_toward, :2251) is
     * pose bookkeeping the host does for it */
    const float    *ent_rot;          /* [n][4] Entity_GetRot(uid) (x, y, z, w)                                          */
    const float    *target_dir;       /* [n][4] movestate.target_dir                                                     */
    /* STATE_ENTER_ENTITY_RANGE (:2569-2604), all six NULL = its units stay NAVHIP_SU_HOST: no target -> ARRIVED; within
     * movestate.target_range of the target, or next to an impassable tile and on one of the closest island tiles of
     * the target's position (N_IsMaximallyClose with tolerance 0, nav.c:4707) -> WAITING; else, once the target has
     * moved more than 5 units from target_prev_pos, NAVHIP_SU_SET_DEST.  The target's position is world->pos_xz of its
     * row.  Needs the BLOCKERS plane of the units' layers. */
    const int32_t  *range_target;     /* [n] row of movestate.surround_target_uid in the world's arrays; -1 = NULL_UID;
                                             -2 (any value below -1) = leave the unit to the host (a target outside
                                             the snapshot)                                                               */
    const float    *target_range;     /* [n] movestate.target_range                                                      */
    const float    *target_prev_xz;   /* [n][2] movestate.target_prev_pos                                                */
    const int32_t  *range_tiles_row;  /* [n] row of range_tiles_off for the unit: the closest island tiles of ITS target's
                                             position on ITS nav layer (n_closest_island_tiles, nav.c:4725, as flock_tiles
                                             of navhip_state_in)                                                         */
    const int32_t  *range_tiles_off;  /* [rows + 1] CSR offsets into range_tiles                                         */
    const int16_t  *range_tiles;      /* [..][2] absolute nav tiles (row, column)                                        */
    int32_t         n_range_rows;     /* rows of range_tiles_off (host-buffer call: sizes the transfer)                  */
    /* STATE_SURROUND_ENTITY (:2509-2567), surround_target NULL = its units stay NAVHIP_SU_HOST (as do all of them at a
     * rate below 20 Hz).  The two nav queries on the unit-query context stay the host's, handed over per unit exactly as
     * flock_nearest_xz is for arrived(): whether the unit already touches its target (M_NavObjAdjacentFrom, map.c:1061 --
     * or the target is gone), and the closest reachable position next to the target (M_NavClosestReachableAdjacentPosFrom,
     * map.c:860) from BOTH positions the tick can test: pos + new velocity, and pos (the heading gate zeroed the
     * velocity).  The host only needs to fill the query answers for units that reach the query (:2532-2545: the target
     * has moved since surround_target_prev, or the unit stands still).  The device runs the switch: no target / adjacent
     * / no reachable position -> ARRIVED; the position differs from the flock's target -> NAVHIP_SU_SURROUND_DEST with the
     * position in out_surround_dest_xz; no guidance -> WAITING; and reports NAVHIP_SU_SURROUND_PREV where the reference
     * stores surround_target_prev = the target's position, surround_nearest_prev = out_surround_dest_xz[i] (:2551-2552).
     * world: pos_xz, vel_xz (movestate.velocity), flock, flock_target_xz too. */
    const int32_t  *surround_target;  /* [n] row of movestate.surround_target_uid; -1 = NULL_UID; below -1 = leave to the host */
    const uint8_t  *surround_query;   /* [n] NAVHIP_SQ_*                                                                 */
    const float    *surround_target_prev_xz;   /* [n][2] movestate.surround_target_prev                                  */
    const float    *surround_nearest_prev_xz;  /* [n][2] movestate.surround_nearest_prev                                 */
    const float    *surround_dest_xz; /* [n][2][2] the query's answer from pos + new velocity ([i][0]) and from pos ([i][1]) */
    const float    *vdes_xz;          /* [n][2] move_work_out.ent_des_v (the surround arm's no-guidance test)            */
    float          *out_surround_dest_xz;      /* [n][2] written for units flagged NAVHIP_SU_SURROUND_PREV               */
    /* SPARSE ROWS (navhip_state_pass_resident only; NULL / 0 everywhere else).  The three arms above concern a few units
     * in a hundred, and their arrays are 108 bytes per ENTITY.  With sparse_units != NULL every per-unit array of the
     * three arms that is given -- ent_rot, target_dir; range_target, target_range, target_prev_xz, range_tiles_row;
     * surround_target, surround_query, surround_target_prev_xz, surround_nearest_prev_xz, surround_dest_xz,
     * out_surround_dest_xz -- holds ONE ROW PER LISTED UNIT, row k for unit sparse_units[k], instead of one per entity;
     * the device lays them out.  List every unit of the three states that the device is to decide: an unlisted
     * ENTER_ENTITY_RANGE / SURROUND_ENTITY unit is left to the host (as a target below -1), an unlisted TURNING unit is
     * computed from a zero rotation (as a dense row nobody filled would be).  A unit may be listed once. */
    const int32_t  *sparse_units;     /* [n_sparse] rows of the world's arrays, inside the slab                          */
    int32_t         n_sparse;
} navhip_state_aux_in;
#define NAVHIP_SQ_ADJACENT   0x01   /* !entity_exists(target) || M_NavObjAdjacentFrom(map, uid, target, ctx)              */
#define NAVHIP_SQ_HAS_DEST_0 0x02   /* M_NavClosestReachableAdjacentPosFrom(.., pos + new velocity, ..) found a position */
#define NAVHIP_SQ_HAS_DEST_1 0x04   /* ... from pos                                                                      */
int  navhip_state_update_aux(navhip_ctx *ctx, const navhip_world *world, const navhip_state_aux_in *in,
                             uint8_t *inout_state, uint8_t *inout_flags, int32_t *out_wait_ticks_left);
/* Everything resident on the device, asynchronous on `stream`. */
int  navhip_state_update_aux_dev(navhip_ctx *ctx, const navhip_world *dev_world, const navhip_state_aux_in *dev_in,
                                 uint8_t *dev_inout_state, uint8_t *dev_inout_flags, int32_t *dev_out_wait_ticks_left,
                                 void *stream);

/* The state half of the tick in ONE host-buffer call (fork_join_state_updates, movement.c:4196, as navhip_agent_step is
 * the velocity half): heading gate -> navhip_state_update on the gate's new positions -> navhip_state_update_aux, the
 * snapshot staged once, one wait at the end.  state.new_pos_xz, state.vdes_xz and aux.new_pos_xz are IGNORED (the gate's
 * output and gate.vdes_xz are used); aux.fstate == NULL: no aux pass.  A unit the gate leaves to the host
 * (NAVHIP_GATE_HOST) comes back NAVHIP_SU_HOST with its state and wait counter untouched: the host decides all of it.
 * Units of flocks with an active arrival zone are skipped as before (state.skip) and decided afterwards by
 * navhip_settled_count / navhip_arrival_settle on the returned positions.  Rows of the slab are written. */
typedef struct navhip_state_pass_in {
    navhip_gate_in      gate;
    navhip_state_in     state;
    navhip_state_aux_in aux;
} navhip_state_pass_in;
typedef struct navhip_state_pass_out {
    uint8_t  *state, *flags;          /* [n] next state, NAVHIP_SU_*                                                   */
    uint8_t  *gate;                   /* [n] NAVHIP_GATE_*                                                             */
    float    *new_pos_xz;             /* [n][2] new_pos_for_vel of the gated velocity                                  */
    float    *vel_xz;                 /* [n][2] the gated velocity, or NULL                                            */
    int32_t  *wait_ticks_left;        /* [n] or NULL (required with an aux pass)                                       */
} navhip_state_pass_out;
int  navhip_state_pass(navhip_ctx *ctx, const navhip_world *world, const navhip_state_pass_in *in,
                       const navhip_state_pass_out *out);

/* The same pass ON THE SNAPSHOT THE VELOCITY HALF LEFT ON THE DEVICE: between navhip_agent_step_wait (or a poll that
 * returned 0) and the next submit, the arrays of that step -- positions, velocities, radii, flags, states, flock tables --
 * and its results -- the new velocities, the desired directions -- are still in HBM.  The state half of the same tick
 * (fork_join_state_updates follows the velocity fork-join in navigation_tick_task, movement.c:4263-4280) only uploads
 * what it alone reads: movestate.next_rot, the per-flock query results, the flag / counter / target inputs (24 bytes
 * per unit in the common case instead of 95), packed through pinned memory as one transfer each way.  in->gate.new_vel_xz
 * and .vdes_xz are ignored (the step's own outputs are used: the step must have been asked for vdes_xz unless every
 * desired direction was the caller's, world->vdes_xz without NaN entries); world = that step's, slab included.
 * NAVHIP_ERR_INVALID when no completed host-buffer step is resident: call navhip_state_pass instead. */
int  navhip_state_pass_resident(navhip_ctx *ctx, const navhip_state_pass_in *in, const navhip_state_pass_out *out);

/* adjacent_settled_count (movement.c:982) for nq units of the snapshot: G_Pos_EntsInCircleFrom (r = max(30,
 * 2 radius + 5), at most 128 results, garrisoned entities dropped, position.c:379) and of those the movable
 * ones of the same air / ground kind in STATE_ARRIVED that touch the unit (distance <= both radii +
 * ADJACENCY_SEP_DIST).  out_counts[q] = -1 for a unit with radius > 12.5 (its query is wider than the 30 units
 * the spatial index is asked for here): the host counts.  world: n_ents, pos_xz, radius, flags, state, grid
 * bounds.  Host buffers. */
int  navhip_settled_count(navhip_ctx *ctx, const navhip_world *world, int nq, const int32_t *uids,
                          int32_t *out_counts);
/* The same on the snapshot the velocity half of the tick left on the device (see navhip_state_pass_resident): nothing of
 * the snapshot travels, the index is built over the resident positions, the query ids stay in HBM; only the uids go up
 * and the counts come back.  world: n_ents and the grid bounds.  NAVHIP_ERR_INVALID when no such step is resident. */
int  navhip_settled_count_resident(navhip_ctx *ctx, const navhip_world *world, int nq, const int32_t *uids,
                                   int32_t *out_counts);

/* G_Arrival_ShouldSettle (arrival.c:946) for nq units whose flock has an active arrival zone (G_Arrival_IsActive)
 * for their nav layer -- the arm of entity_compute_update at movement.c:2443-2451 that navhip_state_update leaves
 * to the host (skip[i]).  A zone is one struct arrival_state (arrival.h:66): the slots, their fill ranks and the
 * sorted tile keys of its footprint are given as ranges of three shared arrays.  Per unit: the position the
 * state update tests, the count of settled neighbours (navhip_settled_count), and its struct
 * arrival_unit_state (arrival.h:105) -- which the rule also UPDATES (arming, the progress anchor, the stuck
 * count): the out_* arrays receive the state after the call, as the reference leaves it in movestate.arrival.
 * world: n_ents, vel_xz (movestate.velocity), radius, map_pos.  Needs the BLOCKERS plane of the zones' layers
 * (M_NavPositionBlocked on a slot).  out_settle[q] = 1: UPDATE_SET_STATE, STATE_ARRIVED, next_block. */
typedef struct navhip_arrival_zone {
    float    centre_x, centre_z;      /* arrival_state.centre                                                   */
    float    unit_radius;             /* .unit_radius                                                           */
    float    fill_frac;               /* .fill_frac                                                             */
    int32_t  radius;                  /* .radius (nav tiles)                                                    */
    int32_t  layer;                   /* .layer                                                                 */
    int32_t  active_row, num_rows;    /* .active_row, .num_rows                                                 */
    int32_t  slot_begin, slot_end;    /* .slots / .slot_ring [0, num_slots) as a range of slots_xz / slot_ring  */
    int32_t  key_begin, key_end;      /* .region_keys [0, num_region) as a range of region_keys                 */
} navhip_arrival_zone;
typedef struct navhip_settle_in {
    int32_t  n_zones;
    int32_t  nq;
    const navhip_arrival_zone *zones; /* [n_zones]                                                              */
    const float    *slots_xz;         /* [..][2]                                                                */
    const int32_t  *slot_ring;        /* [..]                                                                   */
    const uint64_t *region_keys;      /* [..] td_key (nav.c:207): chunk_r << 48 | chunk_c << 32 | tile_r << 16 | tile_c,
                                              ascending inside a zone (N_TileKeysForPositions, nav.c:4303)      */
    const int32_t  *uid;              /* [nq] the unit (row of the world's arrays)                              */
    const int32_t  *zone;             /* [nq] its zone                                                          */
    const float    *new_pos_xz;       /* [nq][2] the position entity_compute_update tests (navhip_state_in)     */
    const int32_t  *nsettled;         /* [nq] adjacent_settled_count                                            */
    const uint8_t  *substate;         /* [nq] arrival_unit_state.substate (enum arrival_substate, arrival.h:55) */
    const uint8_t  *sink_valid;       /* [nq] .sink_valid                                                       */
    const float    *sink_xz;          /* [nq][2] .sink                                                          */
    const float    *order_pos_xz;     /* [nq][2] .order_pos                                                     */
    const float    *progress_anchor_xz; /* [nq][2] .progress_anchor                                             */
    const uint8_t  *progress_anchored;  /* [nq] .progress_anchored                                              */
    const int32_t  *stuck;            /* [nq] .stuck                                                            */
} navhip_settle_in;
typedef struct navhip_settle_out {
    uint8_t  *settle;                 /* [nq] the rule's answer                                                 */
    uint8_t  *substate;               /* [nq] arrival_unit_state after the call                                 */
    float    *progress_anchor_xz;     /* [nq][2]                                                                */
    uint8_t  *progress_anchored;      /* [nq]                                                                   */
    int32_t  *stuck;                  /* [nq]                                                                   */
    int32_t  *nsettled;               /* [nq] or NULL: the counts the rule used (navhip_arrival_settle_resident with
                                              in->nsettled == NULL); -1 = the host counts AND decides this unit     */
} navhip_settle_out;
int  navhip_arrival_settle(navhip_ctx *ctx, const navhip_world *world, const navhip_settle_in *in,
                           const navhip_settle_out *out);
/* The same with world->vel_xz / ->radius read from the snapshot the velocity half of the tick left on the device (see
 * navhip_state_pass_resident; world: n_ents, map_pos).  NAVHIP_ERR_INVALID when no such step is resident.
 * in->nsettled == NULL: adjacent_settled_count is taken on the device too (navhip_settled_count_resident's rule on the
 * same snapshot; world: the grid bounds as well) -- the two passes of the arm in ONE call, every per-unit array packed
 * through pinned memory as one transfer each way. */
int  navhip_arrival_settle_resident(navhip_ctx *ctx, const navhip_world *world, const navhip_settle_in *in,
                                    const navhip_settle_out *out);
/* Everything resident on the device (the zones' arrays and the per-unit arrays too), asynchronous on `stream`. */
int  navhip_arrival_settle_dev(navhip_ctx *ctx, const navhip_world *dev_world, const navhip_settle_in *dev_in,
                               const navhip_settle_out *dev_out, void *stream);

/* N_DesiredGroupArrivalVelocity (nav.c:3561) for nq points: the direction under each point in the chunk
 * field of mapping row rows[q] (region_field_slot / field_pool as in navhip_world; resident pool: pass
 * region_field_slot = field_pool = NULL), and whether that tile is a sink inside the zone's disc
 * (out_at_slot; centre_abs = zone centre in absolute nav tiles [nq][2] (row, column), radius[nq] in tiles).
 * out_dir[q]: 0..8 = enum flow_dir, 0xff = no field cached for the point's chunk (the reference returns
 * false).  Host buffers.  The arrival overlay's decisions around it (G_Arrival_DesiredVelocity,
 * arrival.c:862) stay with the host. */
int  navhip_region_lookup(navhip_ctx *ctx, int nq, const float *pos_xz, const int32_t *rows,
                          const int32_t *region_field_slot, int n_region_rows, const uint8_t *field_pool,
                          int n_field_slots, const int32_t *centre_abs, const int32_t *radius,
                          float map_pos_x, float map_pos_z, uint8_t *out_dir, uint8_t *out_at_slot);

/* Device spatial index only (bg_ent insert-all + cleanup + inrange_circle, bitmap_grid.h:1376):
 * for each query point the ids within `range`, in the reference's visiting order, capped at
 * maxout.  Host buffers.  Used by the parity tests of the neighbour gather. */
int  navhip_spatial_query(navhip_ctx *ctx, const navhip_world *world, const float *query_xz,
                          int nq, float range, int maxout, int32_t *out_counts, uint32_t *out_ids);

/* G_ClearPath_NewVelocity (clearpath.c:694) for nq independent problems, host buffers.
 * ent: [nq][5] {pos.x,pos.z,vel.x,vel.z,radius}; des_v: [nq][2]; dyn/stat: [nq][32][5] with
 * n_dyn/n_stat counts; out: [nq][2]. */
int  navhip_clearpath(navhip_ctx *ctx, int nq, const float *ent, const float *des_v,
                      const float *dyn, const int32_t *n_dyn, const float *stat,
                      const int32_t *n_stat, float *out);
/* The same problems on one row of 16 lanes each -- the form the agent step uses for agents with at most
 * 16 ClearPath neighbours (n_dyn + n_stat <= 16; navhip_clearpath runs one wave per problem, the
 * form for agents in a crowd). */
int  navhip_clearpath_rows(navhip_ctx *ctx, int nq, const float *ent, const float *des_v,
                           const float *dyn, const int32_t *n_dyn, const float *stat,
                           const int32_t *n_stat, float *out);
/* The same problems, each searched by the waves of one workgroup as a team -- the form the agent step
 * uses for agents with 17..64 ClearPath neighbours. */
int  navhip_clearpath_team(navhip_ctx *ctx, int nq, const float *ent, const float *des_v,
                           const float *dyn, const int32_t *n_dyn, const float *stat,
                           const int32_t *n_stat, float *out);

#ifdef __cplusplus
}
#endif
#endif /* NAVHIP_H */
