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

/* field_target.type values, field.h:85-98 (only the chunk-aligned targets) */
enum { NAVHIP_TARGET_PORTAL = 0, NAVHIP_TARGET_TILE = 1 };

/* per-layer planes held on the device (struct nav_chunk members, nav_data.h:118-158) */
enum {
    NAVHIP_PLANE_COST_BASE     = 0,   /* uint8_t  [64][64] per chunk              */
    NAVHIP_PLANE_BLOCKERS      = 1,   /* uint16_t [64][64] per chunk              */
    NAVHIP_PLANE_LOCAL_ISLANDS = 2,   /* uint16_t [64][64] per chunk              */
    NAVHIP_PLANE_FACTIONS      = 3,   /* uint8_t  [15][64][64] per chunk          */
    NAVHIP_PLANE_COUNT
};

/* navhip_field_req.flags */
#define NAVHIP_REQ_INOUT   0x1   /* update an existing field in place: unreachable cells keep
                                    the bytes already in the output slot (field.c:737-751,
                                    nav.c:1998-2008) instead of starting from N_FlowFieldInit */

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
    uint8_t  tile_r, tile_c;                       /* TARGET_TILE: target.tile        */
    uint8_t  port_r0, port_c0, port_r1, port_c1;   /* TARGET_PORTAL: pd.port->endpoints */
    uint8_t  next_r0, next_c0, next_r1, next_c1;   /*                pd.next->endpoints */
    uint16_t next_chunk_r, next_chunk_c;           /*                pd.next->chunk     */
    uint16_t port_iid, next_iid;                   /*                pd.port_iid/next_iid */
    uint16_t _pad[2];
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
int  navhip_sync(navhip_ctx *ctx);

/* Upload one whole plane of one layer (all chunks).  Replaces N_CopyCostBasePacked /
 * N_CopyBlockersPacked consumers (nav.c:2408-2490): same layout, same sizes. */
int  navhip_upload_plane(navhip_ctx *ctx, int layer, int plane, const void *host, size_t bytes);
/* Upload one chunk of one plane (dirty-chunk update after N_Update, nav.c:2119). */
int  navhip_upload_chunk(navhip_ctx *ctx, int layer, int plane, int chunk_r, int chunk_c,
                         const void *host, size_t bytes);
/* Device pointer of a resident plane (NULL when never uploaded); for zero-copy producers. */
void *navhip_plane_dev(navhip_ctx *ctx, int layer, int plane);

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

/* N_FlowFieldID (field.c:1952) for TILE / PORTAL targets: the 64-bit cache key the reference's
 * fieldcache uses; pure bit packing, host side. */
uint64_t navhip_flow_field_id(const navhip_field_req *req);

/* kernel selection override for tests/bench: 0 = auto (bit-parallel BFS when every passable
 * cell of the chunk has cost 1, generic relaxation otherwise), 1 = force generic. */
int  navhip_set_field_kernel(navhip_ctx *ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif /* NAVHIP_H */
