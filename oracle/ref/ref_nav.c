/* oracle/ref/ref_nav.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Pulls the reference's src/navigation/nav.c into this translation unit (by
 * #include from /root/reference/src; nothing is copied) so the harness can
 * reach its static planner and map-preprocessing functions:
 *   n_update_portals (nav.c:1710), n_update_island_field (:1731),
 *   n_update_local_island_field (:986), n_update_dirty_local_islands (:996),
 *   n_update_blockers (:1017), n_request_path (:1774).
 * Every call nav.c makes to N_FlowFieldUpdate is routed through a recorder
 * so that the planner's chunk-field request stream can be replayed against
 * the HIP path.
 */
#define N_FlowFieldUpdate pfref_traced_N_FlowFieldUpdate
#define N_FlowFieldUpdateToNearestPathable pfref_hook_NearestPathable
#define N_FlowFieldUpdateIslandToNearest pfref_hook_IslandToNearest
/* the two other call-site changes of INTEGRATION.md, made by renaming instead of editing nav.c:
 * N_LOSFieldCreate (nav.c:1843,2035) and the Sched_Create(field_task) of the three N_RequestAsync*Field
 * functions (nav.c:3824,3878,3907) */
#define N_LOSFieldCreate pfref_hook_N_LOSFieldCreate
#define Sched_Create pfref_hook_Sched_Create
/* ... and the field cache's writers (nav.c:1833-1835,2008-2021,3533,3547,3632,3715,3964), mirrored into
 * the device-resident pool when the binding has one */
#define N_FC_PutFlowField pfref_hook_FC_PutFlowField
#define N_FC_PutDestFFMapping pfref_hook_FC_PutDestFFMapping
#include "navigation/nav.c"
#undef N_FlowFieldUpdate
#undef N_FlowFieldUpdateToNearestPathable
#undef N_FlowFieldUpdateIslandToNearest
#undef N_LOSFieldCreate
#undef Sched_Create
#undef N_FC_PutFlowField
#undef N_FC_PutDestFFMapping

#include "pfref.h"
#include "ref_internal.h"

/* the real builders (field.c) */
void N_FlowFieldUpdate(struct coord chunk_coord, const struct nav_private *priv, int faction_id,
                       enum nav_layer layer, struct field_target target,
                       struct nav_unit_query_ctx *ctx, struct flow_field *inout_flow);
void N_FlowFieldUpdateIslandToNearest(uint16_t local_iid, const struct nav_private *priv,
                                      enum nav_layer layer, int faction_id,
                                      struct nav_unit_query_ctx *ctx, struct flow_field *inout_flow);
void N_FlowFieldUpdateToNearestPathable(const struct nav_private *priv, enum nav_layer layer,
                                        struct coord chunk, struct coord start, int faction_id,
                                        struct nav_unit_query_ctx *ctx, struct flow_field *inout_flow);

void N_LOSFieldCreate(dest_id_t id, struct coord chunk_coord, struct tile_desc target,
                      const struct nav_private *priv, vec3_t map_pos, struct nav_unit_query_ctx *ctx,
                      struct LOS_field *out_los, const struct LOS_field *prev_los);
uint32_t Sched_Create(int prio, task_func_t code, void *arg, const char *name, struct future *result, int flags);
void N_FC_PutFlowField(struct fieldcache_ctx *ctx, ff_id_t ffid, const struct flow_field *ff);
void N_FC_PutDestFFMapping(struct fieldcache_ctx *ctx, dest_id_t dest_id, struct coord chunk_coord, ff_id_t ffid);

/* The reference-side binding of libnavhip.so (what a maintainer appends to nav.c): every field build
 * nav.c asks for goes through it while s_use_binding is set. */
#include "nav_hip.c"
static bool s_use_binding;

/* every N_LOSFieldCreate call of the planner: which chunk, built from which previous field (the chain of
 * nav.c:1843 / :2026-2039) -- the request stream of the LOS fields, like the flow-field trace below */
#define LOS_TRACE_MAX (1 << 20)
static int32_t (*s_los_trace)[6];          /* dest id, chunk r, c, has_prev, prev r, c */
static int s_los_trace_n;
int  pfref_los_trace_count(void) { return s_los_trace_n; }
void pfref_los_trace_clear(void) { s_los_trace_n = 0; }
void pfref_los_trace_get(int idx, int32_t out[6]) { memcpy(out, s_los_trace[idx], sizeof(int32_t) * 6); }

void pfref_hook_N_LOSFieldCreate(dest_id_t id, struct coord chunk_coord, struct tile_desc target,
                                 const struct nav_private *priv, vec3_t map_pos, struct nav_unit_query_ctx *ctx,
                                 struct LOS_field *out_los, const struct LOS_field *prev_los)
{
    if(!s_los_trace) s_los_trace = malloc(sizeof(*s_los_trace) * LOS_TRACE_MAX);
    if(s_los_trace && s_los_trace_n < LOS_TRACE_MAX) {
        int32_t *r = s_los_trace[s_los_trace_n++];
        r[0] = (int32_t)id; r[1] = chunk_coord.r; r[2] = chunk_coord.c;
        r[3] = prev_los != NULL; r[4] = prev_los ? prev_los->chunk.r : 0; r[5] = prev_los ? prev_los->chunk.c : 0;
    }
    if(s_use_binding) N_HIP_LOSFieldCreate(id, chunk_coord, target, priv, map_pos, ctx, out_los, prev_los);
    else              N_LOSFieldCreate(id, chunk_coord, target, priv, map_pos, ctx, out_los, prev_los);
}

void pfref_hook_FC_PutFlowField(struct fieldcache_ctx *ctx, ff_id_t ffid, const struct flow_field *ff)
{
    if(s_use_binding) N_HIP_FC_PutFlowField(ctx, ffid, ff);
    else              N_FC_PutFlowField(ctx, ffid, ff);
}

void pfref_hook_FC_PutDestFFMapping(struct fieldcache_ctx *ctx, dest_id_t dest_id, struct coord chunk, ff_id_t ffid)
{
    if(s_use_binding) N_HIP_FC_PutDestFFMapping(ctx, dest_id, chunk, ffid);
    else              N_FC_PutDestFFMapping(ctx, dest_id, chunk, ffid);
}

uint32_t pfref_hook_Sched_Create(int prio, task_func_t code, void *arg, const char *name,
                                 struct future *result, int flags)
{
    if(s_use_binding) return N_HIP_FieldTaskCreate(prio, code, arg, name, result, flags);
    return Sched_Create(prio, code, arg, name, result, flags);
}

void pfref_hook_NearestPathable(const struct nav_private *priv, enum nav_layer layer, struct coord chunk,
                                struct coord start, int faction_id, struct nav_unit_query_ctx *ctx,
                                struct flow_field *inout_flow)
{
    if(s_use_binding) N_HIP_FlowFieldUpdateToNearestPathable(priv, layer, chunk, start, faction_id, ctx, inout_flow);
    else              N_FlowFieldUpdateToNearestPathable(priv, layer, chunk, start, faction_id, ctx, inout_flow);
}

void pfref_hook_IslandToNearest(uint16_t local_iid, const struct nav_private *priv, enum nav_layer layer,
                                int faction_id, struct nav_unit_query_ctx *ctx, struct flow_field *inout_flow)
{
    if(s_use_binding) N_HIP_FlowFieldUpdateIslandToNearest(local_iid, priv, layer, faction_id, ctx, inout_flow);
    else              N_FlowFieldUpdateIslandToNearest(local_iid, priv, layer, faction_id, ctx, inout_flow);
}

/* ------------------------------------------------------------------------ */
/* planner trace                                                            */
/* ------------------------------------------------------------------------ */

struct trace_rec{
    pfref_field_req req;
    uint8_t         before[FIELD_RES_R * FIELD_RES_C];
    uint8_t         after[FIELD_RES_R * FIELD_RES_C];
};

static struct trace_rec *s_trace;
static int               s_trace_n, s_trace_cap;
static bool              s_trace_on;

void pfref_traced_N_FlowFieldUpdate(struct coord chunk_coord, const struct nav_private *priv,
                                    int faction_id, enum nav_layer layer,
                                    struct field_target target,
                                    struct nav_unit_query_ctx *ctx,
                                    struct flow_field *inout_flow)
{
    bool rec = s_trace_on && (target.type == TARGET_TILE || target.type == TARGET_PORTAL);
    struct trace_rec *tr = NULL;
    if(rec) {
        if(s_trace_n == s_trace_cap) {
            s_trace_cap = s_trace_cap ? s_trace_cap * 2 : 256;
            s_trace = realloc(s_trace, sizeof(struct trace_rec) * s_trace_cap);
        }
        tr = &s_trace[s_trace_n++];
        pfref_req_from_target(chunk_coord, faction_id, layer, &target, &tr->req);
        pfref_ff_to_dirs(inout_flow, tr->before);
        bool any = false;
        for(int i = 0; i < FIELD_RES_R * FIELD_RES_C; i++)
            any |= (tr->before[i] != FD_NONE);
        tr->req.inout = any;
    }
    if(s_use_binding) N_HIP_FlowFieldUpdate(chunk_coord, priv, faction_id, layer, target, ctx, inout_flow);
    else              N_FlowFieldUpdate(chunk_coord, priv, faction_id, layer, target, ctx, inout_flow);
    if(rec) {
        /* realloc may have moved the buffer only before this call; tr is still valid */
        pfref_ff_to_dirs(inout_flow, tr->after);
    }
}

int  pfref_trace_count(void) { return s_trace_n; }
void pfref_trace_clear(void) { s_trace_n = 0; }
void pfref_trace_get(int idx, pfref_field_req *out_req, uint8_t *out_before, uint8_t *out_after)
{
    *out_req = s_trace[idx].req;
    if(out_before) memcpy(out_before, s_trace[idx].before, sizeof(s_trace[idx].before));
    if(out_after)  memcpy(out_after,  s_trace[idx].after,  sizeof(s_trace[idx].after));
}

/* ------------------------------------------------------------------------ */
/* context                                                                  */
/* ------------------------------------------------------------------------ */

static bool s_nav_inited;

pfref_nav *pfref_nav_create(int w, int h, const uint8_t *cost_base, unsigned layer_mask)
{
    if(!s_nav_inited) {
        if(!N_Init())
            return NULL;
        s_nav_inited = true;
    }
    pfref_nav *nav = calloc(1, sizeof(*nav));
    if(!nav || !N_InitCtx(&nav->priv))
        return NULL;
    N_FC_ClearAll(nav->priv.fieldcache);

    nav->priv.width = w;
    nav->priv.height = h;
    nav->layer_mask = layer_mask;
    /* the engine centres the map on the origin (X grows to the left, tile.c:565) */
    nav->map_pos = (vec3_t){
        (w * TILES_PER_CHUNK_WIDTH  * X_COORDS_PER_TILE) / 2.0f, 0.0f,
       -(h * TILES_PER_CHUNK_HEIGHT * Z_COORDS_PER_TILE) / 2.0f};

    size_t nchunks = (size_t)w * h;
    for(int layer = 0; layer < NAV_LAYER_MAX; layer++) {
        if(!(layer_mask & (1u << layer)))
            continue;
        struct nav_chunk *chunks = calloc(nchunks, sizeof(struct nav_chunk));
        if(!chunks)
            return NULL;
        nav->priv.chunks[layer] = chunks;
        for(size_t i = 0; i < nchunks; i++) {
            memcpy(chunks[i].cost_base, cost_base + i * FIELD_RES_R * FIELD_RES_C,
                   FIELD_RES_R * FIELD_RES_C);
        }
        /* nav.c:2340-2343 minus the tile-derived cliff edges */
        n_update_portals(&nav->priv, layer);
        n_update_island_field(&nav->priv, layer);
        n_update_local_island_field(&nav->priv, layer);
    }
    return nav;
}

void pfref_nav_destroy(pfref_nav *nav)
{
    if(!nav) return;
    N_FC_ClearAll(nav->priv.fieldcache);
    for(int layer = 0; layer < NAV_LAYER_MAX; layer++)
        free(nav->priv.chunks[layer]);
    N_DestroyCtx(&nav->priv);
    free(nav);
}

void *pfref_nav_private(pfref_nav *nav)     { return &nav->priv; }
int   pfref_nav_width(const pfref_nav *nav) { return (int)nav->priv.width; }
int   pfref_nav_height(const pfref_nav *nav){ return (int)nav->priv.height; }
void  pfref_map_pos(const pfref_nav *nav, float out[3])
{
    out[0] = nav->map_pos.x; out[1] = nav->map_pos.y; out[2] = nav->map_pos.z;
}

void pfref_nav_set_blockers(pfref_nav *nav, int layer, const uint16_t *blockers)
{
    N_HIP_BlockersTouched();                     /* (a writer of the planes beside N_BlockersIncref / Decref) */
    struct nav_private *priv = &nav->priv;
    size_t nchunks = priv->width * priv->height;
    for(size_t i = 0; i < nchunks; i++) {
        memcpy(priv->chunks[layer][i].blockers, blockers + i * FIELD_RES_R * FIELD_RES_C,
               sizeof(priv->chunks[layer][i].blockers));
    }
    n_update_local_island_field(priv, layer);
    n_update_all_edge_states(priv, layer);
    N_FC_ClearAll(priv->fieldcache);
}

void pfref_nav_blockers_circle(pfref_nav *nav, float x, float z, float range, int faction_id,
                               uint32_t flags, int incref)
{
    struct nav_private *priv = &nav->priv;
    vec2_t xz = (vec2_t){x, z};
    const unsigned ground = 0xfu, water = 0xf0u;
    /* (the statement INTEGRATION.md adds to N_BlockersIncref / N_BlockersDecref themselves) */
    if(s_use_binding)
        N_HIP_BlockersRecord(xz, range, faction_id, flags, nav->map_pos, incref ? +1 : -1);
    else
        N_HIP_BlockersTouched();                 /* (the recorder's first statement: the planes are about to change) */
    if(!(flags & ENTITY_FLAG_AIR)
    && (nav->layer_mask & (ground | water)) == (ground | water)) {
        if(incref) N_BlockersIncref(xz, range, faction_id, flags, nav->map_pos, priv);
        else       N_BlockersDecref(xz, range, faction_id, flags, nav->map_pos, priv);
        return;
    }
    /* only GROUND_1X1 is allocated: first statement pair of
     * n_update_blockers_circle_ground (nav.c:1051-1055) */
    struct tile_desc tds[1024];
    int ntds = M_Tile_AllUnderCircle(n_res(priv), xz, range, nav->map_pos, tds, ARR_SIZE(tds));
    n_update_blockers(priv, NAV_LAYER_GROUND_1X1, faction_id, tds, ntds, incref ? +1 : -1);
}

void pfref_nav_flush_dirty(pfref_nav *nav)
{
    struct nav_private *priv = &nav->priv;
    for(int layer = 0; layer < NAV_LAYER_MAX; layer++) {
        if(!(nav->layer_mask & (1u << layer)))
            continue;
        n_update_dirty_local_islands(priv, layer);
        n_update_all_edge_states(priv, layer);
        kh_clear(coord, priv->dirty_chunks[layer]);
    }
    N_FC_ClearAll(priv->fieldcache);
}

size_t pfref_nav_copy_plane(const pfref_nav *nav, int layer, int plane, void *out)
{
    const struct nav_private *priv = &nav->priv;
    size_t nchunks = priv->width * priv->height;
    const size_t cells = FIELD_RES_R * FIELD_RES_C;
    size_t elem = (plane == 0) ? 1 : (plane == 4) ? MAX_FACTIONS : 2;
    if(!out)
        return nchunks * cells * elem;
    for(size_t i = 0; i < nchunks; i++) {
        const struct nav_chunk *ch = &priv->chunks[layer][i];
        char *dst = (char*)out + i * cells * elem;
        switch(plane) {
        case 0: memcpy(dst, ch->cost_base,     cells);      break;
        case 1: memcpy(dst, ch->blockers,      cells * 2);  break;
        case 2: memcpy(dst, ch->islands,       cells * 2);  break;
        case 3: memcpy(dst, ch->local_islands, cells * 2);  break;
        case 4: memcpy(dst, ch->factions,      cells * MAX_FACTIONS); break;
        default: return 0;
        }
    }
    return nchunks * cells * elem;
}

int pfref_nav_num_portals(const pfref_nav *nav, int layer, int chunk_r, int chunk_c)
{
    const struct nav_private *priv = &nav->priv;
    return (int)priv->chunks[layer][IDX(chunk_r, priv->width, chunk_c)].num_portals;
}

void pfref_nav_get_portal(const pfref_nav *nav, int layer, int chunk_r, int chunk_c, int idx,
                          pfref_portal *out)
{
    const struct nav_private *priv = &nav->priv;
    const struct portal *p = &priv->chunks[layer][IDX(chunk_r, priv->width, chunk_c)].portals[idx];
    const struct portal *c = n_portal(priv, layer, p->connected);
    *out = (pfref_portal){
        p->chunk.r, p->chunk.c,
        p->endpoints[0].r, p->endpoints[0].c, p->endpoints[1].r, p->endpoints[1].c,
        c->chunk.r, c->chunk.c,
        c->endpoints[0].r, c->endpoints[0].c, c->endpoints[1].r, c->endpoints[1].c,
        p->component_id
    };
}

/* ------------------------------------------------------------------------ */
/* planner / sampling                                                       */
/* ------------------------------------------------------------------------ */

int pfref_request_path(pfref_nav *nav, int layer, int faction_id,
                       float src_x, float src_z, float dst_x, float dst_z,
                       int clear_cache, uint32_t *out_dest_id)
{
    struct nav_private *priv = &nav->priv;
    if(clear_cache)
        N_FC_ClearAll(priv->fieldcache);
    dest_id_t id = 0;
    s_trace_on = true;
    bool ok = n_request_path(priv, (vec2_t){src_x, src_z}, (vec2_t){dst_x, dst_z}, faction_id,
                             nav->map_pos, layer, &id);
    s_trace_on = false;
    if(out_dest_id)
        *out_dest_id = id;
    return ok;
}

void pfref_desired_point_seek_velocity(pfref_nav *nav, uint32_t dest_id, float x, float z,
                                       float dst_x, float dst_z, float out[2])
{
    s_trace_on = true;
    vec2_t v = N_DesiredPointSeekVelocity(dest_id, (vec2_t){x, z}, (vec2_t){dst_x, dst_z},
                                          &nav->priv, nav->map_pos);
    s_trace_on = false;
    out[0] = v.x;
    out[1] = v.z;
}

int pfref_cached_field(pfref_nav *nav, uint32_t dest_id, int chunk_r, int chunk_c,
                       uint8_t *out_dirs)
{
    struct nav_private *priv = &nav->priv;
    ff_id_t ffid;
    if(!N_FC_GetDestFFMapping(priv->fieldcache, dest_id, (struct coord){chunk_r, chunk_c}, &ffid))
        return 0;
    const struct flow_field *ff = N_FC_FlowFieldAt(priv->fieldcache, ffid);
    if(!ff)
        return 0;
    pfref_ff_to_dirs(ff, out_dirs);
    return 1;
}

/* N_FC_GetDestFFMapping: the id of the field the cache maps for (dest, chunk); 0 = none */
uint64_t pfref_cached_ffid(pfref_nav *nav, uint32_t dest_id, int chunk_r, int chunk_c)
{
    ff_id_t ffid = 0;
    if(!N_FC_GetDestFFMapping(nav->priv.fieldcache, dest_id, (struct coord){chunk_r, chunk_c}, &ffid))
        return 0;
    return ffid;
}

/* the LOS field the field cache holds for (dest, chunk): bit 0 visible, bit 1 wavefront_blocked */
int pfref_cached_los(pfref_nav *nav, uint32_t dest_id, int chunk_r, int chunk_c, uint8_t *out)
{
    struct nav_private *priv = &nav->priv;
    struct coord chunk = (struct coord){chunk_r, chunk_c};
    if(!N_FC_ContainsLOSField(priv->fieldcache, dest_id, chunk))
        return 0;
    const struct LOS_field *lf = N_FC_LOSFieldAt(priv->fieldcache, dest_id, chunk);
    if(!lf)
        return 0;
    for(int r = 0; r < FIELD_RES_R; r++)
    for(int c = 0; c < FIELD_RES_C; c++)
        out[r * FIELD_RES_C + c] = (uint8_t)(lf->field[r][c].visible | (lf->field[r][c].wavefront_blocked << 1));
    return 1;
}

int pfref_has_dest_los(pfref_nav *nav, uint32_t dest_id, float x, float z, float dst_x, float dst_z)
{
    return N_HasDestLOS(dest_id, (vec2_t){x, z}, &nav->priv, nav->map_pos, (vec2_t){dst_x, dst_z});
}

int pfref_position_pathable(pfref_nav *nav, int layer, float x, float z)
{
    return N_PositionPathable((vec2_t){x, z}, layer, &nav->priv, nav->map_pos);
}

int pfref_position_blocked(pfref_nav *nav, int layer, float x, float z)
{
    return N_PositionBlocked((vec2_t){x, z}, layer, &nav->priv, nav->map_pos);
}


/* ------------------------------------------------------------------------ */
/* the binding (nav_hip.c) driven by the reference's own planner / sampler   */
/* ------------------------------------------------------------------------ */

int pfref_hip_init(pfref_nav *nav)
{
    return N_HIP_Init(&nav->priv) ? 1 : 0;
}

void pfref_hip_shutdown(void)
{
    s_use_binding = false;
    N_HIP_PoolDisable();
    N_HIP_Shutdown();
}

/* use_binding: nav.c's field builds go through N_HIP_*; backend 0 = the reference's CPU builders
 * behind the binding (control), 1 = libnavhip */
void pfref_hip_mode(int use_binding, int backend)
{
    s_use_binding = use_binding != 0;
    N_HIP_SetBackend(backend);
}

int pfref_hip_sync_layer(pfref_nav *nav, int layer)
{
    return N_HIP_SyncLayer(&nav->priv, layer) ? 1 : 0;
}

void pfref_hip_stats(long out[3]) { N_HIP_Stats(out); }

/* N_DesiredPointSeekVelocity for n agents: batched = 0 the reference's serial calls, in order;
 * batched = 1 N_HIP_DesiredPointSeekVelocities (collect -> batch build -> sample) */
void pfref_desired_velocities(pfref_nav *nav, int n, const uint32_t *dest_ids, const float *pos_xz,
                              const float *dest_xz, int batched, float *out_xz)
{
    if(!batched) {
        for(int i = 0; i < n; i++) {
            vec2_t v = N_DesiredPointSeekVelocity(dest_ids[i], (vec2_t){pos_xz[2 * i], pos_xz[2 * i + 1]},
                (vec2_t){dest_xz[2 * i], dest_xz[2 * i + 1]}, &nav->priv, nav->map_pos);
            out_xz[2 * i] = v.x; out_xz[2 * i + 1] = v.z;
        }
        return;
    }
    vec2_t *pos = malloc(sizeof(vec2_t) * (n > 0 ? n : 1)), *dst = malloc(sizeof(vec2_t) * (n > 0 ? n : 1));
    vec2_t *out = malloc(sizeof(vec2_t) * (n > 0 ? n : 1));
    for(int i = 0; i < n; i++) {
        pos[i] = (vec2_t){pos_xz[2 * i], pos_xz[2 * i + 1]};
        dst[i] = (vec2_t){dest_xz[2 * i], dest_xz[2 * i + 1]};
    }
    N_HIP_DesiredPointSeekVelocities(&nav->priv, nav->map_pos, n, dest_ids, pos, dst, out);
    for(int i = 0; i < n; i++) { out_xz[2 * i] = out[i].x; out_xz[2 * i + 1] = out[i].z; }
    free(pos); free(dst); free(out);
}

/* n_dest_id (nav.c:848) of a destination position */
uint32_t pfref_dest_id(pfref_nav *nav, int layer, int faction_id, float dst_x, float dst_z)
{
    struct map_resolution res;
    N_GetResolution(&nav->priv, &res);
    struct tile_desc td;
    if(!M_Tile_DescForPoint2D(res, nav->map_pos, (vec2_t){dst_x, dst_z}, &td))
        return 0;
    return n_dest_id(td, layer, faction_id);
}

void pfref_cache_clear(pfref_nav *nav) { N_HIP_FC_ClearAll(nav->priv.fieldcache); }

/* n fields into the reference's own field cache, as n_request_path leaves them (nav.c:1833-1835, :2008-2021):
 * N_FC_PutFlowField under N_FlowFieldID of the request + N_FC_PutDestFFMapping(dest id, chunk).  For tests that
 * sample (N_DesiredPointSeekVelocity, cache-hit path) fields the planner would take minutes to request one by one.
 * Returns the number of fields put (a request whose target cannot be rebuilt from the record is skipped). */
int pfref_cache_put_fields(pfref_nav *nav, int n, const pfref_field_req *reqs, const uint32_t *dest_ids,
                           const uint8_t *dirs)
{
    struct nav_private *priv = &nav->priv;
    int put = 0;
    for(int i = 0; i < n; i++) {
        struct field_target target;
        if(!pfref_make_target(priv, &reqs[i], &target))
            continue;
        struct coord chunk = (struct coord){reqs[i].chunk_r, reqs[i].chunk_c};
        struct flow_field ff;
        N_FlowFieldInit(chunk, &ff);
        ff.target = target;
        pfref_dirs_to_ff(dirs + (size_t)i * FIELD_RES_R * FIELD_RES_C, &ff);
        ff_id_t id = N_FlowFieldID(chunk, target, (enum nav_layer)reqs[i].layer);
        N_FC_PutFlowField(priv->fieldcache, id, &ff);
        N_FC_PutDestFFMapping(priv->fieldcache, dest_ids[i], chunk, id);
        put++;
    }
    return put;
}
int  pfref_hip_pool_enable(int n_slots, int n_rows) { return N_HIP_PoolEnable(n_slots, n_rows) ? 1 : 0; }
void pfref_hip_pool_disable(void) { N_HIP_PoolDisable(); }
void pfref_hip_pool_stats(long out[3]) { N_HIP_PoolStats(out); }


/* ------------------------------------------------------------------------ */
/* the asynchronous field batch, LOS and blocker seams of the binding        */
/* ------------------------------------------------------------------------ */

/* One movement tick's compute_async_fields (movement.c:4149-4164): N_PrepareAsyncWork, the requests
 * (kind 0 = N_RequestAsyncEnemySeekField, 1 = N_RequestAsyncSurroundField, 2 =
 * N_RequestAsyncGroupArrivalField), [the binding's one device call], N_AwaitAsyncFields.  The ids of the
 * jobs the batch accepted are returned (<= max_ids); the fields are in the reference's field cache. */
int pfref_async_batch(pfref_nav *nav, int n, const pfref_async_req *reqs, uint64_t *out_ids, int max_ids)
{
    struct nav_private *priv = &nav->priv;
    N_PrepareAsyncWork();
    for(int i = 0; i < n; i++) {
        const pfref_async_req *r = &reqs[i];
        vec2_t xz = (vec2_t){r->x, r->z};
        switch(r->kind) {
        case 0: N_RequestAsyncEnemySeekField(xz, priv, r->layer, nav->map_pos, r->faction_id); break;
        case 1: N_RequestAsyncSurroundField(xz, priv, r->layer, nav->map_pos, r->ent, r->faction_id); break;
        case 2: N_RequestAsyncGroupArrivalField(xz, priv, r->layer, nav->map_pos, (uint16_t)r->radius); break;
        }
    }
    int njobs = (int)s_field_work.nwork;
    for(int i = 0; i < njobs && i < max_ids; i++)
        out_ids[i] = vec_AT(&s_field_work.in, i).id;
    if(s_use_binding)
        N_HIP_BuildAsyncFields();                  /* (the statement added to compute_async_fields) */
    N_AwaitAsyncFields();
    vec_in_destroy(&s_field_work.in);
    vec_out_destroy(&s_field_work.out);
    return njobs;
}

int pfref_cached_field_by_id(pfref_nav *nav, uint64_t ffid, uint8_t *out_dirs)
{
    if(!N_FC_ContainsFlowField(nav->priv.fieldcache, ffid))
        return 0;
    const struct flow_field *ff = N_FC_FlowFieldAt(nav->priv.fieldcache, ffid);
    if(!ff)
        return 0;
    pfref_ff_to_dirs(ff, out_dirs);
    return 1;
}

void pfref_hip_async_stats(long out[3])  { N_HIP_AsyncStats(out); }
void pfref_hip_los_stats(long out[2])    { N_HIP_LOSStats(out); }
void pfref_hip_blockers_stats(long out[2]) { N_HIP_BlockersStats(out); }
int  pfref_hip_blockers_flush(void)      { return N_HIP_BlockersFlush() ? 1 : 0; }
void *pfref_hip_ctx(void)                { return N_HIP_Ctx(); }

/* the chunks N_Update (nav.c:2119) would invalidate for `layer`: the dirty set n_update_blockers leaves
 * (nav.c:1033-1046); flags[h*w] */
void pfref_nav_dirty_chunks(pfref_nav *nav, int layer, uint8_t *flags)
{
    struct nav_private *priv = &nav->priv;
    memset(flags, 0, priv->width * priv->height);
    khash_t(coord) *set = priv->dirty_chunks[layer];
    for(int i = kh_begin(set); i != kh_end(set); i++) {
        if(!kh_exist(set, i))
            continue;
        uint32_t key = kh_key(set, i);                    /* nav.c:1008-1009 */
        flags[(key >> 16) * priv->width + (key & 0xffff)] = 1;
    }
}


/* N_DesiredEnemySeekVelocity (nav.c:3603), N_DesiredSurroundVelocity (:3687), N_DesiredGroupArrivalVelocity
 * (:3561) for n agents; kind as in pfref_async_req.  out_xz[n][2]; out_flags[n]: bit 0 = the group-arrival
 * lookup found a field (its return value), bit 1 = at_slot */
void pfref_desired_region_velocities(pfref_nav *nav, int n, const pfref_async_req *reqs, float *out_xz, uint8_t *out_flags)
{
    struct nav_private *priv = &nav->priv;
    for(int i = 0; i < n; i++) {
        const pfref_async_req *r = &reqs[i];
        vec2_t xz = (vec2_t){r->x, r->z}, v = (vec2_t){0.0f, 0.0f};
        out_flags[i] = 0;
        switch(r->kind) {
        case 0: v = N_DesiredEnemySeekVelocity(xz, priv, r->layer, nav->map_pos, r->faction_id); break;
        case 1: v = N_DesiredSurroundVelocity(xz, priv, r->layer, nav->map_pos, r->ent, r->faction_id); break;
        case 2: {
            bool at_slot = false;
            /* (x, z) = the agent; the zone centre travels in (ent as float bits, radius): see pfref.py */
            vec2_t centre;
            memcpy(&centre.x, &r->ent, sizeof(float));
            memcpy(&centre.z, &r->faction_id, sizeof(float));
            bool ok = N_DesiredGroupArrivalVelocity(xz, priv, r->layer, nav->map_pos, centre, (uint16_t)r->radius, &v, &at_slot);
            out_flags[i] = (ok ? 1 : 0) | (at_slot ? 2 : 0);
            break;
        }
        }
        out_xz[2 * i] = v.x; out_xz[2 * i + 1] = v.z;
    }
}


/* the two per-destination nav queries behind arrived() (movement.c:2170): N_ClosestPathable (nav.c:4126) and the
 * tile list N_IsMaximallyClose tests (n_closest_island_tiles of the destination, nav.c:4716-4725) */
int pfref_closest_pathable(pfref_nav *nav, int layer, float x, float z, float out[2])
{
    vec2_t v = (vec2_t){0.0f, 0.0f};
    bool ok = N_ClosestPathable(&nav->priv, layer, nav->map_pos, (vec2_t){x, z}, &v);
    out[0] = v.x; out[1] = v.z;
    return ok ? 1 : 0;
}

int pfref_dest_island_tiles(pfref_nav *nav, int layer, float x, float z, int16_t *out_abs, int max_tiles)
{
    return N_HIP_ClosestIslandTiles(&nav->priv, (enum nav_layer)layer, nav->map_pos, (vec2_t){x, z}, out_abs, max_tiles);
}
