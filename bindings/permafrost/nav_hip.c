/* bindings/permafrost/nav_hip.c -- the nav.c half of the reference-side binding of libnavhip.so.
 *
 * This is the file INTEGRATION.md tells a maintainer of permafrost-engine to add as
 * src/navigation/nav_hip.c (appended to nav.c's translation unit, because the batched sampler needs
 * nav.c's static n_request_path and n_interpolated_flow_dir, and the asynchronous batch its static
 * s_field_work).  It is written against the reference's own headers; in this repository it is compiled
 * and executed inside the test harness (oracle/ref/ref_nav.c #includes it right after nav.c), where the
 * reference's planner, field cache, sampler and async field batch drive the HIP library through
 * include/navhip.h.  Companion files: field_hip.c (appended to field.c), move_hip.c (to movement.c).
 *
 * What it binds
 *   N_HIP_FlowFieldUpdate                    same signature as N_FlowFieldUpdate (field.h:146): one
 *   N_HIP_FlowFieldUpdateToNearestPathable   chunk field through the device, drop-in for the call
 *   N_HIP_FlowFieldUpdateIslandToNearest     sites nav.c:1831,2001,2017,3530,3546
 *   deferred mode                            the same three entry points only RECORD the build and
 *                                            hand back a tagged placeholder; N_HIP_Flush() builds the
 *                                            whole batch in one navhip_build_fields call (in-place
 *                                            chains ordered into rounds) and puts the results into
 *                                            the reference's field cache under their own ids
 *   N_HIP_DesiredPointSeekVelocities         N_DesiredPointSeekVelocity (nav.c:3468) for n agents:
 *                                            the miss-collecting pre-pass.  The serial control flow of
 *                                            nav.c:3483-3554 runs unchanged per agent, but in PHASES
 *                                            with one batched device build between them:
 *                                            (A) missing (dest, chunk) mapping -> n_request_path,
 *                                            (B) FD_NONE under the agent -> n_request_path again,
 *                                            (C) repairs (blocked tile / orphaned island), one per
 *                                                field and round,
 *                                            then n_interpolated_flow_dir on the now complete cache.
 *   N_HIP_BuildAsyncFields                   the asynchronous field batch (N_PrepareAsyncWork /
 *                                            N_RequestAsync{EnemySeek,Surround,GroupArrival}Field /
 *                                            field_task / N_AwaitAsyncFields, nav.c:3767-3969,2049): the
 *                                            request functions stay as they are but create no fiber
 *                                            (N_HIP_FieldTaskCreate instead of Sched_Create); before
 *                                            N_AwaitAsyncFields joins and N_FC_PutFlowField's the results,
 *                                            this builds all <= 256 jobs in ONE navhip_build_region_fields
 *                                            call -- the game-side frontier extraction of every job on the
 *                                            host (field_hip.c), the grid work on the device
 *   N_HIP_LOSFieldCreate                     same signature as N_LOSFieldCreate (field.h:195): the call
 *                                            sites nav.c:1843,2035; immediate, or recorded in deferred mode
 *                                            and built per chain level (a LOS field needs its predecessor's)
 *   N_HIP_BlockersRecord / N_HIP_BlockersFlush   N_BlockersIncref / N_BlockersDecref (nav.c:4663,4674) keep
 *                                            updating the host planes (the planner reads them); the same
 *                                            circles, recorded, update the device planes in one
 *                                            navhip_blockers_circles call per tick -- no plane re-upload
 * Backend 0 runs the same phases with the reference's own CPU builders -- the control of the tests.
 */
#include <navhip.h>

enum { N_HIP_BACKEND_CPU = 0, N_HIP_BACKEND_HIP = 1 };

static struct {
    navhip_ctx *ctx;
    int         backend;
    bool        deferred;
    /* pending builds (deferred mode) */
    struct n_hip_pending{
        ff_id_t             id;
        struct coord        chunk;
        struct field_target target;
        int                 faction_id;
        enum nav_layer      layer;
        int                 kind;          /* 0 update, 1 nearest pathable, 2 island to nearest */
        struct coord        start;         /* kind 1 */
        uint16_t            local_iid;     /* kind 2 */
        int                 base;          /* index of the pending build this one updates in place, -1 */
        uint8_t             dirs[FIELD_RES_R * FIELD_RES_C];   /* existing content (base == -1) / result */
        struct nav_private *priv;
    }          *pend;
    int         npend, cappend;
    /* statistics */
    long        n_builds, n_batches, n_requests;
}s_hip;

#define N_HIP_TAG 0xF    /* dir_idx value no real field holds: marks a placeholder */

/* the device image of the field cache (the functions are at the end of this file) */
static struct{
    bool        on;
    int         n_rows;
    dest_id_t  *row_dest;
    bool       *row_used;
    int32_t    *m_row; uint16_t *m_r, *m_c; uint64_t *m_id;
    int         nm, capm;
    long        n_puts, n_maps, n_built;
}s_hip_pool;


/* -------------------------------------------------------------------------------------------- */

static void n_hip_pack_plane(const struct nav_private *priv, enum nav_layer layer, int plane, void *out)
{
    size_t nchunks = priv->width * priv->height;
    const size_t cells = FIELD_RES_R * FIELD_RES_C;
    for(size_t i = 0; i < nchunks; i++) {
        const struct nav_chunk *ch = &priv->chunks[layer][i];
        switch(plane) {
        case NAVHIP_PLANE_COST_BASE:     memcpy((uint8_t*)out  + i * cells, ch->cost_base, cells); break;
        case NAVHIP_PLANE_BLOCKERS:      memcpy((uint16_t*)out + i * cells, ch->blockers, cells * 2); break;
        case NAVHIP_PLANE_LOCAL_ISLANDS: memcpy((uint16_t*)out + i * cells, ch->local_islands, cells * 2); break;
        case NAVHIP_PLANE_ISLANDS:       memcpy((uint16_t*)out + i * cells, ch->islands, cells * 2); break;
        case NAVHIP_PLANE_FACTIONS:      memcpy((uint8_t*)out  + i * cells * MAX_FACTIONS, ch->factions, cells * MAX_FACTIONS); break;
        }
    }
}

/* upload every plane of `layer` the field builders read (after N_NewCtxForMapData, and again after
 * N_Update has applied blocker changes and relabelled the dirty local islands, nav.c:2119) */
bool N_HIP_SyncLayer(const struct nav_private *priv, enum nav_layer layer)
{
    if(!s_hip.ctx || !priv->chunks[layer])
        return false;
    size_t nchunks = priv->width * priv->height;
    const size_t cells = FIELD_RES_R * FIELD_RES_C;
    void *buf = malloc(nchunks * cells * MAX_FACTIONS);
    if(!buf)
        return false;
    static const int planes[] = {NAVHIP_PLANE_COST_BASE, NAVHIP_PLANE_BLOCKERS, NAVHIP_PLANE_LOCAL_ISLANDS,
                                 NAVHIP_PLANE_ISLANDS, NAVHIP_PLANE_FACTIONS};
    static const size_t elem[] = {1, 2, 2, 2, MAX_FACTIONS};
    bool ok = true;
    for(int p = 0; p < 5 && ok; p++) {
        n_hip_pack_plane(priv, layer, planes[p], buf);
        ok = navhip_upload_plane(s_hip.ctx, layer, planes[p], buf, nchunks * cells * elem[p]) == NAVHIP_OK;
    }
    free(buf);
    return ok;
}

bool N_HIP_Init(const struct nav_private *priv)
{
    memset(&s_hip, 0, sizeof(s_hip));
    if(navhip_ctx_create(&s_hip.ctx, priv->width, priv->height, 0) != NAVHIP_OK) {
        s_hip.ctx = NULL;
        return false;                          /* no GPU: the CPU path stays in charge */
    }
    for(int l = 0; l < NAV_LAYER_MAX; l++) {
        if(priv->chunks[l] && !N_HIP_SyncLayer(priv, l))
            return false;
    }
    s_hip.backend = N_HIP_BACKEND_HIP;
    return true;
}

void N_HIP_Shutdown(void)
{
    if(s_hip.ctx)
        navhip_ctx_destroy(s_hip.ctx);
    free(s_hip.pend);
    memset(&s_hip, 0, sizeof(s_hip));
}

void N_HIP_SetBackend(int backend) { s_hip.backend = backend; }
navhip_ctx *N_HIP_Ctx(void)        { return s_hip.ctx; }
void N_HIP_Stats(long out[3])      { out[0] = s_hip.n_builds; out[1] = s_hip.n_batches; out[2] = s_hip.n_requests; }

/* struct field_target / portal_desc (field.h:67-101) -> navhip_field_req */
static bool n_hip_make_req(const struct n_hip_pending *p, navhip_field_req *r)
{
    memset(r, 0, sizeof(*r));
    r->layer = p->layer;
    r->faction_id = p->faction_id;
    r->enemies = (p->faction_id == FACTION_ID_NONE) ? 0 : G_GetEnemyFactions(p->faction_id);   /* field.c:166 */
    r->chunk_r = p->chunk.r; r->chunk_c = p->chunk.c;
    r->flags = NAVHIP_REQ_INOUT;               /* an all-FD_NONE slot == N_FlowFieldInit */
    if(p->target.type == TARGET_TILE) {
        r->type = NAVHIP_TARGET_TILE;
        r->tile_r = p->target.tile.r; r->tile_c = p->target.tile.c;
    }else if(p->target.type == TARGET_PORTAL) {
        const struct portal_desc *pd = &p->target.pd;
        r->type = NAVHIP_TARGET_PORTAL;
        r->port_r0 = pd->port->endpoints[0].r; r->port_c0 = pd->port->endpoints[0].c;
        r->port_r1 = pd->port->endpoints[1].r; r->port_c1 = pd->port->endpoints[1].c;
        r->next_r0 = pd->next->endpoints[0].r; r->next_c0 = pd->next->endpoints[0].c;
        r->next_r1 = pd->next->endpoints[1].r; r->next_c1 = pd->next->endpoints[1].c;
        r->next_chunk_r = pd->next->chunk.r;   r->next_chunk_c = pd->next->chunk.c;
        r->port_iid = pd->port_iid;            r->next_iid = pd->next_iid;
    }else{
        return false;
    }
    if(p->kind == 1) {                         /* N_FlowFieldUpdateToNearestPathable: ignores the target */
        r->type = NAVHIP_TARGET_NEAREST_PATHABLE;
        r->tile_r = p->start.r; r->tile_c = p->start.c;
    }else if(p->kind == 2) {                   /* N_FlowFieldUpdateIslandToNearest */
        r->flags |= NAVHIP_REQ_ISLAND_NEAREST;
        r->aux_iid = p->local_iid;
    }
    return true;
}

static void n_hip_ff_to_dirs(const struct flow_field *ff, uint8_t *dirs)
{
    for(int r = 0; r < FIELD_RES_R; r++)
    for(int c = 0; c < FIELD_RES_C; c++)
        dirs[r * FIELD_RES_C + c] = ff->field[r][c].dir_idx;
}

static void n_hip_dirs_to_ff(const uint8_t *dirs, struct flow_field *ff)
{
    for(int r = 0; r < FIELD_RES_R; r++)
    for(int c = 0; c < FIELD_RES_C; c++)
        ff->field[r][c].dir_idx = dirs[r * FIELD_RES_C + c];
}

/* the reference's own builders (nav.c's calls are renamed onto the hooks below by the harness, so
 * these are reached through the real symbols) */
static void n_hip_cpu_build(const struct n_hip_pending *p, struct flow_field *ff)
{
    switch(p->kind) {
    case 0: (N_FlowFieldUpdate)(p->chunk, p->priv, p->faction_id, p->layer, p->target,
                p->priv->unit_query_ctx, ff); break;
    case 1: (N_FlowFieldUpdateToNearestPathable)(p->priv, p->layer, p->chunk, p->start, p->faction_id,
                p->priv->unit_query_ctx, ff); break;
    case 2: (N_FlowFieldUpdateIslandToNearest)(p->local_iid, p->priv, p->layer, p->faction_id,
                p->priv->unit_query_ctx, ff); break;
    }
}

/* one entry: record or build */
static void n_hip_entry(struct n_hip_pending *p, struct flow_field *inout_flow)
{
    s_hip.n_requests++;
    navhip_field_req req;
    bool on_device = s_hip.backend == N_HIP_BACKEND_HIP && s_hip.ctx && n_hip_make_req(p, &req);
    if(p->kind == 0)
        inout_flow->target = p->target;        /* N_FlowFieldUpdate records its target (field.c:2076) */

    if(s_hip.deferred) {
        /* is the field we are asked to update itself a placeholder of this batch? */
        p->base = -1;
        if(inout_flow->field[0][0].dir_idx == N_HIP_TAG) {
            int idx = 0;
            for(int k = 0; k < 6; k++)
                idx |= inout_flow->field[0][1 + k].dir_idx << (4 * k);
            p->base = idx;
        }else{
            n_hip_ff_to_dirs(inout_flow, p->dirs);
        }
        if(s_hip.npend == s_hip.cappend) {
            s_hip.cappend = s_hip.cappend ? s_hip.cappend * 2 : 256;
            s_hip.pend = realloc(s_hip.pend, sizeof(*s_hip.pend) * s_hip.cappend);
        }
        int me = s_hip.npend++;
        s_hip.pend[me] = *p;
        /* the placeholder: tag + the index of this build; chunk stays what the caller set */
        inout_flow->field[0][0].dir_idx = N_HIP_TAG;
        for(int k = 0; k < 6; k++)
            inout_flow->field[0][1 + k].dir_idx = (me >> (4 * k)) & 0xf;
        return;
    }
    if(!on_device) {
        n_hip_cpu_build(p, inout_flow);
        return;
    }
    uint8_t dirs[FIELD_RES_R * FIELD_RES_C];
    n_hip_ff_to_dirs(inout_flow, dirs);
    if(navhip_build_fields(s_hip.ctx, &req, 1, dirs, NULL) != NAVHIP_OK) {
        n_hip_cpu_build(p, inout_flow);        /* any error: the CPU path stays compiled in */
        return;
    }
    s_hip.n_builds++; s_hip.n_batches++;
    n_hip_dirs_to_ff(dirs, inout_flow);
}

/* ---- the three drop-in entry points (signatures of field.h:146-183) ------------------------- */

void N_HIP_FlowFieldUpdate(struct coord chunk_coord, const struct nav_private *priv, int faction_id,
                           enum nav_layer layer, struct field_target target,
                           struct nav_unit_query_ctx *ctx, struct flow_field *inout_flow)
{
    if(target.type != TARGET_TILE && target.type != TARGET_PORTAL) {
        /* enemy / entity / zone fields: the region builders (host, or navhip_build_region_fields) */
        (N_FlowFieldUpdate)(chunk_coord, priv, faction_id, layer, target, ctx, inout_flow);
        return;
    }
    struct n_hip_pending p;
    memset(&p, 0, sizeof(p));
    p.id = N_FlowFieldID(chunk_coord, target, layer);
    p.chunk = chunk_coord; p.target = target; p.faction_id = faction_id; p.layer = layer;
    p.kind = 0; p.priv = (struct nav_private*)priv;
    n_hip_entry(&p, inout_flow);
}

void N_HIP_FlowFieldUpdateToNearestPathable(const struct nav_private *priv, enum nav_layer layer,
                                            struct coord chunk, struct coord start, int faction_id,
                                            struct nav_unit_query_ctx *ctx, struct flow_field *inout_flow)
{
    struct n_hip_pending p;
    memset(&p, 0, sizeof(p));
    p.chunk = chunk; p.target = inout_flow->target; p.faction_id = faction_id; p.layer = layer;
    p.id = N_FlowFieldID(chunk, inout_flow->target, layer);
    p.kind = 1; p.start = start; p.priv = (struct nav_private*)priv;
    if(p.target.type != TARGET_TILE && p.target.type != TARGET_PORTAL) {
        (N_FlowFieldUpdateToNearestPathable)(priv, layer, chunk, start, faction_id, ctx, inout_flow);
        return;
    }
    n_hip_entry(&p, inout_flow);
}

void N_HIP_FlowFieldUpdateIslandToNearest(uint16_t local_iid, const struct nav_private *priv,
                                          enum nav_layer layer, int faction_id,
                                          struct nav_unit_query_ctx *ctx, struct flow_field *inout_flow)
{
    struct n_hip_pending p;
    memset(&p, 0, sizeof(p));
    p.chunk = inout_flow->chunk; p.target = inout_flow->target; p.faction_id = faction_id; p.layer = layer;
    p.id = N_FlowFieldID(inout_flow->chunk, inout_flow->target, layer);
    p.kind = 2; p.local_iid = local_iid; p.priv = (struct nav_private*)priv;
    if(p.target.type != TARGET_TILE && p.target.type != TARGET_PORTAL) {
        (N_FlowFieldUpdateIslandToNearest)(local_iid, priv, layer, faction_id, ctx, inout_flow);
        return;
    }
    n_hip_entry(&p, inout_flow);
}

/* ---- deferred mode --------------------------------------------------------------------------- */

static bool n_hip_flush_los(void);     /* the LOS fields recorded in the same batch (below) */
static void n_hip_los_reset(void);

void N_HIP_BeginBatch(void) { s_hip.deferred = true; s_hip.npend = 0; n_hip_los_reset(); }

/* Build every recorded field -- one navhip_build_fields call per round (a build that updates another
 * pending build in place waits for it) -- and put the results into the field cache. */
bool N_HIP_Flush(void)
{
    s_hip.deferred = false;
    const int n = s_hip.npend;
    if(n == 0)
        return n_hip_flush_los();
    bool ok = true;
    int *round = malloc(sizeof(int) * n);
    int maxr = 0;
    for(int i = 0; i < n; i++) {
        round[i] = s_hip.pend[i].base < 0 ? 0 : round[s_hip.pend[i].base] + 1;     /* base < i always */
        if(round[i] > maxr) maxr = round[i];
    }
    navhip_field_req *reqs = malloc(sizeof(navhip_field_req) * n);
    uint8_t *dirs = malloc((size_t)n * FIELD_RES_R * FIELD_RES_C);
    int *who = malloc(sizeof(int) * n);
    for(int r = 0; r <= maxr; r++) {
        int m = 0;
        for(int i = 0; i < n; i++) {
            if(round[i] != r) continue;
            struct n_hip_pending *p = &s_hip.pend[i];
            if(p->base >= 0)
                memcpy(p->dirs, s_hip.pend[p->base].dirs, sizeof(p->dirs));
            bool dev = s_hip.backend == N_HIP_BACKEND_HIP && s_hip.ctx && n_hip_make_req(p, &reqs[m]);
            if(!dev) {
                struct flow_field ff;
                memset(&ff, 0, sizeof(ff));
                ff.chunk = p->chunk; ff.target = p->target;
                n_hip_dirs_to_ff(p->dirs, &ff);
                n_hip_cpu_build(p, &ff);
                n_hip_ff_to_dirs(&ff, p->dirs);
                continue;
            }
            memcpy(dirs + (size_t)m * FIELD_RES_R * FIELD_RES_C, p->dirs, sizeof(p->dirs));
            who[m++] = i;
        }
        if(m > 0) {
            int rc;
            if(s_hip_pool.on) {
                /* build INTO the resident pool (and read back for the host cache): the fields never
                 * cross the bus a second time.  An update in place starts from the pending build it
                 * names (base id) or from the host's existing content, which is put first */
                uint64_t *ids = malloc(sizeof(uint64_t) * m), *bases = calloc(m, sizeof(uint64_t));
                for(int k = 0; k < m; k++) {
                    struct n_hip_pending *p = &s_hip.pend[who[k]];
                    ids[k] = p->id;
                    if(p->base >= 0) {
                        if(s_hip.pend[p->base].id != p->id) bases[k] = s_hip.pend[p->base].id;
                        continue;
                    }
                    bool any = false;
                    for(size_t t = 0; t < sizeof(p->dirs); t++) any = any || p->dirs[t] != FD_NONE;
                    if(any)
                        navhip_pool_put(s_hip.ctx, p->id, p->dirs);
                    else if(p->kind == 0)
                        reqs[k].flags &= ~NAVHIP_REQ_INOUT;        /* N_FlowFieldInit, not the slot's old field */
                    else
                        navhip_pool_put(s_hip.ctx, p->id, p->dirs); /* a repair of an all-FD_NONE field */
                }
                rc = navhip_pool_build(s_hip.ctx, reqs, ids, bases, m, dirs);
                if(rc == NAVHIP_OK) s_hip_pool.n_built += m;
                free(ids); free(bases);
            }else{
                rc = navhip_build_fields(s_hip.ctx, reqs, m, dirs, NULL);
            }
            if(rc == NAVHIP_OK) {
                s_hip.n_builds += m; s_hip.n_batches++;
                for(int k = 0; k < m; k++)
                    memcpy(s_hip.pend[who[k]].dirs, dirs + (size_t)k * FIELD_RES_R * FIELD_RES_C,
                           sizeof(s_hip.pend[0].dirs));
            }else{
                for(int k = 0; k < m; k++) {               /* fall back to the CPU builders */
                    struct n_hip_pending *p = &s_hip.pend[who[k]];
                    struct flow_field ff;
                    memset(&ff, 0, sizeof(ff));
                    ff.chunk = p->chunk; ff.target = p->target;
                    n_hip_dirs_to_ff(p->dirs, &ff);
                    n_hip_cpu_build(p, &ff);
                    n_hip_ff_to_dirs(&ff, p->dirs);
                }
                ok = false;
            }
        }
    }
    /* N_FC_PutFlowField (nav.c:1833,2008,2018,3533,3547) with the real contents, in request order
     * (a later build of the same id is the one that stays, as in the serial run) */
    for(int i = 0; i < n; i++) {
        struct n_hip_pending *p = &s_hip.pend[i];
        struct flow_field ff;
        memset(&ff, 0, sizeof(ff));
        ff.chunk = p->chunk; ff.target = p->target;
        n_hip_dirs_to_ff(p->dirs, &ff);
        N_FC_PutFlowField(p->priv->fieldcache, p->id, &ff);
    }
    free(round); free(reqs); free(dirs); free(who);
    s_hip.npend = 0;
    return n_hip_flush_los() && ok;
}

/* ---- N_DesiredPointSeekVelocity for n agents (nav.c:3468-3559) ------------------------------- */

static bool n_hip_tile(struct nav_private *priv, vec3_t map_pos, vec2_t pos, struct tile_desc *out)
{
    struct map_resolution res;
    N_GetResolution(priv, &res);
    return M_Tile_DescForPoint2D(res, map_pos, pos, out);
}

static const struct flow_field *n_hip_field_under(struct nav_private *priv, dest_id_t id, struct tile_desc tile,
                                                 ff_id_t *out_ffid)
{
    if(!N_FC_GetDestFFMapping(priv->fieldcache, id, (struct coord){tile.chunk_r, tile.chunk_c}, out_ffid))
        return NULL;
    return N_FC_FlowFieldAt(priv->fieldcache, *out_ffid);
}

/* ids[i] / pos[i] / dest[i]: the arguments of the i-th N_DesiredPointSeekVelocity call; out[i] its
 * return value (zero for the `return (vec2_t){0.0f}` exits of nav.c:3489,3501).
 *
 * Every agent walks through the serial code's own decisions, one decision per round, and every
 * decision that builds fields only records them; the round's builds run as one batch.  An agent
 * whose field is being rebuilt in the current round (tagged placeholder in the cache) waits for the
 * next round, exactly as it would have seen the finished field in the serial run. */
void N_HIP_DesiredPointSeekVelocities(struct nav_private *priv, vec3_t map_pos, int n, const dest_id_t *ids,
                                      const vec2_t *pos, const vec2_t *dest, vec2_t *out)
{
    struct map_resolution res;
    N_GetResolution(priv, &res);
    enum { ST_A = 0, ST_B, ST_C, ST_SAMPLE, ST_ZERO };
    uint8_t *st = calloc(n > 0 ? n : 1, 1);

    for(int round = 0; round < 256; round++) {
        int issued = 0, waiting = 0;
        N_HIP_BeginBatch();
        for(int i = 0; i < n; i++) {
            if(st[i] >= ST_SAMPLE) continue;
            struct tile_desc tile;
            if(!n_hip_tile(priv, map_pos, pos[i], &tile)) { st[i] = ST_ZERO; continue; }
            const struct coord chunk = (struct coord){tile.chunk_r, tile.chunk_c};
            enum nav_layer layer = N_DestLayer(ids[i]);
            int faction_id = N_DestFactionID(ids[i]);
            ff_id_t ffid;
            dest_id_t ret;
            if(st[i] == ST_A) {
                /* nav.c:3483-3492: no field mapped for the agent's chunk */
                st[i] = ST_B;
                if(!N_FC_GetDestFFMapping(priv->fieldcache, ids[i], chunk, &ffid)) {
                    if(!n_request_path(priv, pos[i], dest[i], faction_id, map_pos, layer, &ret))
                        st[i] = ST_ZERO;
                    issued++;
                    continue;
                }
            }
            const struct flow_field *ff = n_hip_field_under(priv, ids[i], tile, &ffid);
            if(ff && ff->field[0][0].dir_idx == N_HIP_TAG) { waiting++; continue; }
            if(st[i] == ST_B) {
                /* nav.c:3494-3504: evicted, or FD_NONE under the agent: ask the planner from here */
                st[i] = ST_C;
                if(!ff || ff->field[tile.tile_r][tile.tile_c].dir_idx == FD_NONE) {
                    if(!n_request_path(priv, pos[i], dest[i], faction_id, map_pos, layer, &ret))
                        st[i] = ST_ZERO;
                    issued++;
                    continue;
                }
            }
            /* nav.c:3506-3554 */
            st[i] = ST_SAMPLE;
            if(!ff) { st[i] = ST_ZERO; continue; }
            if(ff->field[tile.tile_r][tile.tile_c].dir_idx != FD_NONE)
                continue;
            const struct nav_chunk *nchunk = &priv->chunks[layer][IDX(tile.chunk_r, priv->width, tile.chunk_c)];
            uint16_t local_iid = nchunk->local_islands[tile.tile_r][tile.tile_c];
            struct flow_field exist_ff = *ff;
            if(local_iid == ISLAND_NONE) {
                N_HIP_FlowFieldUpdateToNearestPathable(priv, layer, chunk,
                    (struct coord){tile.tile_r, tile.tile_c}, faction_id, priv->unit_query_ctx, &exist_ff);
            }else{
                N_HIP_FlowFieldUpdateIslandToNearest(local_iid, priv, layer, faction_id, priv->unit_query_ctx, &exist_ff);
            }
            N_FC_PutFlowField(priv->fieldcache, ffid, &exist_ff);
            issued++;
        }
        N_HIP_Flush();
        if(!issued && !waiting) break;
    }

    /* sampling (nav.c:3556-3558) */
    for(int i = 0; i < n; i++) {
        out[i] = (vec2_t){0.0f, 0.0f};
        if(st[i] == ST_ZERO) continue;
        struct tile_desc tile;
        if(!n_hip_tile(priv, map_pos, pos[i], &tile)) continue;
        ff_id_t ffid;
        const struct flow_field *ff = n_hip_field_under(priv, ids[i], tile, &ffid);
        if(!ff) continue;
        out[i] = n_interpolated_flow_dir(priv, ids[i], res, map_pos, pos[i], ff, tile);
    }
    free(st);
}

/* ---- the asynchronous field batch (nav.c:3767-3969, field_task :2049) ------------------------ */

bool N_HIP_RegionRequest(struct coord chunk_coord, const struct nav_private *priv, enum nav_layer layer,
                         struct field_target target, struct nav_unit_query_ctx *ctx,
                         navhip_region_req *out_req, int16_t *out_seeds, size_t max_seeds, size_t *out_nseeds);   /* field_hip.c */

static struct{
    long n_jobs, n_batches, n_cpu_jobs;
}s_hip_async;

void N_HIP_AsyncStats(long out[3]) { out[0] = s_hip_async.n_jobs; out[1] = s_hip_async.n_batches; out[2] = s_hip_async.n_cpu_jobs; }

/* What N_RequestAsync*Field call instead of Sched_Create(1, field_task, arg, ...) (nav.c:3824,3878,3907):
 * with the device in charge no fiber is created -- the job stays in s_field_work.in[] and its future is
 * complete at once, so that field_join_work (:2062) has nothing to wait for; N_HIP_BuildAsyncFields fills
 * s_field_work.out[] before N_AwaitAsyncFields puts the results into the cache.  Without the device:
 * the reference's own Sched_Create. */
uint32_t N_HIP_FieldTaskCreate(int prio, task_func_t code, void *arg, const char *name,
                               struct future *result, int flags)
{
    if(s_hip.backend == N_HIP_BACKEND_HIP && s_hip.ctx) {
        SDL_AtomicSet(&result->status, FUTURE_COMPLETE);
        return 1;                              /* any tid but NULL_TID: the job counts (nav.c:3826) */
    }
    return (Sched_Create)(prio, code, arg, name, result, flags);
}

/* Called by the movement tick between the last N_RequestAsync*Field and N_AwaitAsyncFields
 * (compute_async_fields, movement.c:4149-4164): every job of the batch in ONE device call. */
bool N_HIP_BuildAsyncFields(void)
{
    const int n = (int)s_field_work.nwork;
    if(n == 0 || !(s_hip.backend == N_HIP_BACKEND_HIP && s_hip.ctx))
        return true;                           /* the fibers built (or are building) them */
    const size_t max_seeds_per = (size_t)(2 * FIELD_RES_R) * (2 * FIELD_RES_C);
    navhip_region_req *reqs = malloc(sizeof(navhip_region_req) * n);
    uint8_t *dirs = calloc((size_t)n, FIELD_RES_R * FIELD_RES_C);       /* N_FlowFieldInit: FD_NONE == 0 */
    int16_t *seeds = NULL;
    size_t nseeds = 0, capseeds = 0;
    int *who = malloc(sizeof(int) * n);
    int16_t *tmp = malloc(sizeof(int16_t) * 2 * max_seeds_per);
    int m = 0;
    for(int i = 0; i < n; i++) {
        struct field_work_in *in = &vec_AT(&s_field_work.in, i);
        struct field_work_out *out = &vec_AT(&s_field_work.out, i);
        size_t k = 0;
        if(!N_HIP_RegionRequest(in->chunk, in->priv, in->layer, in->target, in->priv->unit_query_ctx,
                                &reqs[m], tmp, max_seeds_per, &k)) {
            /* not a region target the device covers: what field_task does (nav.c:2055-2057) */
            N_FlowFieldInit(in->chunk, &out->field);
            (N_FlowFieldUpdate)(in->chunk, in->priv, in->faction_id, in->layer, in->target,
                in->priv->unit_query_ctx, &out->field);
            s_hip_async.n_cpu_jobs++;
            continue;
        }
        if(nseeds + k > capseeds) {
            capseeds = (nseeds + k) * 2 + 1024;
            seeds = realloc(seeds, sizeof(int16_t) * 2 * capseeds);
        }
        memcpy(seeds + 2 * nseeds, tmp, sizeof(int16_t) * 2 * k);
        reqs[m].seed_begin = (uint32_t)nseeds;
        nseeds += k;
        who[m++] = i;
    }
    bool ok = true;
    if(m > 0) {
        ok = navhip_build_region_fields(s_hip.ctx, reqs, m, seeds, nseeds, NULL, 0, dirs,
                                        FIELD_RES_R * FIELD_RES_C) == NAVHIP_OK;
        s_hip_async.n_batches++;
    }
    for(int k = 0; k < m; k++) {
        struct field_work_in *in = &vec_AT(&s_field_work.in, who[k]);
        struct field_work_out *out = &vec_AT(&s_field_work.out, who[k]);
        N_FlowFieldInit(in->chunk, &out->field);
        if(ok) {
            n_hip_dirs_to_ff(dirs + (size_t)k * FIELD_RES_R * FIELD_RES_C, &out->field);
            out->field.target = in->target;    /* field_update_enemies / _entity / _zone record it (:1589,:1660,:1872) */
            s_hip_async.n_jobs++;
        }else{
            (N_FlowFieldUpdate)(in->chunk, in->priv, in->faction_id, in->layer, in->target,
                in->priv->unit_query_ctx, &out->field);
            s_hip_async.n_cpu_jobs++;
        }
    }
    free(reqs); free(dirs); free(seeds); free(who); free(tmp);
    return ok;
}

/* ---- line-of-sight fields (N_LOSFieldCreate, field.c:2085; call sites nav.c:1843,2035) ------- */

static struct{
    struct n_hip_los_pending{
        dest_id_t        id;
        struct coord     chunk;
        struct tile_desc target;
        struct nav_private *priv;
        vec3_t           map_pos;
        int              base;                     /* pending index of prev_los, -1: given / none */
        bool             has_prev;
        struct coord     prev_chunk;
        uint8_t          prev[FIELD_RES_R * FIELD_RES_C];
        uint8_t          out[FIELD_RES_R * FIELD_RES_C];
    }          *pend;
    int         npend, cappend;
    long        n_builds, n_batches;
}s_hip_los;

void N_HIP_LOSStats(long out[2]) { out[0] = s_hip_los.n_builds; out[1] = s_hip_los.n_batches; }
static void n_hip_los_reset(void) { s_hip_los.npend = 0; }

static void n_hip_los_to_bytes(const struct LOS_field *lf, uint8_t *out)
{
    for(int r = 0; r < FIELD_RES_R; r++)
    for(int c = 0; c < FIELD_RES_C; c++)
        out[r * FIELD_RES_C + c] = (uint8_t)(lf->field[r][c].visible | (lf->field[r][c].wavefront_blocked << 1));
}

static void n_hip_bytes_to_los(const uint8_t *in, struct coord chunk, struct LOS_field *lf)
{
    lf->chunk = chunk;                             /* field.c:2091 */
    for(int r = 0; r < FIELD_RES_R; r++)
    for(int c = 0; c < FIELD_RES_C; c++) {
        lf->field[r][c].visible = in[r * FIELD_RES_C + c] & 1;
        lf->field[r][c].wavefront_blocked = (in[r * FIELD_RES_C + c] >> 1) & 1;
    }
}

static void n_hip_make_los_req(const struct n_hip_los_pending *p, navhip_los_req *r)
{
    memset(r, 0, sizeof(*r));
    r->layer = N_DestLayer(p->id);
    r->faction_id = N_DestFactionID(p->id);
    r->enemies = (r->faction_id == FACTION_ID_NONE) ? 0 : G_GetEnemyFactions(r->faction_id);   /* enemy_faction_from, field.c:151 */
    r->chunk_r = p->chunk.r; r->chunk_c = p->chunk.c;
    r->target_chunk_r = p->target.chunk_r; r->target_chunk_c = p->target.chunk_c;
    r->target_tile_r = p->target.tile_r;   r->target_tile_c = p->target.tile_c;
    if(p->has_prev) {
        r->prev_dr = (int8_t)(p->prev_chunk.r - p->chunk.r);
        r->prev_dc = (int8_t)(p->prev_chunk.c - p->chunk.c);
    }
}

void N_HIP_LOSFieldCreate(dest_id_t id, struct coord chunk_coord, struct tile_desc target,
                          const struct nav_private *priv, vec3_t map_pos, struct nav_unit_query_ctx *ctx,
                          struct LOS_field *out_los, const struct LOS_field *prev_los)
{
    if(!(s_hip.backend == N_HIP_BACKEND_HIP && s_hip.ctx)) {
        (N_LOSFieldCreate)(id, chunk_coord, target, priv, map_pos, ctx, out_los, prev_los);
        return;
    }
    struct n_hip_los_pending p;
    memset(&p, 0, offsetof(struct n_hip_los_pending, prev));
    p.id = id; p.chunk = chunk_coord; p.target = target; p.priv = (struct nav_private*)priv; p.map_pos = map_pos;
    p.base = -1;
    p.has_prev = prev_los != NULL;
    if(prev_los) {
        p.prev_chunk = prev_los->chunk;
        /* deferred mode: is the predecessor a field of this batch that has not been built yet? */
        if(s_hip.deferred) {
            for(int k = s_hip_los.npend - 1; k >= 0; k--) {
                const struct n_hip_los_pending *q = &s_hip_los.pend[k];
                if(q->id == id && q->chunk.r == prev_los->chunk.r && q->chunk.c == prev_los->chunk.c) { p.base = k; break; }
            }
        }
    }
    if(s_hip.deferred) {
        if(s_hip_los.npend == s_hip_los.cappend) {
            s_hip_los.cappend = s_hip_los.cappend ? s_hip_los.cappend * 2 : 64;
            s_hip_los.pend = realloc(s_hip_los.pend, sizeof(*s_hip_los.pend) * s_hip_los.cappend);
        }
        struct n_hip_los_pending *slot = &s_hip_los.pend[s_hip_los.npend++];
        memcpy(slot, &p, offsetof(struct n_hip_los_pending, prev));
        if(prev_los && p.base < 0)
            n_hip_los_to_bytes(prev_los, slot->prev);
        /* the placeholder the planner puts into the cache: the right chunk, nothing visible yet */
        memset(out_los, 0, sizeof(*out_los));
        out_los->chunk = chunk_coord;
        return;
    }
    navhip_los_req req;
    n_hip_make_los_req(&p, &req);
    static uint8_t prev_b[FIELD_RES_R * FIELD_RES_C], out_b[FIELD_RES_R * FIELD_RES_C];
    if(prev_los)
        n_hip_los_to_bytes(prev_los, prev_b);
    if(navhip_build_los(s_hip.ctx, &req, 1, prev_los ? prev_b : NULL, out_b, map_pos.x, map_pos.z) != NAVHIP_OK) {
        (N_LOSFieldCreate)(id, chunk_coord, target, priv, map_pos, ctx, out_los, prev_los);
        return;
    }
    s_hip_los.n_builds++; s_hip_los.n_batches++;
    n_hip_bytes_to_los(out_b, chunk_coord, out_los);
}

/* deferred LOS fields: one navhip_build_los call per chain level, results into the cache */
static bool n_hip_flush_los(void)
{
    const int n = s_hip_los.npend;
    if(n == 0)
        return true;
    bool ok = true;
    int *level = malloc(sizeof(int) * n), maxl = 0;
    for(int i = 0; i < n; i++) {
        level[i] = s_hip_los.pend[i].base < 0 ? 0 : level[s_hip_los.pend[i].base] + 1;
        if(level[i] > maxl) maxl = level[i];
    }
    navhip_los_req *reqs = malloc(sizeof(navhip_los_req) * n);
    uint8_t *prev = malloc((size_t)n * FIELD_RES_R * FIELD_RES_C), *out = malloc((size_t)n * FIELD_RES_R * FIELD_RES_C);
    int *who = malloc(sizeof(int) * n);
    for(int l = 0; l <= maxl; l++) {
        int m = 0;
        for(int i = 0; i < n; i++) {
            if(level[i] != l) continue;
            struct n_hip_los_pending *p = &s_hip_los.pend[i];
            n_hip_make_los_req(p, &reqs[m]);
            memcpy(prev + (size_t)m * sizeof(p->prev), p->base >= 0 ? s_hip_los.pend[p->base].out : p->prev, sizeof(p->prev));
            who[m++] = i;
        }
        if(m == 0) continue;
        const vec3_t mp = s_hip_los.pend[who[0]].map_pos;
        if(navhip_build_los(s_hip.ctx, reqs, m, prev, out, mp.x, mp.z) == NAVHIP_OK) {
            s_hip_los.n_builds += m; s_hip_los.n_batches++;
            for(int k = 0; k < m; k++)
                memcpy(s_hip_los.pend[who[k]].out, out + (size_t)k * sizeof(s_hip_los.pend[0].out), sizeof(s_hip_los.pend[0].out));
        }else{
            ok = false;
            for(int k = 0; k < m; k++) {               /* the CPU builder, on the predecessor's bytes */
                struct n_hip_los_pending *p = &s_hip_los.pend[who[k]];
                struct LOS_field lf, pl;
                if(p->has_prev)
                    n_hip_bytes_to_los(p->base >= 0 ? s_hip_los.pend[p->base].out : p->prev, p->prev_chunk, &pl);
                (N_LOSFieldCreate)(p->id, p->chunk, p->target, p->priv, p->map_pos, p->priv->unit_query_ctx,
                    &lf, p->has_prev ? &pl : NULL);
                n_hip_los_to_bytes(&lf, p->out);
            }
        }
    }
    for(int i = 0; i < n; i++) {
        struct n_hip_los_pending *p = &s_hip_los.pend[i];
        struct LOS_field lf;
        n_hip_bytes_to_los(p->out, p->chunk, &lf);
        N_FC_PutLOSField(p->priv->fieldcache, p->id, p->chunk, &lf);
    }
    free(level); free(reqs); free(prev); free(out); free(who);
    s_hip_los.npend = 0;
    return ok;
}

/* ---- dynamic obstacles (N_BlockersIncref / N_BlockersDecref, nav.c:4663,4685) ----------------- */

static struct{
    navhip_circle *circ;
    int            n, cap;
    vec3_t         map_pos;
    long           n_flushed, n_batches;
}s_hip_blk;

/* One more statement in N_BlockersIncref (ref_delta = +1) and N_BlockersDecref (-1): the host planes are
 * updated as before (the planner, the portal states and N_Update read them); the device copy follows
 * from the same call arguments at the next flush. */
/* How often the blockers planes have been written so far.  Answers that READ them -- N_ClosestPathable goes through
 * n_tile_blocked (nav.c:235: chunk->blockers > 0), n_closest_island_tiles is called with ignore_blockers = false
 * (nav.c:4725) -- may only be kept between ticks under this number (move_hip.c's per-flock queries).  Bumped by every
 * N_BlockersIncref / N_BlockersDecref through the recorder below; the OBB variants (nav.c:4685, 4696) and any other
 * writer of the planes call N_HIP_BlockersTouched(). */
static uint32_t s_hip_blk_generation = 1;
void     N_HIP_BlockersTouched(void) { if(++s_hip_blk_generation == 0) s_hip_blk_generation = 1; }
uint32_t N_HIP_BlockersGeneration(void) { return s_hip_blk_generation; }

void N_HIP_BlockersRecord(vec2_t xz_pos, float range, int faction_id, uint32_t flags, vec3_t map_pos, int ref_delta)
{
    N_HIP_BlockersTouched();
    if(!s_hip.ctx)
        return;
    if(s_hip_blk.n == s_hip_blk.cap) {
        s_hip_blk.cap = s_hip_blk.cap ? s_hip_blk.cap * 2 : 256;
        s_hip_blk.circ = realloc(s_hip_blk.circ, sizeof(navhip_circle) * s_hip_blk.cap);
    }
    s_hip_blk.circ[s_hip_blk.n++] = (navhip_circle){
        .x = xz_pos.x, .z = xz_pos.z, .radius = range, .faction_id = faction_id,
        .flags = (flags & ENTITY_FLAG_AIR) ? NAVHIP_ENTITY_FLAG_AIR : 0, .delta = ref_delta};
    s_hip_blk.map_pos = map_pos;
}

/* Once per tick, after N_Update (nav.c:2119) and before the tick's field builds: the recorded circles in
 * one device call.  The device rebuilds passability, flags the chunks whose passability changed
 * (navhip_changed_chunks: a subset of the chunks N_Update invalidates, nav.c:2143-2154) and relabels
 * their local islands.  false: the caller re-uploads the planes (N_HIP_SyncLayer). */
bool N_HIP_BlockersFlush(void)
{
    if(!s_hip.ctx || s_hip_blk.n == 0)
        return true;
    const int rc = navhip_blockers_circles(s_hip.ctx, s_hip_blk.circ, s_hip_blk.n, s_hip_blk.map_pos.x, s_hip_blk.map_pos.z);
    s_hip_blk.n_flushed += s_hip_blk.n; s_hip_blk.n_batches++;
    s_hip_blk.n = 0;
    return rc == NAVHIP_OK;
}

void N_HIP_BlockersStats(long out[2]) { out[0] = s_hip_blk.n_flushed; out[1] = s_hip_blk.n_batches; }

/* ---- the field cache's device image (fieldcache.c -> navhip_pool_*) --------------------------- */
/* The host keeps its field cache (the planner and the CPU fallbacks read it); with the pool enabled
 * every flow field the cache receives is ALSO resident on the device under the same N_FlowFieldID, and
 * every (dest, chunk) -> field mapping is mirrored into the pool's mapping table, so that the movement
 * tick can let the device sample the fields (navhip_world.n_field_slots = NAVHIP_POOL_RESIDENT,
 * vdes_xz = NaN) instead of calling N_DesiredPointSeekVelocity per agent on the host.
 *   N_FC_PutFlowField      -> N_HIP_FC_PutFlowField       (nav.c:1833,2008,2018,3533,3547,3632,3715,3964)
 *   N_FC_PutDestFFMapping  -> N_HIP_FC_PutDestFFMapping   (nav.c:1835,2021)
 *   N_FC_ClearAll          -> N_HIP_FC_ClearAll
 *   lru_flow_remove / LRU eviction (fieldcache.c:253,520,580; the on-evict argument of lru_flow_init :272)
 *                          -> navhip_pool_invalidate(ctx, key)
 * Mapping rows: the agent step uses the FLOCK INDEX as row; the movement tick announces the destination
 * of every flock (N_HIP_PoolSetRows) and the binding fans a destination's mappings out to its rows. */

bool N_HIP_PoolEnable(int n_slots, int n_rows)
{
    if(!s_hip.ctx || navhip_pool_create(s_hip.ctx, n_slots, n_rows) != NAVHIP_OK)
        return false;
    free(s_hip_pool.row_dest); free(s_hip_pool.row_used);
    s_hip_pool.row_dest = calloc(n_rows, sizeof(dest_id_t));
    s_hip_pool.row_used = calloc(n_rows, sizeof(bool));
    s_hip_pool.n_rows = n_rows;
    s_hip_pool.nm = 0;
    s_hip_pool.on = true;
    return true;
}

void N_HIP_PoolDisable(void)
{
    if(s_hip.ctx && s_hip_pool.on)
        navhip_pool_destroy(s_hip.ctx);
    free(s_hip_pool.row_dest); free(s_hip_pool.row_used);
    free(s_hip_pool.m_row); free(s_hip_pool.m_r); free(s_hip_pool.m_c); free(s_hip_pool.m_id);
    memset(&s_hip_pool, 0, sizeof(s_hip_pool));
}

bool N_HIP_PoolOn(void) { return s_hip_pool.on; }
void N_HIP_PoolStats(long out[3]) { out[0] = s_hip_pool.n_puts; out[1] = s_hip_pool.n_maps; out[2] = s_hip_pool.n_built; }

static void n_hip_pool_record(int row, struct coord chunk, ff_id_t ffid)
{
    if(s_hip_pool.nm == s_hip_pool.capm) {
        s_hip_pool.capm = s_hip_pool.capm ? s_hip_pool.capm * 2 : 1024;
        s_hip_pool.m_row = realloc(s_hip_pool.m_row, sizeof(int32_t) * s_hip_pool.capm);
        s_hip_pool.m_r = realloc(s_hip_pool.m_r, sizeof(uint16_t) * s_hip_pool.capm);
        s_hip_pool.m_c = realloc(s_hip_pool.m_c, sizeof(uint16_t) * s_hip_pool.capm);
        s_hip_pool.m_id = realloc(s_hip_pool.m_id, sizeof(uint64_t) * s_hip_pool.capm);
    }
    const int k = s_hip_pool.nm++;
    s_hip_pool.m_row[k] = row; s_hip_pool.m_r[k] = chunk.r; s_hip_pool.m_c[k] = chunk.c; s_hip_pool.m_id[k] = ffid;
}

void N_HIP_FC_PutFlowField(struct fieldcache_ctx *fc, ff_id_t ffid, const struct flow_field *ff)
{
    (N_FC_PutFlowField)(fc, ffid, ff);
    if(!s_hip_pool.on || ff->field[0][0].dir_idx == N_HIP_TAG)      /* (a placeholder of a deferred batch) */
        return;
    uint8_t dirs[FIELD_RES_R * FIELD_RES_C];
    n_hip_ff_to_dirs(ff, dirs);
    if(navhip_pool_put(s_hip.ctx, ffid, dirs) == NAVHIP_OK)
        s_hip_pool.n_puts++;
}

void N_HIP_FC_PutDestFFMapping(struct fieldcache_ctx *fc, dest_id_t dest_id, struct coord chunk, ff_id_t ffid)
{
    (N_FC_PutDestFFMapping)(fc, dest_id, chunk, ffid);
    if(!s_hip_pool.on)
        return;
    for(int row = 0; row < s_hip_pool.n_rows; row++)
        if(s_hip_pool.row_used[row] && s_hip_pool.row_dest[row] == dest_id)
            n_hip_pool_record(row, chunk, ffid);
}

void N_HIP_FC_ClearAll(struct fieldcache_ctx *fc)
{
    N_FC_ClearAll(fc);
    if(s_hip_pool.on) {
        navhip_pool_clear(s_hip.ctx);
        s_hip_pool.nm = 0;
        memset(s_hip_pool.row_used, 0, sizeof(bool) * s_hip_pool.n_rows);
    }
}

/* the destination of every flock, flock index = mapping row (movement.c: s_flocks[f].dest_id) */
void N_HIP_PoolSetRows(struct nav_private *priv, int n, const dest_id_t *dest_ids)
{
    if(!s_hip_pool.on)
        return;
    for(int row = 0; row < s_hip_pool.n_rows; row++) {
        const bool used = row < n;
        if(used == s_hip_pool.row_used[row] && (!used || s_hip_pool.row_dest[row] == dest_ids[row]))
            continue;
        /* the row changes hands: drop what it maps, then take over the host cache's mappings of the
         * new destination (N_FC_GetDestFFMapping, fieldcache.c:417) */
        for(int r = 0; r < (int)priv->height; r++)
        for(int c = 0; c < (int)priv->width; c++) {
            ff_id_t ffid = 0;                       /* id 0 is never resident: "no field" */
            if(used)
                N_FC_GetDestFFMapping(priv->fieldcache, dest_ids[row], (struct coord){r, c}, &ffid);
            if(used || s_hip_pool.row_used[row])
                n_hip_pool_record(row, (struct coord){r, c}, ffid);
        }
        s_hip_pool.row_used[row] = used;
        s_hip_pool.row_dest[row] = used ? dest_ids[row] : 0;
    }
}

/* the recorded mapping updates in ONE navhip_pool_map call (before the tick's agent step) */
bool N_HIP_PoolSync(void)
{
    if(!s_hip_pool.on || s_hip_pool.nm == 0)
        return true;
    const int rc = navhip_pool_map(s_hip.ctx, s_hip_pool.nm, s_hip_pool.m_row, s_hip_pool.m_r, s_hip_pool.m_c, s_hip_pool.m_id);
    s_hip_pool.n_maps += s_hip_pool.nm;
    s_hip_pool.nm = 0;
    return rc == NAVHIP_OK;
}

/* ---- the destination-only half of N_IsMaximallyClose (nav.c:4707-4742) ------------------------------------
 * The tiles of the destination's global island closest to it (n_closest_island_tiles, static in nav.c), as
 * absolute (row, column) nav tiles: what move_hip.c hands navhip_state_update once per flock, so that the
 * per-unit distance test of arrived() (movement.c:2170) runs on the device. */
int N_HIP_ClosestIslandTiles(struct nav_private *priv, enum nav_layer layer, vec3_t map_pos, vec2_t xz_dest,
                             int16_t *out_abs, int max_tiles)
{
    struct map_resolution res;
    N_GetResolution(priv, &res);
    struct tile_desc dest_td;
    if(!M_Tile_DescForPoint2D(res, map_pos, xz_dest, &dest_td))
        return 0;
    struct tile_desc tds[FIELD_RES_R * 2 + FIELD_RES_C * 2];
    const struct nav_chunk *chunk = &priv->chunks[layer][IDX(dest_td.chunk_r, priv->width, dest_td.chunk_c)];
    const uint16_t giid = chunk->islands[dest_td.tile_r][dest_td.tile_c];
    int ntds = n_closest_island_tiles(priv, layer, dest_td, giid, false, tds, ARR_SIZE(tds));
    if(ntds > max_tiles)
        ntds = max_tiles;
    for(int i = 0; i < ntds; i++) {
        out_abs[2 * i + 0] = (int16_t)(tds[i].chunk_r * FIELD_RES_R + tds[i].tile_r);
        out_abs[2 * i + 1] = (int16_t)(tds[i].chunk_c * FIELD_RES_C + tds[i].tile_c);
    }
    return ntds;
}
