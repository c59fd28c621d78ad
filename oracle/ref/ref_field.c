/* oracle/ref/ref_field.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Pulls the reference's src/navigation/field.c into this translation unit
 * (by #include, from where it lies under /root/reference/src; nothing is
 * copied) so that the harness can (a) call N_FlowFieldInit/N_FlowFieldUpdate
 * and (b) replay N_FlowFieldUpdate's own sequence of *static* helpers
 * (field.c:2055-2077) to capture the float integration field, which the
 * reference never exposes.
 */
#include "navigation/field.c"

/* the field.c half of the reference-side binding (game-side frontier extraction of the region builders) */
#include "field_hip.c"

#include "pfref.h"
#include "ref_internal.h"

#include <pthread.h>
#include <time.h>

static const struct portal *find_portal(const struct nav_chunk *chunk,
                                        int r0, int c0, int r1, int c1)
{
    for(size_t i = 0; i < chunk->num_portals; i++) {
        const struct portal *p = &chunk->portals[i];
        if(p->endpoints[0].r == r0 && p->endpoints[0].c == c0
        && p->endpoints[1].r == r1 && p->endpoints[1].c == c1)
            return p;
    }
    return NULL;
}

bool pfref_make_target(const struct nav_private *priv, const pfref_field_req *req,
                       struct field_target *out)
{
    memset(out, 0, sizeof(*out));
    if(req->type == TARGET_TILE) {
        out->type = TARGET_TILE;
        out->tile = (struct coord){req->tile_r, req->tile_c};
        return true;
    }
    if(req->type != TARGET_PORTAL)
        return false;

    const struct nav_chunk *chunk =
        &priv->chunks[req->layer][IDX(req->chunk_r, priv->width, req->chunk_c)];
    const struct portal *port = find_portal(chunk, req->port_r0, req->port_c0,
                                            req->port_r1, req->port_c1);
    if(!port)
        return false;
    const struct portal *next = n_portal(priv, req->layer, port->connected);
    if(!next)
        return false;
    if(next->chunk.r != req->next_chunk_r || next->chunk.c != req->next_chunk_c
    || next->endpoints[0].r != req->next_r0 || next->endpoints[0].c != req->next_c0
    || next->endpoints[1].r != req->next_r1 || next->endpoints[1].c != req->next_c1)
        return false;

    out->type = TARGET_PORTAL;
    out->pd = (struct portal_desc){
        .port = port, .port_iid = (uint16_t)req->port_iid,
        .next = next, .next_iid = (uint16_t)req->next_iid,
    };
    return true;
}

/* A request whose portal endpoints are not a portal of the reference's own portal build (synthetic
 * request streams cut portals their own way): field.c only reads the endpoints and the chunk of the
 * two portals (field_portal_initial_frontier :1160, field_fixup_portal_edges :830), so two
 * free-standing struct portal carry them.  storage: 2 portals that outlive the build. */
static bool make_target_synth(const struct nav_private *priv, const pfref_field_req *req,
                              struct field_target *out, struct portal *storage)
{
    if(pfref_make_target(priv, req, out))
        return true;
    if(req->type != TARGET_PORTAL)
        return false;
    memset(storage, 0, 2 * sizeof(struct portal));
    /* field_fixup_portal_edges (:830) takes the direction from the chunk of n_portal(port->connected):
     * any real portal of the next chunk serves (two chunks that share a passable edge have one) */
    int next_idx = IDX(req->next_chunk_r, priv->width, req->next_chunk_c);
    if(priv->chunks[req->layer][next_idx].num_portals == 0)
        return false;
    storage[0].connected = portal_ref_make(next_idx, 0);
    storage[1].connected = portal_ref_make(IDX(req->chunk_r, priv->width, req->chunk_c), 0);
    storage[0].chunk = (struct coord){req->chunk_r, req->chunk_c};
    storage[0].endpoints[0] = (struct coord){req->port_r0, req->port_c0};
    storage[0].endpoints[1] = (struct coord){req->port_r1, req->port_c1};
    storage[1].chunk = (struct coord){req->next_chunk_r, req->next_chunk_c};
    storage[1].endpoints[0] = (struct coord){req->next_r0, req->next_c0};
    storage[1].endpoints[1] = (struct coord){req->next_r1, req->next_c1};
    memset(out, 0, sizeof(*out));
    out->type = TARGET_PORTAL;
    out->pd = (struct portal_desc){
        .port = &storage[0], .port_iid = (uint16_t)req->port_iid,
        .next = &storage[1], .next_iid = (uint16_t)req->next_iid,
    };
    return true;
}

/* N_FlowFieldID (field.c:1952) itself, for every target kind.  kind = the reference's field_target.type;
 * a/b/c/d: ENEMIES faction_id | ENTITY target uid | ZONE centre (abs_r, abs_c), radius */
uint64_t pfref_flow_field_id(pfref_nav *nav, const pfref_field_req *req)
{
    struct field_target target;
    struct portal storage[2];
    if(!make_target_synth(&nav->priv, req, &target, storage))
        return 0;
    return N_FlowFieldID((struct coord){req->chunk_r, req->chunk_c}, target, (enum nav_layer)req->layer);
}

uint64_t pfref_region_field_id(int kind, int layer, int chunk_r, int chunk_c, uint32_t a, int b, int c)
{
    struct field_target target;
    memset(&target, 0, sizeof(target));
    target.type = kind;
    if(kind == TARGET_ENEMIES) {
        target.enemies.faction_id = (int)a;
        target.enemies.chunk = (struct coord){chunk_r, chunk_c};
    }else if(kind == TARGET_ENTITY) {
        target.ent.target = a;
    }else if(kind == TARGET_ZONE) {
        target.zone.centre = (struct tile_desc){(int)a / FIELD_RES_R, b / FIELD_RES_C, (int)a % FIELD_RES_R, b % FIELD_RES_C};
        target.zone.radius = (uint16_t)c;
    }else{
        return 0;
    }
    return N_FlowFieldID((struct coord){chunk_r, chunk_c}, target, (enum nav_layer)layer);
}

void pfref_req_from_target(struct coord chunk, int faction_id, enum nav_layer layer,
                           const struct field_target *t, pfref_field_req *out)
{
    memset(out, 0, sizeof(*out));
    out->layer = layer;
    out->type = t->type;
    out->faction_id = faction_id;
    out->chunk_r = chunk.r;
    out->chunk_c = chunk.c;
    if(t->type == TARGET_TILE) {
        out->tile_r = t->tile.r;
        out->tile_c = t->tile.c;
    }else if(t->type == TARGET_PORTAL) {
        out->port_r0 = t->pd.port->endpoints[0].r; out->port_c0 = t->pd.port->endpoints[0].c;
        out->port_r1 = t->pd.port->endpoints[1].r; out->port_c1 = t->pd.port->endpoints[1].c;
        out->next_chunk_r = t->pd.next->chunk.r;   out->next_chunk_c = t->pd.next->chunk.c;
        out->next_r0 = t->pd.next->endpoints[0].r; out->next_c0 = t->pd.next->endpoints[0].c;
        out->next_r1 = t->pd.next->endpoints[1].r; out->next_c1 = t->pd.next->endpoints[1].c;
        out->port_iid = t->pd.port_iid;
        out->next_iid = t->pd.next_iid;
    }
}

void pfref_dirs_to_ff(const uint8_t *dirs, struct flow_field *ff)
{
    for(int r = 0; r < FIELD_RES_R; r++)
    for(int c = 0; c < FIELD_RES_C; c++)
        ff->field[r][c].dir_idx = dirs[r * FIELD_RES_C + c] & 0xf;
}

void pfref_ff_to_dirs(const struct flow_field *ff, uint8_t *dirs)
{
    for(int r = 0; r < FIELD_RES_R; r++)
    for(int c = 0; c < FIELD_RES_C; c++)
        dirs[r * FIELD_RES_C + c] = ff->field[r][c].dir_idx;
}

int pfref_field_update(pfref_nav *nav, const pfref_field_req *req,
                       uint8_t *inout_dirs, float *out_integ)
{
    const struct nav_private *priv = pfref_nav_private(nav);
    struct field_target target;
    if(!pfref_make_target(priv, req, &target))
        return -1;

    struct coord chunk_coord = {req->chunk_r, req->chunk_c};
    struct flow_field ff;
    memset(&ff, 0, sizeof(ff));
    if(req->inout) {
        pfref_dirs_to_ff(inout_dirs, &ff);
        ff.chunk = chunk_coord;
    }else{
        N_FlowFieldInit(chunk_coord, &ff);
    }
    N_FlowFieldUpdate(chunk_coord, priv, req->faction_id, req->layer, target,
                      priv->unit_query_ctx, &ff);
    pfref_ff_to_dirs(&ff, inout_dirs);

    if(out_integ) {
        /* replay of field.c:2055-2077 with the reference's own static helpers */
        const struct nav_chunk *chunk =
            &priv->chunks[req->layer][IDX(chunk_coord.r, priv->width, chunk_coord.c)];
        pq_coord_t frontier;
        pq_coord_init(&frontier);

        float (*integ)[FIELD_RES_C] = (float(*)[FIELD_RES_C])out_integ;
        for(int r = 0; r < FIELD_RES_R; r++)
        for(int c = 0; c < FIELD_RES_C; c++)
            integ[r][c] = INFINITY;

        static __thread struct coord init_frontier[FIELD_RES_R * FIELD_RES_C];
        size_t ninit = field_initial_frontier(req->layer, target, chunk, priv, false,
            req->faction_id, priv->unit_query_ctx, init_frontier, ARR_SIZE(init_frontier));
        for(size_t i = 0; i < ninit; i++) {
            struct coord curr = init_frontier[i];
            pq_coord_push(&frontier, 0.0f, curr);
            integ[curr.r][curr.c] = 0.0f;
        }
        field_build_integration(&frontier, chunk, req->faction_id, priv->unit_query_ctx, integ);
        pq_coord_destroy(&frontier);
    }
    return 0;
}

/* N_FlowFieldUpdateToNearestPathable (field.c:2247) on an existing field */
int pfref_field_nearest_pathable(pfref_nav *nav, int layer, int chunk_r, int chunk_c, int start_r,
                                 int start_c, int faction_id, uint8_t *inout_dirs)
{
    const struct nav_private *priv = pfref_nav_private(nav);
    struct flow_field ff;
    memset(&ff, 0, sizeof(ff));
    pfref_dirs_to_ff(inout_dirs, &ff);
    ff.chunk = (struct coord){chunk_r, chunk_c};
    N_FlowFieldUpdateToNearestPathable(priv, layer, (struct coord){chunk_r, chunk_c},
        (struct coord){start_r, start_c}, faction_id, priv->unit_query_ctx, &ff);
    pfref_ff_to_dirs(&ff, inout_dirs);
    return 0;
}

/* N_FlowFieldUpdateIslandToNearest (field.c:2307) on an existing field whose target is `req` */
int pfref_field_island_to_nearest(pfref_nav *nav, const pfref_field_req *req, int local_iid,
                                  uint8_t *inout_dirs)
{
    const struct nav_private *priv = pfref_nav_private(nav);
    struct field_target target;
    if(!pfref_make_target(priv, req, &target))
        return -1;
    struct flow_field ff;
    memset(&ff, 0, sizeof(ff));
    pfref_dirs_to_ff(inout_dirs, &ff);
    ff.chunk = (struct coord){req->chunk_r, req->chunk_c};
    ff.target = target;
    N_FlowFieldUpdateIslandToNearest((uint16_t)local_iid, priv, req->layer, req->faction_id,
        priv->unit_query_ctx, &ff);
    pfref_ff_to_dirs(&ff, inout_dirs);
    return 0;
}

/* N_LOSFieldCreate (field.c:2085).  prev / out: 4096 bytes, bit 0 visible, bit 1
 * wavefront_blocked; prev_dr/prev_dc: chunk offset of the previous field (0,0 = none). */
int pfref_los_field(pfref_nav *nav, int layer, int faction_id, int chunk_r, int chunk_c,
                    int tgt_chunk_r, int tgt_chunk_c, int tgt_tile_r, int tgt_tile_c,
                    int prev_dr, int prev_dc, const uint8_t *prev, uint8_t *out)
{
    const struct nav_private *priv = pfref_nav_private(nav);
    dest_id_t id = (dest_id_t)(((layer & 0xf) << 4) | (faction_id & 0xf));   /* nav.c:5052-5060 */
    struct LOS_field lf, pl;
    struct tile_desc target = {tgt_chunk_r, tgt_chunk_c, tgt_tile_r, tgt_tile_c};
    bool has_prev = prev_dr != 0 || prev_dc != 0;
    if(has_prev) {
        memset(&pl, 0, sizeof(pl));
        pl.chunk = (struct coord){chunk_r + prev_dr, chunk_c + prev_dc};
        for(int r = 0; r < FIELD_RES_R; r++)
        for(int c = 0; c < FIELD_RES_C; c++) {
            pl.field[r][c].visible = prev[r * FIELD_RES_C + c] & 1;
            pl.field[r][c].wavefront_blocked = (prev[r * FIELD_RES_C + c] >> 1) & 1;
        }
    }
    N_LOSFieldCreate(id, (struct coord){chunk_r, chunk_c}, target, priv, nav->map_pos,
                     priv->unit_query_ctx, &lf, has_prev ? &pl : NULL);
    for(int r = 0; r < FIELD_RES_R; r++)
    for(int c = 0; c < FIELD_RES_C; c++)
        out[r * FIELD_RES_C + c] = (uint8_t)(lf.field[r][c].visible | (lf.field[r][c].wavefront_blocked << 1));
    return 0;
}

/* N_CellArrivalFieldCreate (field.c:2445); blocked: n_blocked (abs_r, abs_c) pairs or NULL */
int pfref_cell_arrival_field(pfref_nav *nav, int dim, int layer, int enemies,
                             int tgt_abs_r, int tgt_abs_c, int cen_abs_r, int cen_abs_c,
                             const int16_t *blocked, int n_blocked, uint8_t *out)
{
    struct nav_private *priv = pfref_nav_private(nav);
    struct tile_desc target = {tgt_abs_r / FIELD_RES_R, tgt_abs_c / FIELD_RES_C, tgt_abs_r % FIELD_RES_R, tgt_abs_c % FIELD_RES_C};
    struct tile_desc center = {cen_abs_r / FIELD_RES_R, cen_abs_c / FIELD_RES_C, cen_abs_r % FIELD_RES_R, cen_abs_c % FIELD_RES_C};
    size_t ws = (sizeof(float) + sizeof(bool)) * dim * dim + 64;
    void *work = malloc(ws);
    struct tile_desc *tds = n_blocked ? malloc(sizeof(struct tile_desc) * n_blocked) : NULL;
    for(int i = 0; i < n_blocked; i++)
        tds[i] = (struct tile_desc){blocked[2 * i] / FIELD_RES_R, blocked[2 * i + 1] / FIELD_RES_C,
                                    blocked[2 * i] % FIELD_RES_R, blocked[2 * i + 1] % FIELD_RES_C};
    struct nav_cell_overlay ov = {tds, (size_t)n_blocked};
    N_CellArrivalFieldCreate(priv, dim, dim, layer, (uint16_t)enemies, target, center, out, work, ws,
                             n_blocked ? &ov : NULL);
    free(work);
    free(tds);
    return 0;
}

/* N_GroupArrivalFieldCreate (field.c:2525): targets = world-space XZ points */
int pfref_group_arrival_field(pfref_nav *nav, int dim, int layer, int enemies, const float *targets_xz,
                              int ntargets, float center_x, float center_z,
                              const int16_t *blocked, int n_blocked, uint8_t *out)
{
    struct nav_private *priv = pfref_nav_private(nav);
    size_t ws = (sizeof(float) + sizeof(bool)) * dim * dim + 64;
    void *work = malloc(ws);
    struct tile_desc *tds = n_blocked ? malloc(sizeof(struct tile_desc) * n_blocked) : NULL;
    for(int i = 0; i < n_blocked; i++)
        tds[i] = (struct tile_desc){blocked[2 * i] / FIELD_RES_R, blocked[2 * i + 1] / FIELD_RES_C,
                                    blocked[2 * i] % FIELD_RES_R, blocked[2 * i + 1] % FIELD_RES_C};
    struct nav_cell_overlay ov = {tds, (size_t)n_blocked};
    N_GroupArrivalFieldCreate(priv, dim, dim, layer, (uint16_t)enemies, nav->map_pos,
                              (const vec2_t*)targets_xz, ntargets, (vec2_t){center_x, center_z}, out,
                              work, ws, n_blocked ? &ov : NULL);
    free(work);
    free(tds);
    return 0;
}

/* N_FlowFieldUpdate with a TARGET_ZONE target (field_update_zone, field.c:1822) on an existing
 * field; also returns the zone's initial frontier (field_zone_initial_frontier :1682) as
 * (abs_r, abs_c) pairs together with the padded region's geometry, i.e. what the game side would
 * hand to a region-field builder. */
int pfref_zone_field(pfref_nav *nav, int layer, int chunk_r, int chunk_c, int cen_abs_r, int cen_abs_c,
                     int radius, uint8_t *inout_dirs, int16_t *out_seeds, int max_seeds,
                     int *out_geom /* base_abs_r, base_abs_c, dim, roff, coff */)
{
    const struct nav_private *priv = pfref_nav_private(nav);
    struct zone_desc zone = {{cen_abs_r / FIELD_RES_R, cen_abs_c / FIELD_RES_C, cen_abs_r % FIELD_RES_R,
                              cen_abs_c % FIELD_RES_C}, (uint16_t)radius};
    struct field_target target;
    memset(&target, 0, sizeof(target));
    target.type = TARGET_ZONE;
    target.zone = zone;
    struct flow_field ff;
    memset(&ff, 0, sizeof(ff));
    pfref_dirs_to_ff(inout_dirs, &ff);
    ff.chunk = (struct coord){chunk_r, chunk_c};
    N_FlowFieldUpdate(ff.chunk, priv, FACTION_ID_NONE, layer, target, priv->unit_query_ctx, &ff);
    pfref_ff_to_dirs(&ff, inout_dirs);

    /* the same geometry field_update_zone uses (:1835-1849,1880-1881) */
    const int rdim = (priv->height > 1) ? FIELD_RES_R * 2 : FIELD_RES_R;
    const int cdim = (priv->width  > 1) ? FIELD_RES_C * 2 : FIELD_RES_C;
    struct tile_desc base = {
        (chunk_r > 0) ? chunk_r - 1 : chunk_r, (chunk_c > 0) ? chunk_c - 1 : chunk_c,
        (chunk_r > 0) ? FIELD_RES_R / 2 : 0, (chunk_c > 0) ? FIELD_RES_C / 2 : 0};
    out_geom[0] = base.chunk_r * FIELD_RES_R + base.tile_r;
    out_geom[1] = base.chunk_c * FIELD_RES_C + base.tile_c;
    out_geom[2] = rdim; out_geom[3] = (chunk_r > 0) ? FIELD_RES_R / 2 : 0;
    out_geom[4] = (chunk_c > 0) ? FIELD_RES_C / 2 : 0;
    out_geom[5] = cdim;
    size_t budget = (size_t)(M_PI * radius * radius + 0.5);
    if(budget > (size_t)(rdim * cdim)) budget = rdim * cdim;
    struct tile_desc *init = malloc(sizeof(struct tile_desc) * rdim * cdim);
    size_t ninit = field_zone_initial_frontier(&zone, priv, base, rdim, cdim, layer, init, budget);
    int n = 0;
    for(size_t i = 0; i < ninit && n < max_seeds; i++, n++) {
        out_seeds[2 * n] = (int16_t)(init[i].chunk_r * FIELD_RES_R + init[i].tile_r);
        out_seeds[2 * n + 1] = (int16_t)(init[i].chunk_c * FIELD_RES_C + init[i].tile_c);
    }
    free(init);
    return n;
}

struct bench_arg{
    const struct nav_private *priv;
    const pfref_field_req    *reqs;
    struct field_target      *targets;
    int                       begin, end, reps;
    unsigned                  sink;
    uint8_t                  *out_dirs;     /* optional: n * 4096 dir bytes (full-size parity runs) */
};

static void *bench_thread(void *p)
{
    struct bench_arg *a = p;
    unsigned sink = 0;
    for(int rep = 0; rep < a->reps; rep++) {
        for(int i = a->begin; i < a->end; i++) {
            const pfref_field_req *req = &a->reqs[i];
            struct coord chunk_coord = {req->chunk_r, req->chunk_c};
            struct flow_field ff;
            N_FlowFieldInit(chunk_coord, &ff);
            N_FlowFieldUpdate(chunk_coord, a->priv, req->faction_id, req->layer, a->targets[i],
                              a->priv->unit_query_ctx, &ff);
            sink += ff.field[rep & 63][i & 63].dir_idx;
            if(a->out_dirs)
                pfref_ff_to_dirs(&ff, a->out_dirs + (size_t)i * FIELD_RES_R * FIELD_RES_C);
        }
    }
    a->sink = sink;
    return NULL;
}

static double field_many(pfref_nav *nav, const pfref_field_req *reqs, int n, int reps,
                         int nthreads, uint8_t *out_dirs);

double pfref_field_bench(pfref_nav *nav, const pfref_field_req *reqs, int n, int reps,
                         int nthreads)
{
    return field_many(nav, reqs, n, reps, nthreads, NULL);
}

/* N_FlowFieldInit + N_FlowFieldUpdate for n independent requests on nthreads pthreads, results
 * kept: out_dirs[n][4096] */
double pfref_field_update_many(pfref_nav *nav, const pfref_field_req *reqs, int n, int nthreads,
                               uint8_t *out_dirs)
{
    return field_many(nav, reqs, n, 1, nthreads, out_dirs);
}

static double field_many(pfref_nav *nav, const pfref_field_req *reqs, int n, int reps,
                         int nthreads, uint8_t *out_dirs)
{
    const struct nav_private *priv = pfref_nav_private(nav);
    struct field_target *targets = malloc(sizeof(struct field_target) * (size_t)n);
    struct portal *synth = malloc(sizeof(struct portal) * 2 * (size_t)n);
    for(int i = 0; i < n; i++) {
        if(!make_target_synth(priv, &reqs[i], &targets[i], &synth[2 * (size_t)i])) {
            free(targets); free(synth);
            return -1.0;
        }
    }
    if(nthreads < 1) nthreads = 1;
    if(nthreads > 256) nthreads = 256;
    pthread_t tids[256];
    struct bench_arg args[256];

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for(int t = 0; t < nthreads; t++) {
        args[t] = (struct bench_arg){priv, reqs, targets,
            (int)((long)n * t / nthreads), (int)((long)n * (t + 1) / nthreads), reps, 0, out_dirs};
        pthread_create(&tids[t], NULL, bench_thread, &args[t]);
    }
    for(int t = 0; t < nthreads; t++)
        pthread_join(tids[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);

    free(targets); free(synth);
    return (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
}
