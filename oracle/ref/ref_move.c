/* oracle/ref/ref_move.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Pulls the reference's src/game/movement.c into this translation unit (by #include from
 * /root/reference/src; nothing is copied) so the harness can drive its *static* per-agent
 * velocity pipeline: move_velocity_work (movement.c:3395) -> point_seek_vpref (:1870) ->
 * arrive/cohesion/separation forces (:1546,:1653,:1690), nullify_impass_components (:1831),
 * find_neighbours (:2768) -> G_ClearPath_NewVelocity (clearpath.c:694) -> vec2_truncate.
 *
 * The engine services movement.c reads its snapshot through (flags / radius / faction table
 * getters, the M_Nav* pass-through wrappers of map.c:787-815, Entity_NavLayerWithRadius) are
 * given the minimal bodies below.  The arrival module is the reference's own arrival.c; its
 * per-unit / per-flock state is inactive unless a test sets it (pfref_move_set_arrival).
 */
#include "game/movement.c"

#include "pfref.h"
#include "ref_internal.h"

#include <pthread.h>
#include <time.h>

static float *s_ent_rot;    /* [n][4] what Entity_GetRot answers (pfref_move_set_turning), or NULL: STATE_TURNING units are not driven */
quat_t Entity_GetRot(uint32_t uid)
{
    if(!s_ent_rot)
        return (quat_t){0.0f, 0.0f, 0.0f, 1.0f};        /* (no transform table loaded: nobody is turned) */
    return (quat_t){s_ent_rot[4 * uid], s_ent_rot[4 * uid + 1], s_ent_rot[4 * uid + 2], s_ent_rot[4 * uid + 3]};
}
static bool s_aux_set;     /* pfref_move_set_state_aux gave the formation flags: the drivers below leave fstate.fid alone */

/* ---- engine services movement.c links against ------------------------------------------- */

uint32_t G_FlagsGetFrom(khash_t(id) *table, uint32_t uid)
{
    khiter_t k = kh_get(id, table, uid);
    assert(k != kh_end(table));
    return (uint32_t)kh_value(table, k);
}

float G_GetSelectionRadiusFrom(khash_t(range) *table, uint32_t uid)
{
    khiter_t k = kh_get(range, table, uid);
    assert(k != kh_end(table));
    return kh_value(table, k);
}

int G_GetFactionIDFrom(khash_t(id) *table, uint32_t uid)
{
    khiter_t k = kh_get(id, table, uid);
    assert(k != kh_end(table));
    return kh_value(table, k);
}

/* entity.c:554-575 */
int Entity_NavLayerWithRadius(uint32_t flags, float radius)
{
    bool water = !!(flags & ENTITY_FLAG_WATER);
    bool air = !!(flags & ENTITY_FLAG_AIR);
    int base = water ? NAV_LAYER_WATER_1X1 : air ? NAV_LAYER_AIR_1X1 : NAV_LAYER_GROUND_1X1;
    if(radius >= 15.0f) return base + 3;
    if(radius >= 10.0f) return base + 2;
    if(radius >= 5.0f)  return base + 1;
    return base;
}

/* map.c:787-815 pass-through wrappers: `struct map*` is the harness's pfref_nav here */
bool M_NavPositionPathable(const struct map *map, enum nav_layer layer, vec2_t xz_pos)
{
    pfref_nav *nav = (pfref_nav*)map;
    return N_PositionPathable(xz_pos, layer, &nav->priv, nav->map_pos);
}

bool M_NavPositionBlocked(const struct map *map, enum nav_layer layer, vec2_t xz_pos)
{
    pfref_nav *nav = (pfref_nav*)map;
    return N_PositionBlocked(xz_pos, layer, &nav->priv, nav->map_pos);
}

vec2_t M_NavDesiredPointSeekVelocity(const struct map *map, dest_id_t id, vec2_t curr_pos,
                                     vec2_t xz_dest)
{
    pfref_nav *nav = (pfref_nav*)map;
    return N_DesiredPointSeekVelocity(id, curr_pos, xz_dest, &nav->priv, nav->map_pos);
}

bool M_NavIsAdjacentToImpassable(const struct map *map, enum nav_layer layer, vec2_t xz_pos)
{
    pfref_nav *nav = (pfref_nav*)map;
    return N_IsAdjacentToImpassable(&nav->priv, layer, nav->map_pos, xz_pos);
}

bool M_NavIsMaximallyClose(const struct map *map, enum nav_layer layer, vec2_t xz_pos, vec2_t xz_dest, float tolerance)
{
    pfref_nav *nav = (pfref_nav*)map;
    return N_IsMaximallyClose(&nav->priv, layer, nav->map_pos, xz_pos, xz_dest, tolerance);
}

bool M_NavClosestPathable(const struct map *map, enum nav_layer layer, vec2_t xz_src, vec2_t *out)
{
    pfref_nav *nav = (pfref_nav*)map;
    return N_ClosestPathable(&nav->priv, layer, nav->map_pos, xz_src, out);
}

bool M_NavSegmentWithinRegion(const struct map *map, vec2_t a, vec2_t b, const uint64_t *keys, size_t num)
{
    pfref_nav *nav = (pfref_nav*)map;
    return N_SegmentWithinRegion(&nav->priv, nav->map_pos, a, b, keys, num);
}

size_t M_NavTileKeysForPositions(const struct map *map, const vec2_t *positions, size_t n, uint64_t *out)
{
    pfref_nav *nav = (pfref_nav*)map;
    return N_TileKeysForPositions(&nav->priv, nav->map_pos, positions, n, out);
}

void M_NavGetResolution(const struct map *map, struct map_resolution *out)
{
    pfref_nav *nav = (pfref_nav*)map;
    N_GetResolution(&nav->priv, out);
}

float M_HeightAtPoint(const struct map *map, vec2_t xz) { (void)map; (void)xz; return 0.0f; }

/* ---- world loading ---------------------------------------------------------------------- */

static struct {
    bool        loaded;
    int         n;
    pfref_nav  *nav;
    khash_t(id)    *flags;
    khash_t(pos)   *positions;
    khash_t(range) *radiuses;
    khash_t(id)    *factions;
    bg_ent_t       *postree;
    vec_cp_ent_t   *vecs;       /* 2 per agent */
    uint8_t        *los;
    float          *speed;
} s_w;

static bool uid_eq(const uint32_t *a, const uint32_t *b) { return *a == *b; }

void pfref_move_unload(void)
{
    if(!s_w.loaded)
        return;
    s_aux_set = false;
    free(s_ent_rot); s_ent_rot = NULL;
    kh_destroy(id, s_w.flags);
    kh_destroy(pos, s_w.positions);
    kh_destroy(range, s_w.radiuses);
    kh_destroy(id, s_w.factions);
    bg_ent_destroy(s_w.postree);
    free(s_w.postree);
    for(int i = 0; i < 2 * s_w.n; i++)
        vec_cp_ent_destroy(&s_w.vecs[i]);
    free(s_w.vecs);
    free(s_w.los);
    free(s_w.speed);
    free(s_move_work.in);
    free(s_move_work.out);
    for(int i = 0; i < vec_size(&s_flocks); i++) {
        kh_destroy(entity, vec_AT(&s_flocks, i).ents);
        for(int l = 0; l < NAV_LAYER_MAX; l++)
            free(vec_AT(&s_flocks, i).arrival.layers[l]);
    }
    vec_flock_destroy(&s_flocks);
    kh_destroy(state, s_entity_state_table);
    memset(&s_w, 0, sizeof(s_w));
}

static void move_hip_attrs_changed(void);      /* move_hip.c */

int pfref_move_load(pfref_nav *nav, const pfref_move_world *w)
{
    pfref_move_unload();
    move_hip_attrs_changed();                  /* new entities, new flocks */
    int n = w->n, ret;
    s_w.n = n;
    s_w.nav = nav;
    s_w.flags = kh_init(id);
    s_w.positions = kh_init(pos);
    s_w.radiuses = kh_init(range);
    s_w.factions = kh_init(id);
    s_entity_state_table = kh_init(state);
    vec_flock_init(&s_flocks);

    for(int f = 0; f < w->n_flocks; f++) {
        struct flock fl;
        memset(&fl, 0, sizeof(fl));
        fl.ents = kh_init(entity);
        fl.target_xz = (vec2_t){w->flock_target_xz[2 * f], w->flock_target_xz[2 * f + 1]};
        fl.dest_id = w->flock_dest_id[f];
        vec_flock_push(&s_flocks, fl);
    }

    /* position.c:276-283: the grid spans the whole map, centred on the map centre */
    float half_x = nav->priv.width  * TILES_PER_CHUNK_WIDTH  * X_COORDS_PER_TILE / 2.0f;
    float half_z = nav->priv.height * TILES_PER_CHUNK_HEIGHT * Z_COORDS_PER_TILE / 2.0f;
    float cx = nav->map_pos.x - half_x, cz = nav->map_pos.z + half_z;
    s_w.postree = malloc(sizeof(bg_ent_t));
    bg_ent_init(s_w.postree, cx - half_x, cx + half_x, cz - half_z, cz + half_z, uid_eq);
    bg_ent_reserve(s_w.postree, n > 0 ? n : 1);

    for(int i = 0; i < n; i++) {
        khiter_t k;
        k = kh_put(id, s_w.flags, i, &ret);        kh_value(s_w.flags, k) = (int)w->flags[i];
        k = kh_put(range, s_w.radiuses, i, &ret);  kh_value(s_w.radiuses, k) = w->radius[i];
        k = kh_put(id, s_w.factions, i, &ret);     kh_value(s_w.factions, k) = 0;
        k = kh_put(pos, s_w.positions, i, &ret);
        kh_value(s_w.positions, k) = (vec3_t){w->pos_xz[2 * i], 0.0f, w->pos_xz[2 * i + 1]};

        struct movestate ms;
        memset(&ms, 0, sizeof(ms));
        ms.state = w->state[i];
        ms.max_speed = w->max_speed[i];
        ms.velocity = (vec2_t){w->vel_xz[2 * i], w->vel_xz[2 * i + 1]};
        ms.prev_pos = ms.next_pos = kh_value(s_w.positions, k);
        k = kh_put(state, s_entity_state_table, i, &ret);
        kh_value(s_entity_state_table, k) = ms;

        if(w->flock[i] >= 0) {
            struct flock *fl = &vec_AT(&s_flocks, w->flock[i]);
            kh_put(entity, fl->ents, i, &ret);
        }
        bg_ent_insert(s_w.postree, w->pos_xz[2 * i], w->pos_xz[2 * i + 1], (uint32_t)i);
    }
    bg_ent_cleanup(s_w.postree);     /* G_Pos_CopyBitmapGrid, position.c:359-371 */

    memset(&s_move_work, 0, sizeof(s_move_work));
    s_move_work.gamestate.flags = s_w.flags;
    s_move_work.gamestate.positions = s_w.positions;
    s_move_work.gamestate.postree = s_w.postree;
    s_move_work.gamestate.sel_radiuses = s_w.radiuses;
    s_move_work.gamestate.faction_ids = s_w.factions;
    s_move_work.gamestate.map = (struct map*)nav;
    s_map = (const struct map*)nav;
    s_move_work.hz = (w->hz == 20) ? MOVE_HZ_20 : (w->hz == 10) ? MOVE_HZ_10
                   : (w->hz == 5) ? MOVE_HZ_5 : MOVE_HZ_1;
    s_move_work.in = calloc(n > 0 ? n : 1, sizeof(struct move_work_in));
    s_move_work.out = calloc(n > 0 ? n : 1, sizeof(struct move_work_out));
    s_move_work.nwork = n;

    s_w.vecs = calloc(2 * (n > 0 ? n : 1), sizeof(vec_cp_ent_t));
    s_w.los = malloc(n > 0 ? n : 1);
    s_w.speed = malloc(sizeof(float) * (n > 0 ? n : 1));
    for(int i = 0; i < n; i++) {
        vec_cp_ent_init(&s_w.vecs[2 * i]);
        vec_cp_ent_init(&s_w.vecs[2 * i + 1]);
        vec_cp_ent_resize(&s_w.vecs[2 * i], MAX_NEIGHBOURS);
        vec_cp_ent_resize(&s_w.vecs[2 * i + 1], MAX_NEIGHBOURS);
        s_w.los[i] = w->has_dest_los[i];
        s_w.speed[i] = w->speed[i];
        /* movement.c:4350-4376 */
        s_move_work.in[i] = (struct move_work_in){
            .ent_uid = i,
            .speed = w->speed[i],
            .cp_ent = (struct cp_ent){
                .xz_pos = (vec2_t){w->pos_xz[2 * i], w->pos_xz[2 * i + 1]},
                .xz_vel = (vec2_t){w->vel_xz[2 * i], w->vel_xz[2 * i + 1]},
                .radius = w->radius[i]},
            .save_debug = false,
            .dyn_neighbs = &s_w.vecs[2 * i],
            .stat_neighbs = &s_w.vecs[2 * i + 1],
            .has_dest_los = w->has_dest_los[i],
        };
    }
    s_w.loaded = true;
    return 0;
}

/* struct formation_state / cell_pos of every work item (movement.c:4377-4400 fills them from the
 * formation module; here they are explicit inputs) */
void pfref_move_set_formation(const uint8_t *ready, const float *cell_pos, const float *cohesion,
                              const float *align, const float *drag)
{
    for(int i = 0; i < s_w.n; i++) {
        struct move_work_in *in = &s_move_work.in[i];
        in->fstate.assignment_ready = ready[i];
        in->cell_pos = (vec2_t){cell_pos[2 * i], cell_pos[2 * i + 1]};
        in->fstate.normal_cohesion_force = (vec2_t){cohesion[2 * i], cohesion[2 * i + 1]};
        in->fstate.normal_align_force = (vec2_t){align[2 * i], align[2 * i + 1]};
        in->fstate.normal_drag_force = (vec2_t){drag[2 * i], drag[2 * i + 1]};
    }
}

/* Per-unit fine-arrival state (struct arrival_unit_state, arrival.h:105) and the flocks' per-layer
 * struct arrival_state, as explicit inputs: flags bit 0 = the unit is committed to a valid slot
 * (substate SEEK, sink_valid), bit 1 = its flock's arrival_state for its nav layer exists and is
 * filling.  G_Arrival_SeekTarget / G_Arrival_NeighbourSettling are the reference's own. */
void pfref_move_set_arrival(const float *sink_xz, const uint8_t *flags)
{
    for(int i = 0; i < s_w.n; i++) {
        struct movestate *ms = movestate_get(i);
        memset(&ms->arrival, 0, sizeof(ms->arrival));
        ms->arrival.substate = (flags[i] & 1) ? ARRIVAL_SUBSTATE_SEEK : ARRIVAL_SUBSTATE_APPROACH;
        ms->arrival.sink_valid = (flags[i] & 1) != 0;
        ms->arrival.sink = (vec2_t){sink_xz[2 * i], sink_xz[2 * i + 1]};
        struct flock *fl = flock_for_ent(i);
        if(!fl || !(flags[i] & 2))
            continue;
        float radius = G_GetSelectionRadiusFrom(s_move_work.gamestate.sel_radiuses, i);
        uint32_t eflags = G_FlagsGetFrom(s_move_work.gamestate.flags, i);
        int layer = Entity_NavLayerWithRadius(eflags, radius);
        if(!fl->arrival.layers[layer]) {
            fl->arrival.layers[layer] = calloc(1, sizeof(struct arrival_state));
            fl->arrival.layers[layer]->layer = layer;
        }
        fl->arrival.layers[layer]->phase = ARRIVAL_PHASE_FILLING;
    }
}

static void set_vdes(const float *vdes, int i)
{
    struct move_work_in *in = &s_move_work.in[i];
    if(vdes) {
        in->ent_des_v = (vec2_t){vdes[2 * i], vdes[2 * i + 1]};
    }else{
        const struct movestate *ms = movestate_get(i);
        const struct flock *fl = flock_for_ent(i);
        (void)ms;
        in->ent_des_v = fl ? M_NavDesiredPointSeekVelocity(s_move_work.gamestate.map, fl->dest_id,
                                 in->cp_ent.xz_pos, fl->target_xz)
                           : (vec2_t){0.0f, 0.0f};
    }
    s_move_work.out[i].ent_des_v = in->ent_des_v;          /* (compute_desired_velocity, movement.c:4175) */
    in->dyn_neighbs->size = 0;
    in->stat_neighbs->size = 0;
}

void pfref_move_velocity(const float *vdes, int begin, int end, float *out_vel)
{
    for(int i = begin; i < end; i++)
        set_vdes(vdes, i);
    if(end > begin)
        move_velocity_work(begin, end - 1);
    for(int i = begin; i < end; i++) {
        out_vel[2 * i]     = s_move_work.out[i].ent_vel.x;
        out_vel[2 * i + 1] = s_move_work.out[i].ent_vel.z;
    }
}

void pfref_move_vpref(int uid, const float vdes[2], float out[2])
{
    const struct flock *fl = flock_for_ent(uid);
    vec2_t v = point_seek_vpref(uid, fl, (vec2_t){vdes[0], vdes[1]}, s_w.los[uid], s_w.speed[uid]);
    out[0] = v.x; out[1] = v.z;
}

void pfref_move_forces(int uid, const float vdes[2], float out_arrive[2], float out_cohesion[2],
                       float out_separation[2])
{
    const struct flock *fl = flock_for_ent(uid);
    vec2_t a = arrive_force_point(uid, fl->target_xz, (vec2_t){vdes[0], vdes[1]}, s_w.los[uid]);
    vec2_t c = cohesion_force(uid, fl);
    vec2_t s = separation_force(uid, SEPARATION_BUFFER_DIST);
    out_arrive[0] = a.x; out_arrive[1] = a.z;
    out_cohesion[0] = c.x; out_cohesion[1] = c.z;
    out_separation[0] = s.x; out_separation[1] = s.z;
}

int pfref_move_neighbours(int uid, float *out_dyn, int *n_dyn, float *out_stat, int *n_stat)
{
    struct move_work_in *in = &s_move_work.in[uid];
    in->dyn_neighbs->size = 0;
    in->stat_neighbs->size = 0;
    find_neighbours(uid, in->dyn_neighbs, in->stat_neighbs);
    *n_dyn = vec_size(in->dyn_neighbs);
    *n_stat = vec_size(in->stat_neighbs);
    for(int i = 0; i < *n_dyn; i++)
        memcpy(out_dyn + 5 * i, &vec_AT(in->dyn_neighbs, i), 5 * sizeof(float));
    for(int i = 0; i < *n_stat; i++)
        memcpy(out_stat + 5 * i, &vec_AT(in->stat_neighbs, i), 5 * sizeof(float));
    return 0;
}

/* the WORK_TYPE_HIP arm (what a maintainer adds to movement.c) */
vec3_t              move_hip_map_pos(const struct map *map)     { return ((pfref_nav*)map)->map_pos; }
struct nav_private *move_hip_nav_private(const struct map *map) { return &((pfref_nav*)map)->priv; }
#include "move_hip.c"

void pfref_move_hip_sampling(int on)    { move_hip_set_device_sampling(on != 0); }
void pfref_move_hip_dry_run(int on)     { move_hip_set_dry_run(on != 0); }   /* host side only, no device (timing) */

/* ---- the fork-join the binding's range loops run on (move_hip_set_parallel_for) -----------------------------
 * In the engine this is move_submit_cpu_work's fan-out over worker tasks (movement.c:3751-3783, Sched_Create +
 * Sched_WaitOnFuture); the harness has no scheduler, so a pool of pthreads stands in: T - 1 workers asleep on a
 * condition variable between calls (no spinning: on the virtual machines these boxes are, spinning workers kept
 * the shares on one core), the caller takes a share itself. */
static struct {
    int             nthreads;                  /* workers + the caller */
    pthread_t       th[63];
    pthread_mutex_t mu;
    pthread_cond_t  go, done;
    hip_range_fn    fn; void *arg; int n;
    volatile unsigned gen;
    volatile int    pending;
    bool            started;
} s_pool = { .mu = PTHREAD_MUTEX_INITIALIZER, .go = PTHREAD_COND_INITIALIZER, .done = PTHREAD_COND_INITIALIZER };

static void pool_share(int t)
{
    const int T = s_pool.nthreads, n = s_pool.n;
    const int b = (int)((long)n * t / T), e = (int)((long)n * (t + 1) / T);
    if(e > b) s_pool.fn(b, e, s_pool.arg);
}

static void *pool_worker(void *arg)
{
    const int t = (int)(intptr_t)arg;
    unsigned seen = 0;
    for(;;) {
        if(__atomic_load_n(&s_pool.gen, __ATOMIC_ACQUIRE) == seen) {
            pthread_mutex_lock(&s_pool.mu);
            while(s_pool.gen == seen)
                pthread_cond_wait(&s_pool.go, &s_pool.mu);
            pthread_mutex_unlock(&s_pool.mu);
        }
        seen = __atomic_load_n(&s_pool.gen, __ATOMIC_ACQUIRE);
        pool_share(t);
        if(__atomic_sub_fetch(&s_pool.pending, 1, __ATOMIC_ACQ_REL) == 0) {
            pthread_mutex_lock(&s_pool.mu);
            pthread_cond_signal(&s_pool.done);
            pthread_mutex_unlock(&s_pool.mu);
        }
    }
    return NULL;
}

static void pool_parallel_for(hip_range_fn fn, int n, void *arg)
{
    if(s_pool.nthreads <= 1) { fn(0, n, arg); return; }
    s_pool.fn = fn; s_pool.arg = arg; s_pool.n = n;
    __atomic_store_n(&s_pool.pending, s_pool.nthreads - 1, __ATOMIC_RELEASE);
    pthread_mutex_lock(&s_pool.mu);
    __atomic_add_fetch(&s_pool.gen, 1, __ATOMIC_RELEASE);
    pthread_cond_broadcast(&s_pool.go);
    pthread_mutex_unlock(&s_pool.mu);
    pool_share(s_pool.nthreads - 1);
    if(__atomic_load_n(&s_pool.pending, __ATOMIC_ACQUIRE) != 0) {
        pthread_mutex_lock(&s_pool.mu);
        while(__atomic_load_n(&s_pool.pending, __ATOMIC_ACQUIRE) != 0)
            pthread_cond_wait(&s_pool.done, &s_pool.mu);
        pthread_mutex_unlock(&s_pool.mu);
    }
}

/* threads the binding's host loops fork over (1 = the calling task only).  The workers are created once and live
 * for the process (a later call with another count keeps the pool's); loops shorter than min_items (0: 8192)
 * stay serial. */
void pfref_move_hip_threads(int nthreads, int min_items)
{
    if(nthreads > 64) nthreads = 64;
    if(nthreads < 1) nthreads = 1;
    if(!s_pool.started && nthreads > 1) {
        s_pool.nthreads = nthreads;
        for(int t = 0; t < nthreads - 1; t++)
            pthread_create(&s_pool.th[t], NULL, pool_worker, (void*)(intptr_t)t);
        s_pool.started = true;
    }
    /* (1 = serial again, whatever the pool's size; any other count uses the pool as it was started: its shares are
     * computed from its own nthreads) */
    const bool serial = nthreads == 1;
    move_hip_set_parallel_for((!serial && s_pool.started) ? pool_parallel_for : NULL, min_items);
}
void pfref_move_hip_stats(long out[3])  { move_hip_stats(out); }

/* like pfref_move_velocity, through move_hip_velocity_work; returns 0 when the device arm declined */
int pfref_move_velocity_hip(const float *vdes, int begin, int end, float *out_vel)
{
    static const float zero2[2] = {0.0f, 0.0f};
    for(int i = begin; i < end; i++) {
        /* with device sampling on, compute_desired_velocity's per-agent host sampling is what the arm
         * replaces: the work items arrive without a desired velocity */
        if(!vdes && s_hip_sample_on_device) set_vdes(zero2 - 2 * i, i);
        else                                set_vdes(vdes, i);
    }
    if(end <= begin)
        return 1;
    if(!move_hip_velocity_work(begin, end - 1))
        return 0;
    for(int i = begin; i < end; i++) {
        out_vel[2 * i]     = s_move_work.out[i].ent_vel.x;
        out_vel[2 * i + 1] = s_move_work.out[i].ent_vel.z;
    }
    return 1;
}

/* The snapshot the binding hands to the library (hip_snap_fill + hip_snap_world), copied out for inspection: the host
 * logic of move_hip.c -- the arena, the cached bucket positions, the cached flock tables, the fork over worker
 * threads -- can be checked without a device.  Arrays are in dense (ascending uid) order; returns the entity count,
 * or -1 when a buffer is too small. */
int pfref_move_hip_snapshot(int cap, float *pos, float *vel, float *radius, float *max_speed, uint32_t *flags,
                            uint8_t *state, int32_t *flock, int32_t *flock_offsets, int32_t *flock_members, int32_t *n_flocks)
{
    struct hip_snap S;
    hip_snap_fill(&S);
    if(S.n > cap) return -1;
    memcpy(pos, S.pos, sizeof(float) * 2 * S.n); memcpy(vel, S.vel, sizeof(float) * 2 * S.n);
    memcpy(radius, S.radius, sizeof(float) * S.n); memcpy(max_speed, S.max_speed, sizeof(float) * S.n);
    memcpy(flags, S.flags, sizeof(uint32_t) * S.n); memcpy(state, S.state, S.n);
    memcpy(flock, S.flock, sizeof(int32_t) * S.n);
    memcpy(flock_offsets, S.flock_offsets, sizeof(int32_t) * (S.nflocks + 1));
    memcpy(flock_members, S.flock_members, sizeof(int32_t) * S.flock_offsets[S.nflocks]);
    *n_flocks = (int32_t)S.nflocks;
    hip_snap_free(&S);
    return S.n;
}

static float *s_next_rot_in;           /* movestate.next_rot as an input of the state updates (pfref_move_set_next_rot) */
static double s_hip_state_work_s;      /* wall time of the last move_hip_state_work (the device half of the state pass) */
double pfref_move_hip_state_work_seconds(void) { return s_hip_state_work_s; }
void pfref_move_hip_state_times(double out[6]) { move_hip_state_times(out); }

/* pfref_move_state_update through the binding: move_hip_state_work (ONE navhip_state_update for the slab)
 * then move_hip_update_work per unit.  dev_flags[i] = what the device answered (NAVHIP_SU_*).
 * Returns 0 when the device arm declined. */
int pfref_move_state_update_hip(const float *new_vel, const float *vdes, int begin, int end, uint8_t *out_state,
                                uint8_t *out_flags, uint8_t *dev_flags)
{
    for(int i = begin; i < end; i++) {
        struct move_work_in *in = &s_move_work.in[i];
        struct move_work_out *out = &s_move_work.out[i];
        struct movestate *ms = movestate_get(i);
        if(!s_aux_set) in->fstate.fid = NULL_FID;
        out->ent_uid = i;
        out->ent_vel = (vec2_t){new_vel[2 * i], new_vel[2 * i + 1]};
        out->ent_des_v = (vec2_t){vdes[2 * i], vdes[2 * i + 1]};
        if(s_next_rot_in)
            ms->next_rot = (quat_t){s_next_rot_in[4 * i], s_next_rot_in[4 * i + 1], s_next_rot_in[4 * i + 2], s_next_rot_in[4 * i + 3]};
        else if(PFM_Vec2_Len(&out->ent_vel) > EPSILON)
            ms->next_rot = dir_quat_from_velocity(intended_heading(out->ent_des_v, out->ent_vel));
        memset(&out->patch, 0, sizeof(out->patch));
    }
    if(end <= begin)
        return 1;
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    const bool worked = move_hip_state_work(begin, end - 1);
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    s_hip_state_work_s = (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec);
    if(!worked)
        return 0;
    for(int i = begin; i < end; i++) {
        struct move_work_out *out = &s_move_work.out[i];
        struct movestate *ms = movestate_get(i);
        dev_flags[i] = s_hip_su_flags[i];
        if(ms->state == STATE_TURNING && !s_ent_rot
        && !(G_FlagsGetFrom(s_move_work.gamestate.flags, i) & ENTITY_FLAG_GARRISONED)) {
            out_state[i] = (uint8_t)ms->state; out_flags[i] = 0;       /* (as in pfref_move_state_update) */
            continue;
        }
        move_hip_update_work(i, i);
        const bool set = (out->patch.flags & UPDATE_SET_STATE) != 0, mov = (out->patch.flags & UPDATE_SET_MOVING) != 0;
        out_state[i] = (uint8_t)((set || mov) ? out->patch.next_state : ms->state);
        out_flags[i] = (uint8_t)((set ? 1 : 0) | ((set && out->patch.next_block) ? 2 : 0) | (mov ? 4 : 0)
                                 | ((out->patch.flags & UPDATE_SET_TARGET_DIR) ? 8 : 0) | ((out->patch.flags & UPDATE_SET_DEST) ? 16 : 0));
    }
    return 1;
}
void pfref_move_hip_state_stats(long out[3]) { move_hip_state_stats(out); }
void pfref_move_hip_settle_stats(long out[4]) { move_hip_settle_stats(out); }
long pfref_move_hip_wait_differ(void) { return move_hip_wait_differ(); }
long pfref_move_hip_surround_differ(void) { return move_hip_surround_differ(); }
void pfref_move_hip_resident_state_pass(int on) { move_hip_set_resident_state_pass(on != 0); }
long pfref_move_hip_resident_passes(void) { return move_hip_resident_passes(); }
/* the velocities / desired directions the last velocity pass left in the work items (s_move_work.out) */
void pfref_move_get_out(float *vel, float *vdes)
{
    for(int i = 0; i < s_w.n; i++) {
        vel[2 * i] = s_move_work.out[i].ent_vel.x; vel[2 * i + 1] = s_move_work.out[i].ent_vel.z;
        vdes[2 * i] = s_move_work.out[i].ent_des_v.x; vdes[2 * i + 1] = s_move_work.out[i].ent_des_v.z;
    }
}

struct mbench_arg{ int begin, end, reps; };

static void *mbench_thread(void *p)
{
    struct mbench_arg *a = p;
    for(int r = 0; r < a->reps; r++) {
        for(int i = a->begin; i < a->end; i++) {
            s_move_work.in[i].dyn_neighbs->size = 0;
            s_move_work.in[i].stat_neighbs->size = 0;
        }
        if(a->end > a->begin)
            move_velocity_work(a->begin, a->end - 1);
    }
    return NULL;
}

double pfref_move_bench(const float *vdes, int begin, int end, int reps, int nthreads,
                        float *out_vel)
{
    for(int i = begin; i < end; i++)
        set_vdes(vdes, i);
    if(nthreads < 1) nthreads = 1;
    if(nthreads > 256) nthreads = 256;
    pthread_t tids[256];
    struct mbench_arg args[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int n = end - begin;
    /* same contiguous slab split as move_submit_cpu_work (movement.c:3756-3762) */
    for(int t = 0; t < nthreads; t++) {
        args[t] = (struct mbench_arg){begin + (int)((long)n * t / nthreads),
                                      begin + (int)((long)n * (t + 1) / nthreads), reps};
        pthread_create(&tids[t], NULL, mbench_thread, &args[t]);
    }
    for(int t = 0; t < nthreads; t++)
        pthread_join(tids[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if(out_vel) {
        for(int i = begin; i < end; i++) {
            out_vel[2 * i]     = s_move_work.out[i].ent_vel.x;
            out_vel[2 * i + 1] = s_move_work.out[i].ent_vel.z;
        }
    }
    return (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
}

/* pfref_move_bench through the binding's WORK_TYPE_HIP arm: `reps` calls of move_hip_velocity_work over the work
 * items [begin, end) -- what the engine's nav task sees per tick, host buffers and PCIe included.  Returns the
 * wall seconds (< 0: the arm declined); out_times = {fill, submit..wait, scatter back} seconds of those calls. */
double pfref_move_bench_hip(const float *vdes, int begin, int end, int reps, double out_times[4])
{
    for(int i = begin; i < end; i++)
        set_vdes(vdes, i);
    double dump[4];
    move_hip_times(dump, 1);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for(int r = 0; r < reps; r++)
        if(end <= begin || !move_hip_velocity_work(begin, end - 1))
            return -1.0;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    move_hip_times(out_times, 1);
    return (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
}

/* the ent_des_v each work item carried in the last pfref_move_velocity / _bench call */
void pfref_move_get_vdes(float *out_vdes)
{
    for(int i = 0; i < s_w.n; i++) {
        out_vdes[2 * i]     = s_move_work.in[i].ent_des_v.x;
        out_vdes[2 * i + 1] = s_move_work.in[i].ent_des_v.z;
    }
}

/* kh_foreach order of flock->ents: the order cohesion_force (movement.c:1660) sums in */
int pfref_move_flock_order(int flock, uint32_t *out_uids)
{
    struct flock *fl = &vec_AT(&s_flocks, flock);
    int n = 0;
    uint32_t curr;
    kh_foreach_key(fl->ents, curr, {
        out_uids[n++] = curr;
    });
    return n;
}


/* entity_compute_update (movement.c:2303) for the work items [begin, end): new_vel / vdes are the tick's
 * move_work_out.ent_vel / ent_des_v.  No formation (fstate.fid = NULL_FID); every unit's orientation is set
 * to its intended heading first, so that the heading gate (:2321-2334, host state) lets the velocity
 * through -- the gate itself is orientation bookkeeping the device pass takes as an input.
 * out_state[i] = patch.next_state when UPDATE_SET_STATE is set, else the current state;
 * out_flags[i] bit 0 = UPDATE_SET_STATE, bit 1 = next_block. */
void pfref_move_state_update(const float *new_vel, const float *vdes, int begin, int end, uint8_t *out_state,
                             uint8_t *out_flags)
{
    for(int i = begin; i < end; i++) {
        struct move_work_in *in = &s_move_work.in[i];
        struct move_work_out *out = &s_move_work.out[i];
        struct movestate *ms = movestate_get(i);
        if(ms->state == STATE_TURNING && !s_ent_rot
        && !(G_FlagsGetFrom(s_move_work.gamestate.flags, i) & ENTITY_FLAG_GARRISONED)) {
            /* (its arm reads the entity's transform, Entity_GetRot: not loaded; a garrisoned unit returns
             * before the state switch, :2344-2351) */
            out_state[i] = (uint8_t)ms->state; out_flags[i] = 0;
            continue;
        }
        if(!s_aux_set) in->fstate.fid = NULL_FID;
        out->ent_uid = i;
        out->ent_vel = (vec2_t){new_vel[2 * i], new_vel[2 * i + 1]};
        out->ent_des_v = (vec2_t){vdes[2 * i], vdes[2 * i + 1]};
        if(s_next_rot_in)
            ms->next_rot = (quat_t){s_next_rot_in[4 * i], s_next_rot_in[4 * i + 1], s_next_rot_in[4 * i + 2], s_next_rot_in[4 * i + 3]};
        else if(PFM_Vec2_Len(&out->ent_vel) > EPSILON)
            ms->next_rot = dir_quat_from_velocity(intended_heading(out->ent_des_v, out->ent_vel));
        memset(&out->patch, 0, sizeof(out->patch));
        entity_compute_update(s_move_work.hz, i, out->ent_vel, out->ent_des_v, in, &out->patch);
        const bool set = (out->patch.flags & UPDATE_SET_STATE) != 0, mov = (out->patch.flags & UPDATE_SET_MOVING) != 0;
        out_state[i] = (uint8_t)((set || mov) ? out->patch.next_state : ms->state);
        out_flags[i] = (uint8_t)((set ? 1 : 0) | ((set && out->patch.next_block) ? 2 : 0) | (mov ? 4 : 0)
                                 | ((out->patch.flags & UPDATE_SET_TARGET_DIR) ? 8 : 0) | ((out->patch.flags & UPDATE_SET_DEST) ? 16 : 0));
    }
}

/* ---- SURVEY 8(f4), the heading gate and the arrival overlay's settle rule --------------------------------- */

/* entity_compute_update (movement.c:2303) with movestate.next_rot as an INPUT, so that the heading gate
 * (:2319-2336) decides: out_turn[i] = UPDATE_TURNING_IN_PLACE of the patch (set for turn_to_move, :2405-2410;
 * a combat-held unit sets it regardless, :2399 -- the caller leaves those out), out_vel[i] = the patch's
 * next_velocity.  STATE_TURNING units are skipped as in pfref_move_state_update. */
void pfref_move_heading_gate(const float *new_vel, const float *vdes, const float *next_rot, int begin, int end,
                             uint8_t *out_turn, float *out_vel)
{
    for(int i = begin; i < end; i++) {
        struct move_work_in *in = &s_move_work.in[i];
        struct move_work_out *out = &s_move_work.out[i];
        struct movestate *ms = movestate_get(i);
        out_turn[i] = 0; out_vel[2 * i] = out_vel[2 * i + 1] = 0.0f;
        if(ms->state == STATE_TURNING && !s_ent_rot
        && !(G_FlagsGetFrom(s_move_work.gamestate.flags, i) & ENTITY_FLAG_GARRISONED))
            continue;
        if(!s_aux_set) in->fstate.fid = NULL_FID;
        out->ent_uid = i;
        out->ent_vel = (vec2_t){new_vel[2 * i], new_vel[2 * i + 1]};
        out->ent_des_v = (vec2_t){vdes[2 * i], vdes[2 * i + 1]};
        ms->next_rot = (quat_t){next_rot[4 * i], next_rot[4 * i + 1], next_rot[4 * i + 2], next_rot[4 * i + 3]};
        memset(&out->patch, 0, sizeof(out->patch));
        entity_compute_update(s_move_work.hz, i, out->ent_vel, out->ent_des_v, in, &out->patch);
        out_turn[i] = (out->patch.flags & UPDATE_TURNING_IN_PLACE) != 0;
        if(out->patch.flags & UPDATE_SET_VELOCITY) {
            out_vel[2 * i] = out->patch.next_velocity.x; out_vel[2 * i + 1] = out->patch.next_velocity.z;
        }
    }
}

/* the rotation dir_quat_from_velocity (movement.c:1411) gives a heading, for building test inputs */
void pfref_move_dir_quat(const float *heading_xz, int n, float *out_quat)
{
    for(int i = 0; i < n; i++) {
        quat_t q = dir_quat_from_velocity((vec2_t){heading_xz[2 * i], heading_xz[2 * i + 1]});
        out_quat[4 * i] = q.x; out_quat[4 * i + 1] = q.y; out_quat[4 * i + 2] = q.z; out_quat[4 * i + 3] = q.w;
    }
}

/* adjacent_settled_count (movement.c:982) on the loaded snapshot */
void pfref_move_settled_count(const int32_t *uids, int nq, int32_t *out)
{
    for(int q = 0; q < nq; q++)
        out[q] = adjacent_settled_count((uint32_t)uids[q]);
}

/* G_Arrival_ShouldSettle (arrival.c:946) for nq units against ONE zone given as plain arrays: the zone's
 * struct arrival_state is filled from them (region_keys = N_TileKeysForPositions of region_xz, nav.c:4303,
 * returned in out_keys; returns their number), each unit's struct arrival_unit_state from the per-unit arrays,
 * and written back after the call. */
int pfref_arrival_should_settle(pfref_nav *nav, int layer, const float *centre_xz, int radius, float unit_radius,
                                float fill_frac, int active_row, int num_rows,
                                const float *slots_xz, const int32_t *slot_ring, int num_slots,
                                const float *region_xz, int num_region_pos, uint64_t *out_keys,
                                int nq, const float *new_pos_xz, const float *vel_xz, const float *radius_of,
                                const int32_t *nsettled, uint8_t *substate, const uint8_t *sink_valid,
                                const float *sink_xz, const float *order_pos_xz, float *progress_anchor_xz,
                                uint8_t *progress_anchored, int32_t *stuck, uint8_t *out_settle)
{
    if(num_slots > ARRIVAL_MAX_SLOTS || num_region_pos > ARRIVAL_MAX_SLOTS)
        return -1;
    struct arrival_state *as = calloc(1, sizeof(*as));
    as->phase = ARRIVAL_PHASE_FILLING;
    as->layer = (enum nav_layer)layer;
    as->centre = (vec2_t){centre_xz[0], centre_xz[1]};
    as->radius = (uint16_t)radius;
    as->unit_radius = unit_radius;
    as->fill_frac = fill_frac;
    as->active_row = active_row;
    as->num_rows = num_rows;
    as->num_slots = num_slots;
    for(int i = 0; i < num_slots; i++) {
        as->slots[i] = (vec2_t){slots_xz[2 * i], slots_xz[2 * i + 1]};
        as->slot_ring[i] = slot_ring[i];
    }
    as->num_region = (int)N_TileKeysForPositions(&nav->priv, nav->map_pos, (const vec2_t*)region_xz,
                                                 (size_t)num_region_pos, as->region_keys);
    memcpy(out_keys, as->region_keys, sizeof(uint64_t) * as->num_region);
    for(int q = 0; q < nq; q++) {
        struct arrival_unit_state us;
        memset(&us, 0, sizeof(us));
        us.substate = (enum arrival_substate)substate[q];
        us.sink_valid = sink_valid[q] != 0;
        us.sink = (vec2_t){sink_xz[2 * q], sink_xz[2 * q + 1]};
        us.order_pos = (vec2_t){order_pos_xz[2 * q], order_pos_xz[2 * q + 1]};
        us.progress_anchor = (vec2_t){progress_anchor_xz[2 * q], progress_anchor_xz[2 * q + 1]};
        us.progress_anchored = progress_anchored[q] != 0;
        us.stuck = stuck[q];
        out_settle[q] = G_Arrival_ShouldSettle(as, &us, (const struct map*)nav, (const struct map*)nav,
            (vec2_t){new_pos_xz[2 * q], new_pos_xz[2 * q + 1]}, (vec2_t){vel_xz[2 * q], vel_xz[2 * q + 1]},
            radius_of[q], nsettled[q]) ? 1 : 0;
        substate[q] = (uint8_t)us.substate;
        progress_anchor_xz[2 * q] = us.progress_anchor.x; progress_anchor_xz[2 * q + 1] = us.progress_anchor.z;
        progress_anchored[q] = us.progress_anchored ? 1 : 0;
        stuck[q] = us.stuck;
    }
    const int nk = as->num_region;
    free(as);
    return nk;
}

/* A real arrival zone for flock `flock` and nav layer `layer` (struct arrival_state, arrival.h:66) from plain arrays,
 * as pfref_arrival_should_settle builds one: entity_compute_update then takes the G_Arrival_ShouldSettle arm
 * (:2443) for the flock's units of that layer. */
int pfref_move_set_arrival_zone(int flock, int layer, const float *centre_xz, int radius, float unit_radius,
                                float fill_frac, int active_row, int num_rows, const float *slots_xz,
                                const int32_t *slot_ring, int num_slots, const float *region_xz, int num_region_pos)
{
    if(flock < 0 || flock >= (int)vec_size(&s_flocks) || num_slots > ARRIVAL_MAX_SLOTS || num_region_pos > ARRIVAL_MAX_SLOTS)
        return -1;
    struct flock *fl = &vec_AT(&s_flocks, flock);
    if(!fl->arrival.layers[layer])
        fl->arrival.layers[layer] = calloc(1, sizeof(struct arrival_state));
    struct arrival_state *as = fl->arrival.layers[layer];
    memset(as, 0, sizeof(*as));
    as->phase = ARRIVAL_PHASE_FILLING;
    as->layer = (enum nav_layer)layer;
    as->centre = (vec2_t){centre_xz[0], centre_xz[1]};
    as->radius = (uint16_t)radius;
    as->unit_radius = unit_radius;
    as->fill_frac = fill_frac;
    as->active_row = active_row;
    as->num_rows = num_rows;
    as->num_slots = num_slots;
    for(int i = 0; i < num_slots; i++) {
        as->slots[i] = (vec2_t){slots_xz[2 * i], slots_xz[2 * i + 1]};
        as->slot_ring[i] = slot_ring[i];
    }
    as->num_region = (int)N_TileKeysForPositions(&s_w.nav->priv, s_w.nav->map_pos, (const vec2_t*)region_xz,
                                                 (size_t)num_region_pos, as->region_keys);
    return as->num_region;
}

/* every unit's struct arrival_unit_state (arrival.h:105) from arrays of n rows; get = read them back */
void pfref_move_set_arrival_units(const uint8_t *substate, const uint8_t *sink_valid, const float *sink_xz,
                                  const float *order_pos_xz, const float *progress_anchor_xz,
                                  const uint8_t *progress_anchored, const int32_t *stuck)
{
    for(int i = 0; i < s_w.n; i++) {
        struct arrival_unit_state *us = &movestate_get(i)->arrival;
        memset(us, 0, sizeof(*us));
        us->substate = (enum arrival_substate)substate[i];
        us->sink_valid = sink_valid[i] != 0;
        us->sink = (vec2_t){sink_xz[2 * i], sink_xz[2 * i + 1]};
        us->order_pos = (vec2_t){order_pos_xz[2 * i], order_pos_xz[2 * i + 1]};
        us->progress_anchor = (vec2_t){progress_anchor_xz[2 * i], progress_anchor_xz[2 * i + 1]};
        us->progress_anchored = progress_anchored[i] != 0;
        us->stuck = stuck[i];
    }
}

void pfref_move_get_arrival_units(uint8_t *substate, float *progress_anchor_xz, uint8_t *progress_anchored, int32_t *stuck)
{
    for(int i = 0; i < s_w.n; i++) {
        const struct arrival_unit_state *us = &movestate_get(i)->arrival;
        substate[i] = (uint8_t)us->substate;
        progress_anchor_xz[2 * i] = us->progress_anchor.x; progress_anchor_xz[2 * i + 1] = us->progress_anchor.z;
        progress_anchored[i] = us->progress_anchored ? 1 : 0;
        stuck[i] = us->stuck;
    }
}

/* the inputs of the flag / counter arms of the state switch (:2423-2437, :2630-2668): move_work_in.fstate as bits
 * (1 member, 2 assignment_ready, 4 assigned_to_cell, 8 in_range_of_cell, 16 arrived_at_cell), movestate.wait_ticks_left
 * and .wait_prev; pfref_move_get_wait_ticks reads the counters back */
void pfref_move_set_state_aux(const uint8_t *fstate, const int32_t *wait_ticks_left, const uint8_t *wait_prev)
{
    s_aux_set = true;
    for(int i = 0; i < s_w.n; i++) {
        struct formation_state *fs = &s_move_work.in[i].fstate;
        fs->fid = (fstate[i] & 1) ? 1 : NULL_FID;
        fs->assignment_ready = (fstate[i] & 2) != 0;
        fs->assigned_to_cell = (fstate[i] & 4) != 0;
        fs->in_range_of_cell = (fstate[i] & 8) != 0;
        fs->arrived_at_cell = (fstate[i] & 16) != 0;
        struct movestate *ms = movestate_get(i);
        ms->wait_ticks_left = wait_ticks_left[i];
        ms->wait_prev = (enum move_state)wait_prev[i];
    }
}

void pfref_move_get_wait_ticks(int32_t *out)
{
    for(int i = 0; i < s_w.n; i++)
        out[i] = movestate_get(i)->wait_ticks_left;
}

/* STATE_TURNING (:2606-2628): the entity transform's rotation (what Entity_GetRot answers) and movestate.target_dir
 * per unit; from then on pfref_move_state_update(_hip) drives TURNING units too */
void pfref_move_set_turning(const float *ent_rot, const float *target_dir)
{
    free(s_ent_rot);
    s_ent_rot = malloc(sizeof(float) * 4 * s_w.n);
    memcpy(s_ent_rot, ent_rot, sizeof(float) * 4 * s_w.n);
    for(int i = 0; i < s_w.n; i++)
        movestate_get(i)->target_dir = (quat_t){target_dir[4 * i], target_dir[4 * i + 1], target_dir[4 * i + 2], target_dir[4 * i + 3]};
}

/* ---- STATE_SURROUND_ENTITY (:2509-2567) and movement rates below 20 Hz (:2368-2377) ---------------------------
 * map.c's two unit-query wrappers for a MOVABLE target (nav_target_geom_from map.c:158, nav_obj_adjacent :181,
 * nav_closest_reachable_adjacent_pos :191): straight into nav.c, which this harness builds.  (A static target needs the
 * entity's OBB from the transform tables, which the harness does not load.)  The unit-query context of the tick IS the
 * tick's snapshot (movement.c fills it from the same tables). */
static void surround_target_geom(uint32_t target_uid, vec2_t *xz, float *radius)
{
    const struct move_gamestate *gs = &s_move_work.gamestate;
    if(!(G_FlagsGetFrom(gs->flags, target_uid) & ENTITY_FLAG_MOVABLE))
        pfref_stub_abort("M_Nav*AdjacentFrom: static target");
    *xz = G_Pos_GetXZFrom(gs->positions, target_uid);
    *radius = G_GetSelectionRadiusFrom(gs->sel_radiuses, target_uid);
}

bool M_NavObjAdjacentFrom(const struct map *map, uint32_t uid, uint32_t target_uid, const struct nav_unit_query_ctx *ctx)
{
    (void)ctx;
    pfref_nav *nav = (pfref_nav*)map;
    const struct move_gamestate *gs = &s_move_work.gamestate;
    vec2_t txz; float tr;
    surround_target_geom(target_uid, &txz, &tr);
    return N_ObjAdjacentToDynamicWith(&nav->priv, nav->map_pos, G_Pos_GetXZFrom(gs->positions, uid),
                                      G_GetSelectionRadiusFrom(gs->sel_radiuses, uid), txz, tr);
}

bool M_NavClosestReachableAdjacentPosFrom(const struct map *map, enum nav_layer layer, vec2_t xz_src, uint32_t target_uid,
                                          const struct nav_unit_query_ctx *ctx, vec2_t *out)
{
    (void)ctx;
    pfref_nav *nav = (pfref_nav*)map;
    vec2_t txz; float tr;
    surround_target_geom(target_uid, &txz, &tr);
    return N_ClosestReachableAdjacentPosDynamic(&nav->priv, layer, nav->map_pos, xz_src, txz, tr, out);
}

/* movestate.surround_target_uid (-1 = NULL_UID), .surround_target_prev, .surround_nearest_prev per unit */
void pfref_move_set_surround(const int32_t *target_uid, const float *target_prev_xz, const float *nearest_prev_xz)
{
    for(int i = 0; i < s_w.n; i++) {
        struct movestate *ms = movestate_get(i);
        ms->surround_target_uid = target_uid[i] < 0 ? NULL_UID : (uint32_t)target_uid[i];
        ms->surround_target_prev = (vec2_t){target_prev_xz[2 * i], target_prev_xz[2 * i + 1]};
        ms->surround_nearest_prev = (vec2_t){nearest_prev_xz[2 * i], nearest_prev_xz[2 * i + 1]};
    }
}

void pfref_move_get_surround(float *target_prev_xz, float *nearest_prev_xz, float *next_dest_xz)
{
    for(int i = 0; i < s_w.n; i++) {
        const struct movestate *ms = movestate_get(i);
        target_prev_xz[2 * i] = ms->surround_target_prev.x; target_prev_xz[2 * i + 1] = ms->surround_target_prev.z;
        nearest_prev_xz[2 * i] = ms->surround_nearest_prev.x; nearest_prev_xz[2 * i + 1] = ms->surround_nearest_prev.z;
        next_dest_xz[2 * i] = s_move_work.out[i].patch.next_dest.x; next_dest_xz[2 * i + 1] = s_move_work.out[i].patch.next_dest.z;
    }
}

/* What the host hands the device for the surround arm (navhip_state_aux_in.surround_query / .surround_dest_xz): the two
 * nav queries per STATE_SURROUND_ENTITY unit with a live target, from pos + new_vel ([0]) and from pos ([1]). */
void pfref_move_surround_queries(const float *new_vel, uint8_t *out_query, float *out_dest_xz)
{
    const struct move_gamestate *gs = &s_move_work.gamestate;
    for(int i = 0; i < s_w.n; i++) {
        const struct movestate *ms = movestate_get(i);
        out_query[i] = 0;
        memset(out_dest_xz + 4 * i, 0, sizeof(float) * 4);
        if(ms->state != STATE_SURROUND_ENTITY || ms->surround_target_uid == NULL_UID)
            continue;
        if(!entity_exists(ms->surround_target_uid) || M_NavObjAdjacentFrom(gs->map, i, ms->surround_target_uid, NULL)) {
            out_query[i] = 1;
            continue;
        }
        const enum nav_layer layer = Entity_NavLayerWithRadius(G_FlagsGetFrom(gs->flags, i), G_GetSelectionRadiusFrom(gs->sel_radiuses, i));
        const vec2_t pos = G_Pos_GetXZFrom(gs->positions, i);
        const vec2_t vel = {new_vel[2 * i], new_vel[2 * i + 1]};
        vec2_t from[2] = {{0}, pos}, dest;
        PFM_Vec2_Add((vec2_t*)&pos, (vec2_t*)&vel, &from[0]);
        for(int c = 0; c < 2; c++) {
            if(M_NavClosestReachableAdjacentPosFrom(gs->map, layer, from[c], ms->surround_target_uid, NULL, &dest)) {
                out_query[i] |= (uint8_t)(2 << c);
                out_dest_xz[4 * i + 2 * c] = dest.x; out_dest_xz[4 * i + 2 * c + 1] = dest.z;
            }
        }
    }
}

/* movestate.next_rot as an INPUT of pfref_move_state_update (NULL: facing on the heading, nobody is gated) */
void pfref_move_set_next_rot(const float *next_rot)
{
    free(s_next_rot_in);
    s_next_rot_in = NULL;
    if(next_rot) {
        s_next_rot_in = malloc(sizeof(float) * 4 * s_w.n);
        memcpy(s_next_rot_in, next_rot, sizeof(float) * 4 * s_w.n);
    }
}

/* movestate.next_pos (x, z; y = 0) and .step per unit: what a rate below 20 Hz interpolates from (:2372) */
void pfref_move_set_interp(const float *next_pos_xz, const float *step)
{
    for(int i = 0; i < s_w.n; i++) {
        struct movestate *ms = movestate_get(i);
        ms->next_pos = (vec3_t){next_pos_xz[2 * i], 0.0f, next_pos_xz[2 * i + 1]};
        ms->step = step[i];
    }
}

/* STATE_ENTER_ENTITY_RANGE inputs (:2569-2604): movestate.surround_target_uid (-1 = NULL_UID), .target_range,
 * .target_prev_pos per unit */
void pfref_move_set_range_targets(const int32_t *target_uid, const float *target_range, const float *target_prev_xz)
{
    for(int i = 0; i < s_w.n; i++) {
        struct movestate *ms = movestate_get(i);
        ms->surround_target_uid = target_uid[i] < 0 ? NULL_UID : (uint32_t)target_uid[i];
        ms->target_range = target_range[i];
        ms->target_prev_pos = (vec2_t){target_prev_xz[2 * i], target_prev_xz[2 * i + 1]};
    }
}
