/* bindings/permafrost/move_hip.c -- the WORK_TYPE_HIP arm of fork_join_velocity_computations.
 *
 * What INTEGRATION.md tells a maintainer to add to src/game/movement.c next to the WORK_TYPE_CPU /
 * WORK_TYPE_GPU arms (movement.c:315-318,3737,4182-4194): fill a navhip_world from the tick's
 * snapshot tables (struct move_gamestate, :296) and work items (struct move_work_in, :264), run
 * navhip_agent_step, copy the velocities into s_move_work.out[] like copy_gpu_results does
 * (:4248-4261).  It lives in movement.c's translation unit because those tables are static; in this
 * repository the test harness #include <time.h>
#includes it right after movement.c (oracle/ref/ref_move.c).
 *
 * With device sampling on (move_hip_set_device_sampling; needs the binding's resident pool,
 * N_HIP_PoolEnable in nav_hip.c) the per-agent N_DesiredPointSeekVelocity calls of
 * compute_desired_velocity (movement.c:4166) are skipped for point-seeking agents: the device samples the
 * flow fields of the resident pool itself.  Agents it cannot answer (no field mapped for their chunk,
 * FD_NONE under them: the planner / repair cases of nav.c:3483-3554) come back flagged and are stepped
 * by the host's own move_velocity_work after a host-side N_DesiredPointSeekVelocity.
 */
#include <navhip.h>
#include <math.h>

navhip_ctx *N_HIP_Ctx(void);          /* nav_hip.c */
bool N_HIP_PoolOn(void);
void N_HIP_PoolSetRows(struct nav_private *priv, int n, const dest_id_t *dest_ids);
bool N_HIP_PoolSync(void);
/* the map's position (M_GetPos, map.c) and its nav context (map->nav_private, map.c:787-815) */
vec3_t              move_hip_map_pos(const struct map *map);
struct nav_private *move_hip_nav_private(const struct map *map);

static bool s_hip_sample_on_device;
static long s_hip_stats[3];            /* agents sampled on the device, host fallbacks, steps */
void move_hip_set_device_sampling(bool on) { s_hip_sample_on_device = on; }
void move_hip_stats(long out[3])           { memcpy(out, s_hip_stats, sizeof(s_hip_stats)); }

/* The flock tables (membership, member lists, targets) only change when an entity is added or removed
 * or a flock is made, re-targeted or disbanded: movement.c bumps this epoch there (G_Move_AddEntity :4591,
 * G_Move_RemoveEntity :4615, make_flock :789, the flock disbanding :742,:2859) and the library keeps the
 * flock tables of an unchanged epoch on the device (navhip_world.static_epoch).  Radius, max speed and
 * flags have other writers (do_set_max_speed :3226, the selection-radius setter, ENTITY_FLAG_GARRISONED):
 * the library transfers them every tick. */
static uint32_t s_hip_attr_epoch = 1;
static void move_hip_attrs_changed(void)      /* (declared ahead in the harness: ref_move.c) */
{
    if(++s_hip_attr_epoch == 0)
        s_hip_attr_epoch = 1;
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

/* The tick's snapshot tables (struct move_gamestate, :296) as the dense arrays navhip_world points at:
 * every entity of the position snapshot in ascending uid order (the GL path densifies uids the same way,
 * ent_gpu_id_map :302), the per-entity movestate columns, and the flock tables. */
struct hip_snap {
    int          n;
    uint32_t    *uids;
    khash_t(id) *dense;                 /* uid -> dense index */
    float       *pos, *vel, *radius, *max_speed, *sink;
    uint32_t    *flags;
    uint8_t     *state, *arr_flags;
    int32_t     *flock;
    size_t       nflocks;
    float       *flock_target;
    int32_t     *flock_offsets, *flock_members;
    bool         any_arrival;
};
#define DENSE(S, uid) kh_value((S)->dense, kh_get(id, (S)->dense, (uid)))

/* The sorted uid list and the uid -> dense index map only change when the entity set does; sorting 100 000 uids
 * and building the map were two thirds of the 4.7 ms a tick spent filling the snapshot (bench.py `dropin`).  They
 * are kept between ticks and rebuilt when the set has another size or a cached uid is gone from the tick's
 * position table (an equal count with every cached uid present IS the same set). */
static struct { int n; uint32_t *uids; khash_t(id) *dense; } s_hip_set;

static void hip_set_rebuild(const struct move_gamestate *gs)
{
    const int n = (int)kh_size(gs->positions);
    free(s_hip_set.uids);
    if(s_hip_set.dense) kh_destroy(id, s_hip_set.dense);
    s_hip_set.n = n;
    s_hip_set.uids = malloc(sizeof(uint32_t) * (n > 0 ? n : 1));
    int k = 0;
    uint32_t key;
    kh_foreach_key(gs->positions, key, { s_hip_set.uids[k++] = key; });
    qsort(s_hip_set.uids, n, sizeof(uint32_t), cmp_u32);
    s_hip_set.dense = kh_init(id);
    kh_resize(id, s_hip_set.dense, n + n / 2);
    for(int i = 0; i < n; i++) {
        int ret;
        khiter_t it = kh_put(id, s_hip_set.dense, s_hip_set.uids[i], &ret);
        kh_value(s_hip_set.dense, it) = i;
    }
}

static void hip_snap_fill(struct hip_snap *S)
{
    const struct move_gamestate *gs = &s_move_work.gamestate;
    memset(S, 0, sizeof(*S));
    const int n = S->n = (int)kh_size(gs->positions);
    if(!s_hip_set.dense || s_hip_set.n != n)
        hip_set_rebuild(gs);
    else {
        for(int i = 0; i < n; i++)
            if(kh_get(pos, gs->positions, s_hip_set.uids[i]) == kh_end(gs->positions)) { hip_set_rebuild(gs); break; }
    }
    S->uids = s_hip_set.uids;
    S->dense = s_hip_set.dense;
    S->pos = calloc(2 * n + 2, sizeof(float)); S->vel = calloc(2 * n + 2, sizeof(float));
    S->radius = calloc(n + 1, sizeof(float)); S->max_speed = calloc(n + 1, sizeof(float));
    S->sink = calloc(2 * n + 2, sizeof(float));
    S->flags = calloc(n + 1, sizeof(uint32_t));
    S->state = calloc(n + 1, 1); S->arr_flags = calloc(n + 1, 1);
    S->flock = malloc(sizeof(int32_t) * (n + 1));
    S->nflocks = vec_size(&s_flocks);
    S->flock_target = calloc(2 * (S->nflocks ? S->nflocks : 1), sizeof(float));
    S->flock_offsets = calloc(S->nflocks + 1, sizeof(int32_t));
    S->flock_members = malloc(sizeof(int32_t) * (n > 0 ? n : 1));

    for(int i = 0; i < n; i++) {
        const uint32_t uid = S->uids[i];
        vec2_t p = G_Pos_GetXZFrom(gs->positions, uid);
        S->pos[2 * i] = p.x; S->pos[2 * i + 1] = p.z;
        S->flags[i] = G_FlagsGetFrom(gs->flags, uid);
        S->radius[i] = G_GetSelectionRadiusFrom(gs->sel_radiuses, uid);
        S->flock[i] = -1;
        S->state[i] = STATE_ARRIVED;                  /* no movestate: a still obstacle */
        khiter_t k = kh_get(state, s_entity_state_table, uid);
        if(k == kh_end(s_entity_state_table))
            continue;
        const struct movestate *ms = &kh_value(s_entity_state_table, k);
        S->state[i] = (uint8_t)ms->state;
        S->vel[2 * i] = ms->velocity.x; S->vel[2 * i + 1] = ms->velocity.z;
        S->max_speed[i] = ms->max_speed;
        /* struct arrival_unit_state: committed to a valid slot (unit_committed, arrival.c:90) */
        if((ms->arrival.substate == ARRIVAL_SUBSTATE_SEEK || ms->arrival.substate == ARRIVAL_SUBSTATE_SEEK_ARMED)
        && ms->arrival.sink_valid) {
            S->arr_flags[i] |= 1;
            S->any_arrival = true;
        }
        S->sink[2 * i] = ms->arrival.sink.x; S->sink[2 * i + 1] = ms->arrival.sink.z;
    }
    /* flocks: members in kh_foreach order of flock->ents (the order cohesion_force sums in, :1660) */
    int at = 0;
    for(size_t f = 0; f < S->nflocks; f++) {
        const struct flock *fl = &vec_AT(&s_flocks, f);
        S->flock_target[2 * f] = fl->target_xz.x; S->flock_target[2 * f + 1] = fl->target_xz.z;
        S->flock_offsets[f] = at;
        uint32_t curr;
        kh_foreach_key(fl->ents, curr, {
            const int i = DENSE(S, curr);
            S->flock_members[at++] = i;
            S->flock[i] = (int32_t)f;
            const struct arrival_state *as = G_ArrivalGroup_ForLayer(&fl->arrival,
                Entity_NavLayerWithRadius(S->flags[i], S->radius[i]));
            if(as && as->phase == ARRIVAL_PHASE_FILLING) { S->arr_flags[i] |= 2; S->any_arrival = true; }
        });
    }
    S->flock_offsets[S->nflocks] = at;
}

static void hip_snap_free(struct hip_snap *S)
{
    free(S->pos); free(S->vel); free(S->radius); free(S->max_speed); free(S->sink);
    free(S->flags); free(S->state); free(S->arr_flags); free(S->flock); free(S->flock_target);
    free(S->flock_offsets); free(S->flock_members);
}

/* the snapshot half of a navhip_world */
static void hip_snap_world(const struct hip_snap *S, navhip_world *W)
{
    const struct move_gamestate *gs = &s_move_work.gamestate;
    memset(W, 0, sizeof(*W));
    W->n_ents = S->n; W->n_flocks = (int32_t)S->nflocks; W->hz = hz_count(s_move_work.hz);
    W->pos_xz = S->pos; W->vel_xz = S->vel; W->radius = S->radius; W->max_speed = S->max_speed;
    W->flags = S->flags; W->state = S->state; W->flock = S->flock;
    W->flock_target_xz = S->flock_target; W->flock_offsets = S->flock_offsets; W->flock_members = S->flock_members;
    vec3_t map_pos = move_hip_map_pos(gs->map);
    W->map_pos_x = map_pos.x; W->map_pos_z = map_pos.z;
    /* bg_ent_init bounds of the position snapshot (position.c:276-283) */
    const struct nav_private *np = move_hip_nav_private(gs->map);
    float half_x = np->width * TILES_PER_CHUNK_WIDTH * X_COORDS_PER_TILE / 2.0f;
    float half_z = np->height * TILES_PER_CHUNK_HEIGHT * Z_COORDS_PER_TILE / 2.0f;
    float cx = map_pos.x - half_x, cz = map_pos.z + half_z;
    W->grid_xmin = cx - half_x; W->grid_xmax = cx + half_x; W->grid_zmin = cz - half_z; W->grid_zmax = cz + half_z;
    W->static_epoch = s_hip_attr_epoch;
}

/* move_velocity_work(begin_idx, end_idx) for the work items [begin_idx, end_idx] on the device.
 * Returns false when the library is not available (the caller runs the CPU arm). */
/* where a WORK_TYPE_HIP tick spends its time, for a host that wants to report it (bench.py's `dropin`): seconds
 * since the last reset in {filling the snapshot + work-item arrays, navhip_agent_step_submit .. _wait (staging,
 * PCIe both ways, the kernels), scattering the results back into s_move_work.out[]}, and the calls */
static double s_hip_times[4];
static double hip_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
void move_hip_times(double out[4], int reset)
{
    memcpy(out, s_hip_times, sizeof(s_hip_times));
    if(reset) memset(s_hip_times, 0, sizeof(s_hip_times));
}

static bool move_hip_velocity_work(int begin_idx, int end_idx)
{
    navhip_ctx *ctx = N_HIP_Ctx();
    if(!ctx || end_idx < begin_idx)
        return false;
    const double t_begin = hip_now();
    const struct move_gamestate *gs = &s_move_work.gamestate;
    struct hip_snap S;
    hip_snap_fill(&S);
    const int n = S.n;

    float *speed = calloc(n + 1, sizeof(float));
    uint8_t *los = calloc(n + 1, 1), *form_ready = calloc(n + 1, 1);
    float *vdes = calloc(2 * n + 2, sizeof(float)), *cell_pos = calloc(2 * n + 2, sizeof(float));
    float *f_coh = calloc(2 * n + 2, sizeof(float)), *f_align = calloc(2 * n + 2, sizeof(float));
    float *f_drag = calloc(2 * n + 2, sizeof(float));

    /* work items */
    int lo = n, hi = -1;
    bool any_form = false, ok_pool = true;
    for(int w = begin_idx; w <= end_idx; w++) {
        const struct move_work_in *in = &s_move_work.in[w];
        const int i = DENSE(&S, in->ent_uid);
        vdes[2 * i] = in->ent_des_v.x; vdes[2 * i + 1] = in->ent_des_v.z;
        speed[i] = in->speed;
        los[i] = in->has_dest_los;
        form_ready[i] = in->fstate.assignment_ready;
        cell_pos[2 * i] = in->cell_pos.x; cell_pos[2 * i + 1] = in->cell_pos.z;
        f_coh[2 * i] = in->fstate.normal_cohesion_force.x; f_coh[2 * i + 1] = in->fstate.normal_cohesion_force.z;
        f_align[2 * i] = in->fstate.normal_align_force.x; f_align[2 * i + 1] = in->fstate.normal_align_force.z;
        f_drag[2 * i] = in->fstate.normal_drag_force.x; f_drag[2 * i + 1] = in->fstate.normal_drag_force.z;
        any_form = any_form || S.state[i] == STATE_MOVING_IN_FORMATION || S.state[i] == STATE_ARRIVING_TO_CELL;
        if(i < lo) lo = i;
        if(i > hi) hi = i;
    }
    /* the device steps the contiguous uid slab [lo, hi]; entities inside it that carry no work item
     * (other slabs of a threaded split) are stepped too and their results dropped */
    const bool sample = s_hip_sample_on_device && N_HIP_PoolOn();
    if(sample) {
        /* flock index = mapping row of the resident pool: announce every flock's destination, flush the
         * mappings the planner recorded since the last tick, and leave the sampling of the point-seeking
         * agents (the default arm of ent_desired_velocity, :1510-1521) to the device: vdes.x = NaN */
        dest_id_t *fdest = malloc(sizeof(dest_id_t) * (S.nflocks ? S.nflocks : 1));
        for(size_t f = 0; f < S.nflocks; f++)
            fdest[f] = vec_AT(&s_flocks, f).dest_id;
        N_HIP_PoolSetRows(move_hip_nav_private(gs->map), (int)S.nflocks, fdest);
        free(fdest);
        if(!N_HIP_PoolSync()) {
            ok_pool = false;
        }else{
            for(int w = begin_idx; w <= end_idx; w++) {
                const int i = DENSE(&S, s_move_work.in[w].ent_uid);
                if(S.state[i] == STATE_MOVING && S.flock[i] >= 0 && !(S.arr_flags[i] & 2))
                    vdes[2 * i] = NAN;
            }
        }
    }
    navhip_world W;
    hip_snap_world(&S, &W);
    W.speed = speed; W.has_dest_los = los; W.vdes_xz = vdes;
    if(sample && ok_pool)
        W.n_field_slots = NAVHIP_POOL_RESIDENT;
    W.work_begin = lo; W.work_end = hi + 1;
    if(any_form) {
        W.form_ready = form_ready; W.cell_pos_xz = cell_pos; W.form_cohesion_xz = f_coh;
        W.form_align_xz = f_align; W.form_drag_xz = f_drag;
    }
    if(S.any_arrival) { W.arrival_sink_xz = S.sink; W.arrival_flags = S.arr_flags; }

    float *out_vel = calloc(2 * n + 2, sizeof(float)), *out_vdes = calloc(2 * n + 2, sizeof(float));
    uint8_t *status = calloc(n + 1, 1);
    navhip_step_out O = {out_vel, NULL, out_vdes, NULL, status};
    const double t_filled = hip_now();
    bool ok = hi >= lo && navhip_agent_step_submit(ctx, &W, &O) == NAVHIP_OK;
    /* (the nav task would Task_AwaitEvent(EVENT_UPDATE_START) here, like the GL path :4212-4233) */
    if(ok) ok = navhip_agent_step_wait(ctx) == NAVHIP_OK;
    const double t_stepped = hip_now();
    if(ok) {
        s_hip_stats[2]++;
        for(int w = begin_idx; w <= end_idx; w++) {
            const int i = DENSE(&S, s_move_work.in[w].ent_uid);
            if(status[i] & NAVHIP_ST_UNSUPPORTED) { ok = false; break; }
            if(isnan(vdes[2 * i]) && (status[i] & (NAVHIP_ST_FIELD_MISS | NAVHIP_ST_FIELD_NONE))) {
                /* the cases of nav.c:3483-3554 that need the planner or a repair build: the host samples
                 * (its builds go through the binding and land in the pool) and steps this one agent */
                struct move_work_in *in = &s_move_work.in[w];
                const struct flock *fl = flock_for_ent(in->ent_uid);
                in->ent_des_v = M_NavDesiredPointSeekVelocity(gs->map, fl->dest_id, in->cp_ent.xz_pos, fl->target_xz);
                s_move_work.out[w].ent_des_v = in->ent_des_v;
                in->dyn_neighbs->size = 0; in->stat_neighbs->size = 0;
                move_velocity_work(w, w);
                s_hip_stats[1]++;
                continue;
            }
            if(isnan(vdes[2 * i])) {
                /* what compute_desired_velocity (:4174-4175) would have left for the state update */
                s_hip_stats[0]++;
                s_move_work.in[w].ent_des_v = (vec2_t){out_vdes[2 * i], out_vdes[2 * i + 1]};
                s_move_work.out[w].ent_des_v = s_move_work.in[w].ent_des_v;
            }
            s_move_work.out[w].ent_vel = (vec2_t){out_vel[2 * i], out_vel[2 * i + 1]};
        }
    }
    hip_snap_free(&S);
    free(speed); free(los); free(form_ready); free(vdes); free(cell_pos); free(f_coh); free(f_align); free(f_drag);
    free(out_vel); free(out_vdes); free(status);
    s_hip_times[0] += t_filled - t_begin; s_hip_times[1] += t_stepped - t_filled;
    s_hip_times[2] += hip_now() - t_stepped; s_hip_times[3] += 1.0;
    return ok;
}

/* ---- the state-update half: fork_join_state_updates (movement.c:4196) -> move_update_task (:3496) ->
 * entity_compute_update (:2303) ------------------------------------------------------------------------
 * The data-parallel arm of its state switch (:2441-2520: arrived() with its three nav tests, the
 * arrived-neighbour rule, the no-guidance wait; the garrison rule :2344) runs on the device for every
 * work item at once (navhip_state_update).  The host keeps what is host state: the heading gate
 * (:2321-2334, orientation), the pose / interpolation patch, and the units the device hands back
 * (NAVHIP_SU_HOST: formations, active arrival groups, every other state).
 * move_hip_state_work(begin, end) leaves next state + blocker flag per work item; move_hip_update_work is
 * move_update_work (:3469) with the switch's outcome taken from there. */
int N_HIP_ClosestIslandTiles(struct nav_private *priv, enum nav_layer layer, vec3_t map_pos, vec2_t xz_dest,
                             int16_t *out_abs, int max_tiles);                    /* nav_hip.c */

static uint8_t *s_hip_su_state, *s_hip_su_flags;     /* [nwork] by work item */
static size_t   s_hip_su_cap;
static long     s_hip_su_stats[3];                   /* decided on the device, left to the host, passes */
void move_hip_state_stats(long out[3]) { memcpy(out, s_hip_su_stats, sizeof(s_hip_su_stats)); }

/* the velocity entity_compute_update integrates: zero while the unit still turns towards its heading */
static vec2_t hip_heading_gated(const struct movestate *ms, vec2_t vdes, vec2_t vel)
{
    if(!(PFM_Vec2_Len(&vel) > EPSILON) || !move_gated_by_heading(ms->state))
        return vel;
    quat_t want = dir_quat_from_velocity(intended_heading(vdes, vel));
    const float err = fabs(RAD_TO_DEG(PFM_Quat_PitchDiff((quat_t*)&ms->next_rot, &want)));
    const bool rolling = PFM_Vec2_Len((vec2_t*)&ms->velocity) > EPSILON;
    return err > (rolling ? MOVE_HEADING_HALT : MOVE_HEADING_RESUME) ? (vec2_t){0.0f, 0.0f} : vel;
}

static bool move_hip_state_work(int begin_idx, int end_idx)
{
    navhip_ctx *ctx = N_HIP_Ctx();
    if(!ctx || end_idx < begin_idx)
        return false;
    const struct move_gamestate *gs = &s_move_work.gamestate;
    struct hip_snap S;
    hip_snap_fill(&S);
    const int n = S.n;
    if(s_hip_su_cap < s_move_work.nwork) {
        s_hip_su_cap = s_move_work.nwork;
        s_hip_su_state = realloc(s_hip_su_state, s_hip_su_cap);
        s_hip_su_flags = realloc(s_hip_su_flags, s_hip_su_cap);
    }
    float *new_pos = calloc(2 * n + 2, sizeof(float)), *vdes = calloc(2 * n + 2, sizeof(float));
    uint8_t *skip = calloc(n + 1, 1);
    int lo = n, hi = -1;
    for(int w = begin_idx; w <= end_idx; w++) {
        const struct move_work_in *in = &s_move_work.in[w];
        const struct move_work_out *out = &s_move_work.out[w];
        const struct movestate *ms = movestate_get(in->ent_uid);
        const int i = DENSE(&S, in->ent_uid);
        vec2_t np = new_pos_for_vel(in->ent_uid, hip_heading_gated(ms, out->ent_des_v, out->ent_vel));
        new_pos[2 * i] = np.x; new_pos[2 * i + 1] = np.z;
        vdes[2 * i] = out->ent_des_v.x; vdes[2 * i + 1] = out->ent_des_v.z;
        /* a formation member (:2427-2437) or an active arrival group (:2443): the host's arms.  So is every unit
         * at a movement rate below 20 Hz: entity_compute_update then tests the INTERPOLATED intermediate position
         * (interpolate_positions(next_ppos, next_npos, ms->step), :2368-2377), not pos + vel */
        skip[i] = in->fstate.fid != NULL_FID || (20 / hz_count(s_move_work.hz)) > 1;
        if(!skip[i] && S.flock[i] >= 0) {
            struct flock *fl = &vec_AT(&s_flocks, S.flock[i]);
            struct arrival_state *as = G_ArrivalGroup_ForLayer(&fl->arrival,
                Entity_NavLayerWithRadius(S.flags[i], S.radius[i]));
            skip[i] = as && G_Arrival_IsActive(as);
        }
        if(i < lo) lo = i;
        if(i > hi) hi = i;
    }
    /* the two destination-only queries of arrived() (:2170), once per flock for the nav layer most of its
     * members path on (units of another layer come back as NAVHIP_SU_HOST) */
    const size_t F = S.nflocks;
    uint8_t *flayer = calloc(F + 1, 1);
    float   *nearest = malloc(sizeof(float) * 2 * (F + 1));
    int32_t *toff = calloc(F + 2, sizeof(int32_t));
    const int per = FIELD_RES_R * 2 + FIELD_RES_C * 2;
    int16_t *tiles = malloc(sizeof(int16_t) * 2 * per * (F + 1));
    vec3_t map_pos = move_hip_map_pos(gs->map);
    for(size_t f = 0; f < F; f++) {
        const struct flock *fl = &vec_AT(&s_flocks, f);
        nearest[2 * f] = NAN; nearest[2 * f + 1] = 0.0f;
        toff[f + 1] = toff[f];
        if(S.flock_offsets[f + 1] == S.flock_offsets[f])
            continue;
        int per_layer[NAV_LAYER_MAX] = {0}, best = 0;
        for(int m = S.flock_offsets[f]; m < S.flock_offsets[f + 1]; m++) {
            const int i = S.flock_members[m];
            per_layer[Entity_NavLayerWithRadius(S.flags[i], S.radius[i])]++;
        }
        for(int l = 1; l < NAV_LAYER_MAX; l++)
            if(per_layer[l] > per_layer[best]) best = l;
        const enum nav_layer layer = (enum nav_layer)best;
        flayer[f] = (uint8_t)layer;
        vec2_t near_xz;
        if(M_NavClosestPathable(gs->map, layer, fl->target_xz, &near_xz)) {
            nearest[2 * f] = near_xz.x; nearest[2 * f + 1] = near_xz.z;
        }
        toff[f + 1] += N_HIP_ClosestIslandTiles(move_hip_nav_private(gs->map), layer, map_pos, fl->target_xz,
                                                tiles + 2 * toff[f], per);
    }
    navhip_world W;
    hip_snap_world(&S, &W);
    W.work_begin = lo; W.work_end = hi + 1;
    navhip_state_in in = {new_pos, vdes, skip, flayer, nearest, toff, tiles};
    uint8_t *st = calloc(n + 1, 1), *fl = calloc(n + 1, 1);
    bool ok = hi >= lo && navhip_state_update(ctx, &W, &in, st, fl) == NAVHIP_OK;
    if(ok) {
        s_hip_su_stats[2]++;
        for(int w = begin_idx; w <= end_idx; w++) {
            const int i = DENSE(&S, s_move_work.in[w].ent_uid);
            s_hip_su_state[w] = st[i];
            s_hip_su_flags[w] = fl[i];
            s_hip_su_stats[(fl[i] & NAVHIP_SU_HOST) ? 1 : 0]++;
        }
    }
    hip_snap_free(&S);
    free(new_pos); free(vdes); free(skip); free(flayer); free(nearest); free(toff); free(tiles); free(st); free(fl);
    return ok;
}

/* move_update_work (:3469) for [begin_idx, end_idx] after move_hip_state_work: the pose half of the patch
 * from entity_compute_update as before; next state and blocker flag from the device pass for every unit it
 * decided.  (A maintainer splits entity_compute_update at :2437 and skips the switch for those units; the
 * harness cannot edit the function, so the switch still runs and its outcome is replaced.) */
static void move_hip_update_work(int begin_idx, int end_idx)
{
    for(int w = begin_idx; w <= end_idx; w++) {
        struct move_work_out *out = &s_move_work.out[w];
        entity_compute_update(s_move_work.hz, out->ent_uid, out->ent_vel, out->ent_des_v, &s_move_work.in[w], &out->patch);
        if(s_hip_su_flags[w] & NAVHIP_SU_HOST)
            continue;
        out->patch.flags = (enum movestate_flags)(out->patch.flags & ~UPDATE_SET_STATE);
        if(s_hip_su_flags[w] & NAVHIP_SU_SET_STATE) {
            out->patch.flags = (enum movestate_flags)(out->patch.flags | UPDATE_SET_STATE);
            out->patch.next_state = (enum move_state)s_hip_su_state[w];
            out->patch.next_block = (s_hip_su_flags[w] & NAVHIP_SU_BLOCK) != 0;
        }
    }
}
#undef DENSE
