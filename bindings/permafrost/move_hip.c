/* bindings/permafrost/move_hip.c -- the WORK_TYPE_HIP arm of fork_join_velocity_computations.
 *
 * What INTEGRATION.md tells a maintainer to add to src/game/movement.c next to the WORK_TYPE_CPU /
 * WORK_TYPE_GPU arms (movement.c:315-318,3737,4182-4194): fill a navhip_world from the tick's
 * snapshot tables (struct move_gamestate, :296) and work items (struct move_work_in, :264), run
 * navhip_agent_step, copy the velocities into s_move_work.out[] like copy_gpu_results does
 * (:4248-4261).  It lives in movement.c's translation unit because those tables are static; in this
 * repository the test harness #includes it right after movement.c (oracle/ref/ref_move.c).
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

/* move_velocity_work(begin_idx, end_idx) for the work items [begin_idx, end_idx] on the device.
 * Returns false when the library is not available (the caller runs the CPU arm). */
static bool move_hip_velocity_work(int begin_idx, int end_idx)
{
    navhip_ctx *ctx = N_HIP_Ctx();
    if(!ctx || end_idx < begin_idx)
        return false;
    const struct move_gamestate *gs = &s_move_work.gamestate;

    /* dense entity order: every entity of the position snapshot, ascending uid (the GL path densifies
     * uids the same way, ent_gpu_id_map :302) */
    const int n = (int)kh_size(gs->positions);
    uint32_t *uids = malloc(sizeof(uint32_t) * (n > 0 ? n : 1));
    {
        int k = 0;
        uint32_t key;
        kh_foreach_key(gs->positions, key, { uids[k++] = key; });
        qsort(uids, n, sizeof(uint32_t), cmp_u32);
    }
    khash_t(id) *dense = kh_init(id);
    for(int i = 0; i < n; i++) {
        int ret;
        khiter_t k = kh_put(id, dense, uids[i], &ret);
        kh_value(dense, k) = i;
    }
#define DENSE(uid) kh_value(dense, kh_get(id, dense, (uid)))

    float *pos = calloc(2 * n, sizeof(float)), *vel = calloc(2 * n, sizeof(float));
    float *radius = calloc(n, sizeof(float)), *max_speed = calloc(n, sizeof(float)), *speed = calloc(n, sizeof(float));
    uint32_t *flags = calloc(n, sizeof(uint32_t));
    uint8_t *state = calloc(n, 1), *los = calloc(n, 1), *form_ready = calloc(n, 1), *arr_flags = calloc(n, 1);
    int32_t *flock = malloc(sizeof(int32_t) * n);
    float *vdes = malloc(sizeof(float) * 2 * n), *cell_pos = calloc(2 * n, sizeof(float));
    float *f_coh = calloc(2 * n, sizeof(float)), *f_align = calloc(2 * n, sizeof(float)), *f_drag = calloc(2 * n, sizeof(float));
    float *sink = calloc(2 * n, sizeof(float));
    bool any_arrival = false;

    const size_t nflocks = vec_size(&s_flocks);
    float *flock_target = calloc(2 * (nflocks ? nflocks : 1), sizeof(float));
    int32_t *flock_offsets = calloc(nflocks + 1, sizeof(int32_t));
    int32_t *flock_members = malloc(sizeof(int32_t) * (n > 0 ? n : 1));

    for(int i = 0; i < n; i++) {
        const uint32_t uid = uids[i];
        vec2_t p = G_Pos_GetXZFrom(gs->positions, uid);
        pos[2 * i] = p.x; pos[2 * i + 1] = p.z;
        flags[i] = G_FlagsGetFrom(gs->flags, uid);
        radius[i] = G_GetSelectionRadiusFrom(gs->sel_radiuses, uid);
        flock[i] = -1;
        vdes[2 * i] = vdes[2 * i + 1] = 0.0f;
        state[i] = STATE_ARRIVED;                  /* no movestate: a still obstacle */
        khiter_t k = kh_get(state, s_entity_state_table, uid);
        if(k == kh_end(s_entity_state_table))
            continue;
        const struct movestate *ms = &kh_value(s_entity_state_table, k);
        state[i] = (uint8_t)ms->state;
        vel[2 * i] = ms->velocity.x; vel[2 * i + 1] = ms->velocity.z;
        max_speed[i] = ms->max_speed;
        /* struct arrival_unit_state: committed to a valid slot (unit_committed, arrival.c:90) */
        if((ms->arrival.substate == ARRIVAL_SUBSTATE_SEEK || ms->arrival.substate == ARRIVAL_SUBSTATE_SEEK_ARMED)
        && ms->arrival.sink_valid) {
            arr_flags[i] |= 1;
            any_arrival = true;
        }
        sink[2 * i] = ms->arrival.sink.x; sink[2 * i + 1] = ms->arrival.sink.z;
    }
    /* flocks: members in kh_foreach order of flock->ents (the order cohesion_force sums in, :1660) */
    {
        int at = 0;
        for(size_t f = 0; f < nflocks; f++) {
            const struct flock *fl = &vec_AT(&s_flocks, f);
            flock_target[2 * f] = fl->target_xz.x; flock_target[2 * f + 1] = fl->target_xz.z;
            flock_offsets[f] = at;
            uint32_t curr;
            kh_foreach_key(fl->ents, curr, {
                const int i = DENSE(curr);
                flock_members[at++] = i;
                flock[i] = (int32_t)f;
                const struct arrival_state *as = G_ArrivalGroup_ForLayer(&fl->arrival,
                    Entity_NavLayerWithRadius(flags[i], radius[i]));
                if(as && as->phase == ARRIVAL_PHASE_FILLING) { arr_flags[i] |= 2; any_arrival = true; }
            });
        }
        flock_offsets[nflocks] = at;
    }
    /* work items */
    int lo = n, hi = -1;
    bool any_form = false, ok_pool = true;
    for(int w = begin_idx; w <= end_idx; w++) {
        const struct move_work_in *in = &s_move_work.in[w];
        const int i = DENSE(in->ent_uid);
        vdes[2 * i] = in->ent_des_v.x; vdes[2 * i + 1] = in->ent_des_v.z;
        speed[i] = in->speed;
        los[i] = in->has_dest_los;
        form_ready[i] = in->fstate.assignment_ready;
        cell_pos[2 * i] = in->cell_pos.x; cell_pos[2 * i + 1] = in->cell_pos.z;
        f_coh[2 * i] = in->fstate.normal_cohesion_force.x; f_coh[2 * i + 1] = in->fstate.normal_cohesion_force.z;
        f_align[2 * i] = in->fstate.normal_align_force.x; f_align[2 * i + 1] = in->fstate.normal_align_force.z;
        f_drag[2 * i] = in->fstate.normal_drag_force.x; f_drag[2 * i + 1] = in->fstate.normal_drag_force.z;
        any_form = any_form || state[i] == STATE_MOVING_IN_FORMATION || state[i] == STATE_ARRIVING_TO_CELL;
        if(i < lo) lo = i;
        if(i > hi) hi = i;
    }
    /* the device steps the contiguous uid slab [lo, hi]; entities inside it that carry no work item
     * (other slabs of a threaded split) are stepped too and their results dropped */
    struct nav_private *priv = move_hip_nav_private(gs->map);
    const bool sample = s_hip_sample_on_device && N_HIP_PoolOn();
    if(sample) {
        /* flock index = mapping row of the resident pool: announce every flock's destination, flush the
         * mappings the planner recorded since the last tick, and leave the sampling of the point-seeking
         * agents (the default arm of ent_desired_velocity, :1510-1521) to the device: vdes.x = NaN */
        dest_id_t *fdest = malloc(sizeof(dest_id_t) * (nflocks ? nflocks : 1));
        for(size_t f = 0; f < nflocks; f++)
            fdest[f] = vec_AT(&s_flocks, f).dest_id;
        N_HIP_PoolSetRows(priv, (int)nflocks, fdest);
        free(fdest);
        if(!N_HIP_PoolSync()) {
            ok_pool = false;
        }else{
            for(int w = begin_idx; w <= end_idx; w++) {
                const int i = DENSE(s_move_work.in[w].ent_uid);
                if(state[i] == STATE_MOVING && flock[i] >= 0 && !(arr_flags[i] & 2))
                    vdes[2 * i] = NAN;
            }
        }
    }
    navhip_world W;
    memset(&W, 0, sizeof(W));
    W.n_ents = n; W.n_flocks = (int32_t)nflocks; W.hz = hz_count(s_move_work.hz);
    W.pos_xz = pos; W.vel_xz = vel; W.radius = radius; W.max_speed = max_speed; W.speed = speed;
    W.flags = flags; W.state = state; W.has_dest_los = los; W.flock = flock; W.vdes_xz = vdes;
    W.flock_target_xz = flock_target; W.flock_offsets = flock_offsets; W.flock_members = flock_members;
    vec3_t map_pos = move_hip_map_pos(gs->map);
    W.map_pos_x = map_pos.x; W.map_pos_z = map_pos.z;
    if(sample && ok_pool)
        W.n_field_slots = NAVHIP_POOL_RESIDENT;
    {
        /* bg_ent_init bounds of the position snapshot (position.c:276-283) */
        const struct nav_private *np = priv;
        float half_x = np->width * TILES_PER_CHUNK_WIDTH * X_COORDS_PER_TILE / 2.0f;
        float half_z = np->height * TILES_PER_CHUNK_HEIGHT * Z_COORDS_PER_TILE / 2.0f;
        float cx = map_pos.x - half_x, cz = map_pos.z + half_z;
        W.grid_xmin = cx - half_x; W.grid_xmax = cx + half_x; W.grid_zmin = cz - half_z; W.grid_zmax = cz + half_z;
    }
    W.work_begin = lo; W.work_end = hi + 1;
    if(any_form) {
        W.form_ready = form_ready; W.cell_pos_xz = cell_pos; W.form_cohesion_xz = f_coh;
        W.form_align_xz = f_align; W.form_drag_xz = f_drag;
    }
    if(any_arrival) { W.arrival_sink_xz = sink; W.arrival_flags = arr_flags; }
    W.static_epoch = s_hip_attr_epoch;

    float *out_vel = calloc(2 * n, sizeof(float));
    uint8_t *status = calloc(n, 1);
    navhip_step_out O = {out_vel, NULL, NULL, NULL, status};
    bool ok = hi >= lo && navhip_agent_step_submit(ctx, &W, &O) == NAVHIP_OK;
    /* (the nav task would Task_AwaitEvent(EVENT_UPDATE_START) here, like the GL path :4212-4233) */
    if(ok) ok = navhip_agent_step_wait(ctx) == NAVHIP_OK;
    if(ok) {
        s_hip_stats[2]++;
        for(int w = begin_idx; w <= end_idx; w++) {
            const int i = DENSE(s_move_work.in[w].ent_uid);
            if(status[i] & NAVHIP_ST_UNSUPPORTED) { ok = false; break; }
            if(isnan(vdes[2 * i]) && (status[i] & (NAVHIP_ST_FIELD_MISS | NAVHIP_ST_FIELD_NONE))) {
                /* the cases of nav.c:3483-3554 that need the planner or a repair build: the host samples
                 * (its builds go through the binding and land in the pool) and steps this one agent */
                struct move_work_in *in = &s_move_work.in[w];
                const struct flock *fl = flock_for_ent(in->ent_uid);
                in->ent_des_v = M_NavDesiredPointSeekVelocity(gs->map, fl->dest_id, in->cp_ent.xz_pos, fl->target_xz);
                in->dyn_neighbs->size = 0; in->stat_neighbs->size = 0;
                move_velocity_work(w, w);
                s_hip_stats[1]++;
                continue;
            }
            if(isnan(vdes[2 * i])) s_hip_stats[0]++;
            s_move_work.out[w].ent_vel = (vec2_t){out_vel[2 * i], out_vel[2 * i + 1]};
        }
    }
#undef DENSE
    kh_destroy(id, dense);
    free(uids); free(pos); free(vel); free(radius); free(max_speed); free(speed); free(flags); free(state);
    free(los); free(form_ready); free(arr_flags); free(flock); free(vdes); free(cell_pos); free(f_coh);
    free(f_align); free(f_drag); free(sink); free(flock_target); free(flock_offsets); free(flock_members);
    free(out_vel); free(status);
    return ok;
}
