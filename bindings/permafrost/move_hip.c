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
#include <time.h>      /* hip_now: clock_gettime (movement.c itself does not include it) */

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

/* The loops over entities and work items below are range functions: the engine forks them over its worker tasks
 * the way move_submit_cpu_work (:3751-3783) forks move_velocity_work; a host registers that fork-join here (the
 * harness: oracle/ref/ref_move.c).  Without one they run on the calling task. */
static double hip_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
typedef void (*hip_range_fn)(int begin, int end, void *arg);
static void (*s_hip_parallel_for)(hip_range_fn fn, int n, void *arg);
static int s_hip_parallel_min = 8192;                 /* shorter loops stay on the calling task */
void move_hip_set_parallel_for(void (*pf)(hip_range_fn fn, int n, void *arg), int min_items)
{
    s_hip_parallel_for = pf;
    s_hip_parallel_min = min_items > 0 ? min_items : 8192;
}
static void hip_for(hip_range_fn fn, int n, void *arg)
{
    if(s_hip_parallel_for && n >= s_hip_parallel_min) s_hip_parallel_for(fn, n, arg);
    else if(n > 0) fn(0, n, arg);
}

/* (loops whose items are expensive -- a nav query each -- fork from `min_items` on) */
static void hip_for_min(hip_range_fn fn, int n, void *arg, int min_items)
{
    if(s_hip_parallel_for && n >= min_items) s_hip_parallel_for(fn, n, arg);
    else if(n > 0) fn(0, n, arg);
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
    bool         any_arrival, any_form;
};
#define DENSE(S, uid) kh_value((S)->dense, kh_get(id, (S)->dense, (uid)))

/* The sorted uid list and the uid -> dense index map only change when the entity set does; sorting 100 000 uids
 * and building the map were two thirds of the 4.7 ms a tick spent filling the snapshot (bench.py `dropin`).  They
 * are kept between ticks and rebuilt when the set has another size or a cached uid is gone from the tick's
 * position table (an equal count with every cached uid present IS the same set). */
static struct {
    int n; uint32_t *uids; khash_t(id) *dense;
    /* where each uid was found last tick in the four per-entity tables: a snapshot table is a kh_copy of a table
     * that rarely rehashes, so the bucket is still the uid's (checked: occupied + same key) and the lookup is one
     * indexed read instead of a hash and a probe sequence; any miss falls back to kh_get */
    khint_t *it_pos, *it_flags, *it_rad, *it_state;
    /* the flock tables of the snapshot, kept while the flock epoch (move_hip_attrs_changed) and the set stand */
    uint32_t flock_epoch; size_t nflocks; int32_t *flock, *flock_offsets, *flock_members;
} s_hip_set;
#define HIP_IT(h, name, cache, uid) \
    ((cache) < kh_end(h) && kh_exist(h, cache) && kh_key(h, cache) == (uid) ? (cache) : ((cache) = kh_get(name, h, uid)))

static uint32_t s_hip_set_epoch;
static void hip_set_rebuild(const struct move_gamestate *gs)
{
    s_hip_set_epoch++;
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
    const size_t m = (size_t)(n > 0 ? n : 1);
    s_hip_set.it_pos = realloc(s_hip_set.it_pos, m * sizeof(khint_t));
    s_hip_set.it_flags = realloc(s_hip_set.it_flags, m * sizeof(khint_t));
    s_hip_set.it_rad = realloc(s_hip_set.it_rad, m * sizeof(khint_t));
    s_hip_set.it_state = realloc(s_hip_set.it_state, m * sizeof(khint_t));
    s_hip_set.flock = realloc(s_hip_set.flock, m * sizeof(int32_t));
    s_hip_set.flock_members = realloc(s_hip_set.flock_members, m * sizeof(int32_t));
    memset(s_hip_set.it_pos, 0xff, m * sizeof(khint_t)); memset(s_hip_set.it_flags, 0xff, m * sizeof(khint_t));
    memset(s_hip_set.it_rad, 0xff, m * sizeof(khint_t)); memset(s_hip_set.it_state, 0xff, m * sizeof(khint_t));
    s_hip_set.flock_epoch = 0;                        /* (0 is never a valid epoch: the flock tables are rebuilt) */
}

/* Every per-tick array lives in one arena that only grows (s_hip_arena): a tick of 100 000 entities used to
 * calloc and free two dozen arrays of 0.4-0.8 MB -- each an mmap, its page faults and an munmap -- which was a
 * third of the time the binding spent on the host (bench.py `dropin`).  Nothing relies on zeroed memory any more:
 * the entity loop below writes every column of every row. */
static struct { char *base; size_t cap, used; } s_hip_arena;
static void hip_arena_reset(size_t need)
{
    if(need > s_hip_arena.cap) {
        free(s_hip_arena.base);
        s_hip_arena.cap = need + need / 4;
        s_hip_arena.base = malloc(s_hip_arena.cap);
        if(!s_hip_arena.base) {
            fprintf(stderr, "move_hip: out of memory for the per-tick arena (%zu bytes)\n", s_hip_arena.cap);
            abort();
        }
    }
    s_hip_arena.used = 0;
}
static void *hip_arena(size_t bytes)
{
    bytes = (bytes + 63) & ~(size_t)63;
    if(s_hip_arena.used + bytes > s_hip_arena.cap) {
        /* (hip_arena_need is an upper bound of what a pass takes: reaching this line is a bug in it, and handing
         * out memory past the end would be worse than stopping) */
        fprintf(stderr, "move_hip: per-tick arena exhausted (%zu + %zu of %zu bytes)\n", s_hip_arena.used, bytes, s_hip_arena.cap);
        abort();
    }
    void *p = s_hip_arena.base + s_hip_arena.used;
    s_hip_arena.used += bytes;
    return p;
}
/* bytes of the arena one tick may take for n entities, F flocks (both passes: velocity or state) */
static size_t hip_arena_need(size_t n, size_t F)
{
    /* (twice: the state pass of a tick keeps the velocity pass's snapshot and takes its own arrays behind it) */
    return 2 * ((n + 16) * (4 * 68 + 8) + (F + 2) * (8 + 4 + 8 + 4 + 1 + NAV_LAYER_MAX + 2 * 2 * (FIELD_RES_R * 2 + FIELD_RES_C * 2)) + 64 * 64);
}

static void hip_check_range(int begin, int end, void *arg)
{
    const struct move_gamestate *gs = &s_move_work.gamestate;
    for(int i = begin; i < end; i++)
        if(HIP_IT(gs->positions, pos, s_hip_set.it_pos[i], s_hip_set.uids[i]) == kh_end(gs->positions)) {
            __atomic_store_n((int*)arg, 1, __ATOMIC_RELAXED);
            return;
        }
}

static void hip_fill_range(int begin, int end, void *arg)
{
    struct hip_snap *S = arg;
    const struct move_gamestate *gs = &s_move_work.gamestate;
    bool any_arrival = false, any_form = false;
    for(int i = begin; i < end; i++) {
        const uint32_t uid = S->uids[i];
        khiter_t k = HIP_IT(gs->positions, pos, s_hip_set.it_pos[i], uid);        /* G_Pos_GetXZFrom, position.c:196 */
        assert(k != kh_end(gs->positions));
        const vec3_t p = kh_val(gs->positions, k);
        S->pos[2 * i] = p.x; S->pos[2 * i + 1] = p.z;
        k = HIP_IT(gs->flags, id, s_hip_set.it_flags[i], uid);                    /* G_FlagsGetFrom, game.c:2391 */
        assert(k != kh_end(gs->flags));
        S->flags[i] = kh_value(gs->flags, k);
        k = HIP_IT(gs->sel_radiuses, range, s_hip_set.it_rad[i], uid);            /* G_GetSelectionRadiusFrom, game.c:2862 */
        assert(k != kh_end(gs->sel_radiuses));
        S->radius[i] = kh_value(gs->sel_radiuses, k);
        S->arr_flags[i] = 0;
        k = HIP_IT(s_entity_state_table, state, s_hip_set.it_state[i], uid);
        if(k == kh_end(s_entity_state_table)) {
            S->state[i] = STATE_ARRIVED;              /* no movestate: a still obstacle */
            S->vel[2 * i] = S->vel[2 * i + 1] = 0.0f;
            S->max_speed[i] = 0.0f;
            S->sink[2 * i] = S->sink[2 * i + 1] = 0.0f;
            continue;
        }
        const struct movestate *ms = &kh_value(s_entity_state_table, k);
        S->state[i] = (uint8_t)ms->state;
        S->vel[2 * i] = ms->velocity.x; S->vel[2 * i + 1] = ms->velocity.z;
        S->max_speed[i] = ms->max_speed;
        /* struct arrival_unit_state: committed to a valid slot (unit_committed, arrival.c:90) */
        if((ms->arrival.substate == ARRIVAL_SUBSTATE_SEEK || ms->arrival.substate == ARRIVAL_SUBSTATE_SEEK_ARMED)
        && ms->arrival.sink_valid) {
            S->arr_flags[i] |= 1;
            any_arrival = true;
        }
        S->sink[2 * i] = ms->arrival.sink.x; S->sink[2 * i + 1] = ms->arrival.sink.z;
        any_form = any_form || ms->state == STATE_MOVING_IN_FORMATION || ms->state == STATE_ARRIVING_TO_CELL;
    }
    if(any_arrival) __atomic_store_n(&S->any_arrival, true, __ATOMIC_RELAXED);
    if(any_form) __atomic_store_n(&S->any_form, true, __ATOMIC_RELAXED);
}

static void hip_snap_fill(struct hip_snap *S)
{
    const struct move_gamestate *gs = &s_move_work.gamestate;
    memset(S, 0, sizeof(*S));
    const int n = S->n = (int)kh_size(gs->positions);
    if(!s_hip_set.dense || s_hip_set.n != n)
        hip_set_rebuild(gs);
    else {
        /* (an equal count with every cached uid present IS the same set) */
        int gone = 0;
        hip_for(hip_check_range, n, &gone);
        if(gone) hip_set_rebuild(gs);
    }
    S->uids = s_hip_set.uids;
    S->dense = s_hip_set.dense;
    S->nflocks = vec_size(&s_flocks);
    hip_arena_reset(hip_arena_need((size_t)n, S->nflocks));
    S->pos = hip_arena(sizeof(float) * (2 * n + 2)); S->vel = hip_arena(sizeof(float) * (2 * n + 2));
    S->radius = hip_arena(sizeof(float) * (n + 1)); S->max_speed = hip_arena(sizeof(float) * (n + 1));
    S->sink = hip_arena(sizeof(float) * (2 * n + 2));
    S->flags = hip_arena(sizeof(uint32_t) * (n + 1));
    S->state = hip_arena(n + 1); S->arr_flags = hip_arena(n + 1);
    S->flock_target = hip_arena(sizeof(float) * 2 * (S->nflocks ? S->nflocks : 1));
    hip_for(hip_fill_range, n, S);
    /* flocks: members in kh_foreach order of flock->ents (the order cohesion_force sums in, :1660).  Membership
     * only changes where the flock epoch is bumped (see move_hip_attrs_changed): the three tables are kept */
    if(s_hip_set.flock_epoch != s_hip_attr_epoch || s_hip_set.nflocks != S->nflocks) {
        s_hip_set.flock_offsets = realloc(s_hip_set.flock_offsets, sizeof(int32_t) * (S->nflocks + 1));
        for(int i = 0; i < n; i++) s_hip_set.flock[i] = -1;
        int at = 0;
        for(size_t f = 0; f < S->nflocks; f++) {
            const struct flock *fl = &vec_AT(&s_flocks, f);
            s_hip_set.flock_offsets[f] = at;
            uint32_t curr;
            kh_foreach_key(fl->ents, curr, {
                const int i = DENSE(S, curr);
                s_hip_set.flock_members[at++] = i;
                s_hip_set.flock[i] = (int32_t)f;
            });
        }
        s_hip_set.flock_offsets[S->nflocks] = at;
        s_hip_set.nflocks = S->nflocks;
        s_hip_set.flock_epoch = s_hip_attr_epoch;
    }
    S->flock = s_hip_set.flock; S->flock_offsets = s_hip_set.flock_offsets; S->flock_members = s_hip_set.flock_members;
    for(size_t f = 0; f < S->nflocks; f++) {
        const struct flock *fl = &vec_AT(&s_flocks, f);
        S->flock_target[2 * f] = fl->target_xz.x; S->flock_target[2 * f + 1] = fl->target_xz.z;
        /* members of an arrival group that is filling (per flock and nav layer; usually none) */
        for(int layer = 0; layer < NAV_LAYER_MAX; layer++) {
            const struct arrival_state *as = G_ArrivalGroup_ForLayer(&fl->arrival, (enum nav_layer)layer);
            if(!as || as->phase != ARRIVAL_PHASE_FILLING)
                continue;
            for(int m = S->flock_offsets[f]; m < S->flock_offsets[f + 1]; m++) {
                const int i = S->flock_members[m];
                if(Entity_NavLayerWithRadius(S->flags[i], S->radius[i]) == (enum nav_layer)layer) {
                    S->arr_flags[i] |= 2; S->any_arrival = true;
                }
            }
        }
    }
}

static void hip_snap_free(struct hip_snap *S)
{
    (void)S;                                          /* (the arena is reused by the next tick) */
}

/* what the velocity pass of this tick left behind for its state pass (move_hip_state_work) */
static struct { bool valid; struct hip_snap S; int begin_idx, end_idx, lo, hi; size_t nwork; } s_hip_last;
/* The state pass on the device-resident snapshot of the velocity pass (navhip_state_pass_resident).  On in an engine
 * build (the tick runs the two fork-joins back to back, movement.c:4263-4280); the harness, whose tests call the state
 * pass on its own with velocities of their choosing, switches it on where a velocity pass precedes. */
static bool s_hip_state_resident;
static long s_hip_resident_passes;                   /* state passes that ran on the resident snapshot */
void move_hip_set_resident_state_pass(bool on) { s_hip_state_resident = on; }
long move_hip_resident_passes(void) { return s_hip_resident_passes; }
/* per-unit arrays of the resident pass in page-locked memory (navhip_host_alloc): transferred in place */
static struct { size_t cap; float *next_rot, *new_pos; uint8_t *fstate, *wait_prev, *skip, *st, *fl, *gate; int32_t *wait_ticks, *wait_after; } s_hip_pin;
static bool hip_pin_reserve(size_t n)
{
    if(n + 1 <= s_hip_pin.cap) return true;
    navhip_host_free(s_hip_pin.next_rot); navhip_host_free(s_hip_pin.new_pos); navhip_host_free(s_hip_pin.fstate);
    navhip_host_free(s_hip_pin.wait_prev); navhip_host_free(s_hip_pin.skip); navhip_host_free(s_hip_pin.st); navhip_host_free(s_hip_pin.fl);
    navhip_host_free(s_hip_pin.gate); navhip_host_free(s_hip_pin.wait_ticks); navhip_host_free(s_hip_pin.wait_after);
    const size_t cap = n + n / 4 + 64;
    s_hip_pin.next_rot = navhip_host_alloc(sizeof(float) * 4 * cap); s_hip_pin.new_pos = navhip_host_alloc(sizeof(float) * 2 * cap);
    s_hip_pin.fstate = navhip_host_alloc(cap); s_hip_pin.wait_prev = navhip_host_alloc(cap); s_hip_pin.skip = navhip_host_alloc(cap);
    s_hip_pin.st = navhip_host_alloc(cap); s_hip_pin.fl = navhip_host_alloc(cap); s_hip_pin.gate = navhip_host_alloc(cap);
    s_hip_pin.wait_ticks = navhip_host_alloc(sizeof(int32_t) * cap); s_hip_pin.wait_after = navhip_host_alloc(sizeof(int32_t) * cap);
    s_hip_pin.cap = (s_hip_pin.next_rot && s_hip_pin.new_pos && s_hip_pin.fstate && s_hip_pin.wait_prev && s_hip_pin.skip && s_hip_pin.st
                     && s_hip_pin.fl && s_hip_pin.gate && s_hip_pin.wait_ticks && s_hip_pin.wait_after) ? cap : 0;
    if(s_hip_pin.cap) {
        /* rows of the slab without a work item are computed and dropped, on whatever these arrays hold: page-locked
         * memory has no defined content before its first use, so it is given one here */
        memset(s_hip_pin.next_rot, 0, sizeof(float) * 4 * cap); memset(s_hip_pin.new_pos, 0, sizeof(float) * 2 * cap);
        memset(s_hip_pin.fstate, 0, cap); memset(s_hip_pin.wait_prev, 0, cap); memset(s_hip_pin.skip, 0, cap);
        memset(s_hip_pin.st, 0, cap); memset(s_hip_pin.fl, 0, cap); memset(s_hip_pin.gate, 0, cap);
        memset(s_hip_pin.wait_ticks, 0, sizeof(int32_t) * cap); memset(s_hip_pin.wait_after, 0, sizeof(int32_t) * cap);
    }
    return s_hip_pin.cap != 0;
}

/* dense index of every work item, kept between ticks: the work list is rebuilt every tick in the same entity
 * order (movement.c:4129-4170), so an index is valid while the item still carries the uid it was looked up for */
static struct { size_t cap; uint32_t *uid; int32_t *idx; uint32_t set_epoch; } s_hip_witem;
static void hip_work_dense_prepare(void)              /* (once per pass, on the calling task) */
{
    if(s_move_work.nwork > s_hip_witem.cap) {
        const size_t cap = s_move_work.nwork;
        s_hip_witem.uid = realloc(s_hip_witem.uid, cap * sizeof(uint32_t));
        s_hip_witem.idx = realloc(s_hip_witem.idx, cap * sizeof(int32_t));
        for(size_t k = s_hip_witem.cap; k < cap; k++) s_hip_witem.idx[k] = -1;
        s_hip_witem.cap = cap;
    }
    if(s_hip_witem.set_epoch != s_hip_set_epoch) {
        for(size_t k = 0; k < s_hip_witem.cap; k++) s_hip_witem.idx[k] = -1;
        s_hip_witem.set_epoch = s_hip_set_epoch;
    }
}
static inline int hip_work_dense(const struct hip_snap *S, int w)
{
    const uint32_t uid = s_move_work.in[w].ent_uid;
    if(s_hip_witem.idx[w] < 0 || s_hip_witem.uid[w] != uid) {
        s_hip_witem.uid[w] = uid;
        s_hip_witem.idx[w] = DENSE(S, uid);
    }
    return s_hip_witem.idx[w];
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
void move_hip_times(double out[4], int reset)
{
    memcpy(out, s_hip_times, sizeof(s_hip_times));
    if(reset) memset(s_hip_times, 0, sizeof(s_hip_times));
}

/* (harness only: time the host side without a device -- the step itself is skipped, its outputs read as zero) */
static bool s_hip_dry_run;
void move_hip_set_dry_run(bool on) { s_hip_dry_run = on; }

/* one velocity pass: what its range functions share */
struct hip_vel_pass {
    struct hip_snap *S;
    int    begin_idx;
    float *speed, *vdes, *cell_pos, *f_coh, *f_align, *f_drag;
    uint8_t *los, *form_ready;
    bool   any_form, sample;
    const float *out_vel, *out_vdes; const uint8_t *status;
    int    unsupported;                               /* a work item came back NAVHIP_ST_UNSUPPORTED */
    int    n_fallback; int32_t *fallback;             /* work items the host samples and steps itself */
    long   sampled;
};

static void hip_vel_items_range(int begin, int end, void *arg)
{
    struct hip_vel_pass *V = arg;
    const struct hip_snap *S = V->S;
    for(int k = begin; k < end; k++) {
        const int w = V->begin_idx + k;
        const struct move_work_in *in = &s_move_work.in[w];
        const int i = hip_work_dense(S, w);
        V->vdes[2 * i] = in->ent_des_v.x; V->vdes[2 * i + 1] = in->ent_des_v.z;
        V->speed[i] = in->speed;
        V->los[i] = in->has_dest_los;
        if(V->any_form) {
            V->form_ready[i] = in->fstate.assignment_ready;
            V->cell_pos[2 * i] = in->cell_pos.x; V->cell_pos[2 * i + 1] = in->cell_pos.z;
            V->f_coh[2 * i] = in->fstate.normal_cohesion_force.x; V->f_coh[2 * i + 1] = in->fstate.normal_cohesion_force.z;
            V->f_align[2 * i] = in->fstate.normal_align_force.x; V->f_align[2 * i + 1] = in->fstate.normal_align_force.z;
            V->f_drag[2 * i] = in->fstate.normal_drag_force.x; V->f_drag[2 * i + 1] = in->fstate.normal_drag_force.z;
        }
        /* device sampling: the point-seeking agents (the default arm of ent_desired_velocity, :1510-1521) */
        if(V->sample && S->state[i] == STATE_MOVING && S->flock[i] >= 0 && !(S->arr_flags[i] & 2))
            V->vdes[2 * i] = NAN;
    }
}

static void hip_vel_scatter_range(int begin, int end, void *arg)
{
    struct hip_vel_pass *V = arg;
    const struct hip_snap *S = V->S;
    long sampled = 0;
    for(int k = begin; k < end; k++) {
        const int w = V->begin_idx + k;
        const int i = hip_work_dense(S, w);
        if(V->status[i] & NAVHIP_ST_UNSUPPORTED) { __atomic_store_n(&V->unsupported, 1, __ATOMIC_RELAXED); return; }
        if(isnan(V->vdes[2 * i]) && (V->status[i] & (NAVHIP_ST_FIELD_MISS | NAVHIP_ST_FIELD_NONE))) {
            /* (the planner / repair cases: the calling task handles them after the join) */
            V->fallback[__atomic_fetch_add(&V->n_fallback, 1, __ATOMIC_RELAXED)] = w;
            continue;
        }
        if(isnan(V->vdes[2 * i])) {
            /* what compute_desired_velocity (:4174-4175) would have left for the state update */
            sampled++;
            s_move_work.in[w].ent_des_v = (vec2_t){V->out_vdes[2 * i], V->out_vdes[2 * i + 1]};
            s_move_work.out[w].ent_des_v = s_move_work.in[w].ent_des_v;
        }
        s_move_work.out[w].ent_vel = (vec2_t){V->out_vel[2 * i], V->out_vel[2 * i + 1]};
    }
    if(sampled) __atomic_fetch_add(&V->sampled, sampled, __ATOMIC_RELAXED);
}

static int cmp_i32(const void *a, const void *b) { return (*(const int32_t*)a > *(const int32_t*)b) - (*(const int32_t*)a < *(const int32_t*)b); }


static bool move_hip_velocity_work(int begin_idx, int end_idx)
{
    navhip_ctx *ctx = s_hip_dry_run ? NULL : N_HIP_Ctx();
    if((!ctx && !s_hip_dry_run) || end_idx < begin_idx)
        return false;
    const double t_begin = hip_now();
    const struct move_gamestate *gs = &s_move_work.gamestate;
    struct hip_snap S;
    hip_snap_fill(&S);
    const int n = S.n, nitems = end_idx - begin_idx + 1;

    struct hip_vel_pass V;
    memset(&V, 0, sizeof(V));
    V.S = &S; V.begin_idx = begin_idx; V.any_form = S.any_form;
    V.speed = hip_arena(sizeof(float) * (n + 1));
    V.los = hip_arena(n + 1);
    V.vdes = hip_arena(sizeof(float) * (2 * n + 2));
    /* rows without a work item inside the stepped slab (other slabs of a threaded split, entities without a
     * movestate) are stepped too and their results dropped: they only need defined inputs */
    memset(V.speed, 0, sizeof(float) * (n + 1)); memset(V.los, 0, n + 1); memset(V.vdes, 0, sizeof(float) * (2 * n + 2));
    if(V.any_form) {
        V.form_ready = hip_arena(n + 1); memset(V.form_ready, 0, n + 1);
        V.cell_pos = hip_arena(sizeof(float) * (2 * n + 2)); memset(V.cell_pos, 0, sizeof(float) * (2 * n + 2));
        V.f_coh = hip_arena(sizeof(float) * (2 * n + 2)); memset(V.f_coh, 0, sizeof(float) * (2 * n + 2));
        V.f_align = hip_arena(sizeof(float) * (2 * n + 2)); memset(V.f_align, 0, sizeof(float) * (2 * n + 2));
        V.f_drag = hip_arena(sizeof(float) * (2 * n + 2)); memset(V.f_drag, 0, sizeof(float) * (2 * n + 2));
    }
    bool sample = !s_hip_dry_run && s_hip_sample_on_device && N_HIP_PoolOn();
    if(sample) {
        /* flock index = mapping row of the resident pool: announce every flock's destination, flush the
         * mappings the planner recorded since the last tick, and leave the sampling of the point-seeking
         * agents to the device: vdes.x = NaN */
        dest_id_t *fdest = malloc(sizeof(dest_id_t) * (S.nflocks ? S.nflocks : 1));
        for(size_t f = 0; f < S.nflocks; f++)
            fdest[f] = vec_AT(&s_flocks, f).dest_id;
        N_HIP_PoolSetRows(move_hip_nav_private(gs->map), (int)S.nflocks, fdest);
        free(fdest);
        sample = N_HIP_PoolSync();
    }
    V.sample = sample;

    /* work items */
    hip_work_dense_prepare();
    hip_for(hip_vel_items_range, nitems, &V);
    /* the device steps the contiguous uid slab [lo, hi]; entities inside it that carry no work item
     * (other slabs of a threaded split) are stepped too and their results dropped */
    int lo = n, hi = -1;
    for(int w = begin_idx; w <= end_idx; w++) {
        const int i = s_hip_witem.idx[w];
        if(i < lo) lo = i;
        if(i > hi) hi = i;
    }
    navhip_world W;
    hip_snap_world(&S, &W);
    W.speed = V.speed; W.has_dest_los = V.los; W.vdes_xz = V.vdes;
    if(sample)
        W.n_field_slots = NAVHIP_POOL_RESIDENT;
    W.work_begin = lo; W.work_end = hi + 1;
    if(V.any_form) {
        W.form_ready = V.form_ready; W.cell_pos_xz = V.cell_pos; W.form_cohesion_xz = V.f_coh;
        W.form_align_xz = V.f_align; W.form_drag_xz = V.f_drag;
    }
    if(S.any_arrival) { W.arrival_sink_xz = S.sink; W.arrival_flags = S.arr_flags; }

    /* (navhip_agent_step writes the rows of the stepped slab [lo, hi]: every row read below) */
    float *out_vel = hip_arena(sizeof(float) * (2 * n + 2)), *out_vdes = hip_arena(sizeof(float) * (2 * n + 2));
    uint8_t *status = hip_arena(n + 1);
    V.fallback = hip_arena(sizeof(int32_t) * (nitems + 1));
    navhip_step_out O = {out_vel, NULL, out_vdes, NULL, status};
    const double t_filled = hip_now();
    bool ok = hi >= lo;
    if(s_hip_dry_run) {
        memset(out_vel, 0, sizeof(float) * (2 * n + 2)); memset(out_vdes, 0, sizeof(float) * (2 * n + 2));
        memset(status, 0, n + 1);
    }else{
        ok = ok && navhip_agent_step_submit(ctx, &W, &O) == NAVHIP_OK;
        /* (the nav task would Task_AwaitEvent(EVENT_UPDATE_START) here, like the GL path :4212-4233) */
        if(ok) ok = navhip_agent_step_wait(ctx) == NAVHIP_OK;
    }
    const double t_stepped = hip_now();
    if(ok) {
        s_hip_stats[2]++;
        V.out_vel = out_vel; V.out_vdes = out_vdes; V.status = status;
        hip_for(hip_vel_scatter_range, nitems, &V);
        if(V.unsupported)
            ok = false;
        else {
            s_hip_stats[0] += V.sampled;
            /* the cases of nav.c:3483-3554 that need the planner or a repair build: the host samples (its builds
             * go through the binding and land in the pool) and steps these agents itself, in work-item order */
            qsort(V.fallback, V.n_fallback, sizeof(int32_t), cmp_i32);
            for(int k = 0; k < V.n_fallback; k++) {
                const int w = V.fallback[k];
                struct move_work_in *in = &s_move_work.in[w];
                const struct flock *fl = flock_for_ent(in->ent_uid);
                in->ent_des_v = M_NavDesiredPointSeekVelocity(gs->map, fl->dest_id, in->cp_ent.xz_pos, fl->target_xz);
                s_move_work.out[w].ent_des_v = in->ent_des_v;
                in->dyn_neighbs->size = 0; in->stat_neighbs->size = 0;
                move_velocity_work(w, w);
                s_hip_stats[1]++;
            }
        }
    }
    /* the state half of this tick may run on what this step left on the device (navhip_state_pass_resident): the
     * snapshot tables stay in the arena, and the step's outputs ARE the work items' velocities -- unless some agents
     * were re-stepped on the host above */
    s_hip_last.valid = ok && !s_hip_dry_run && V.n_fallback == 0;
    s_hip_last.S = S; s_hip_last.begin_idx = begin_idx; s_hip_last.end_idx = end_idx; s_hip_last.nwork = s_move_work.nwork;
    s_hip_last.lo = lo; s_hip_last.hi = hi;
    hip_snap_free(&S);
    s_hip_times[0] += t_filled - t_begin; s_hip_times[1] += t_stepped - t_filled;
    s_hip_times[2] += hip_now() - t_stepped; s_hip_times[3] += 1.0;
    return ok;
}

/* ---- the state-update half: fork_join_state_updates (movement.c:4196) -> move_update_task (:3496) ->
 * entity_compute_update (:2303) ------------------------------------------------------------------------
 * The decisions of the function run on the device for every work item at once, in ONE call (navhip_state_pass):
 * the heading gate (:2319-2336), the arrival arm of the state switch (:2441-2520: arrived() with its three nav
 * tests, the arrived-neighbour rule, the no-guidance wait; the garrison rule :2344), and the arms that flags, the
 * wait counter, the angle to target_dir and the distance to the target decide (formation members, ARRIVING_TO_CELL,
 * WAITING, TURNING, ENTER_ENTITY_RANGE).  Units of a flock with an active arrival zone follow in two more calls
 * (adjacent_settled_count + G_Arrival_ShouldSettle: move_hip_settle_work).  The host keeps the pose / interpolation
 * half of the patch and the units the device hands back (NAVHIP_SU_HOST: STATE_SURROUND_ENTITY, a facing within the
 * gate's margin of a tolerance, a nav layer other than most of the flock's, every unit at a rate below 20 Hz).
 * move_hip_state_work(begin, end) leaves next state + flags per work item; move_hip_update_work is
 * move_update_work (:3469) with the switch's outcome taken from there. */
int N_HIP_ClosestIslandTiles(struct nav_private *priv, enum nav_layer layer, vec3_t map_pos, vec2_t xz_dest,
                             int16_t *out_abs, int max_tiles);                    /* nav_hip.c */
uint32_t N_HIP_BlockersGeneration(void);                                          /* nav_hip.c */

static uint8_t *s_hip_su_state, *s_hip_su_flags;     /* [nwork] by work item */
static float   *s_hip_su_dest;                       /* [nwork][2] the surround arm's position (NAVHIP_SU_SURROUND_PREV / _DEST) */
static long     s_hip_surround_differ;               /* surround positions that differ from what the reference's switch stored */
long move_hip_surround_differ(void) { return s_hip_surround_differ; }
static size_t   s_hip_su_cap;
static long     s_hip_su_stats[3];                   /* decided on the device, left to the host, passes */
void move_hip_state_stats(long out[3]) { memcpy(out, s_hip_su_stats, sizeof(s_hip_su_stats)); }
/* where the last move_hip_state_work spent its time, seconds: {snapshot tables, per-unit inputs, the per-flock nav
 * queries, navhip_state_pass (transfers + kernels), the settle pass, scattering the answers to the work items} */
static double   s_hip_su_times[6];
void move_hip_state_times(double out[6]) { memcpy(out, s_hip_su_times, sizeof(s_hip_su_times)); }
/* the arrival overlay's settle rule on the device (navhip_arrival_settle): what it left in the unit's
 * struct arrival_unit_state, kept per work item so that move_hip_update_work can hold it against what the
 * reference's own G_Arrival_ShouldSettle leaves there (the harness cannot take the call out of
 * entity_compute_update; a maintainer applies the device's copy instead) */
struct hip_settle_chk { bool valid; uint8_t substate, anchored; int32_t stuck; vec2_t anchor; };
static struct hip_settle_chk *s_hip_settle_chk;      /* [nwork] by work item */
static int32_t *s_hip_wait_chk;                      /* [nwork][2] {a WAITING unit the device counted, its wait_ticks_left after} */
static long     s_hip_wait_differ;                   /* wait counters that differ from the reference's afterwards */
long move_hip_wait_differ(void) { return s_hip_wait_differ; }
static long     s_hip_settle_stats[4];               /* units decided by the device rule, of those settled, unit states
                                                        that differ from the reference's afterwards, units whose
                                                        heading gate the device left to the host */
void move_hip_settle_stats(long out[4]) { memcpy(out, s_hip_settle_stats, sizeof(s_hip_settle_stats)); }

struct hip_wrow { int32_t w, row; };                  /* a work item and its row in the arrays of its arm */
struct hip_state_pass { struct hip_snap *S; int begin_idx; float *new_vel, *vdes, *next_rot; uint8_t *skip, *zoned;
                        uint8_t *fstate, *wait_prev; int32_t *wait_ticks; float *ent_rot, *target_dir;
                        float *interp_from, *interp_step; int any_turning;
                        /* the work items of the two arms with per-unit host queries, of zoned units */
                        struct hip_wrow *range_items, *surround_items; int32_t *zoned_items; int n_range, n_surround, n_zoned;
                        /* sparse rows (the resident pass): a unit of TURNING / ENTER_ENTITY_RANGE / SURROUND_ENTITY takes the
                         * next row of its arm's arrays; the defaults of the other arms' rows are written with it */
                        bool sparse; int32_t *sparse_units; int n_sparse; int32_t *r_target, *r_row, *s_target; uint8_t *s_query;
                        const uint8_t *zone_active;       /* [nflocks][NAV_LAYER_MAX] the flock has an active arrival zone */
                        int lo, hi; };                    /* range of dense rows the work items cover */

#define HIP_PF_DIST 12
static void hip_state_items_range(int begin, int end, void *arg)
{
    struct hip_state_pass *T = arg;
    const struct hip_snap *S = T->S;
    int lo = INT32_MAX, hi = -1;
    for(int k = begin; k < end; k++) {
        const int w = T->begin_idx + k;
        const struct move_work_in *in = &s_move_work.in[w];
        const struct move_work_out *out = &s_move_work.out[w];
        const int i = hip_work_dense(S, w);
        /* (the movestate of a unit is a cache miss: the bucket of the item HIP_PF_DIST ahead is asked for now) */
        if(k + HIP_PF_DIST < end) {
            const int wa = w + HIP_PF_DIST, ia = s_hip_witem.idx[wa];
            if(ia >= 0 && ia < S->n) {
                const khint_t ca = s_hip_set.it_state[ia];
                if(ca < kh_end(s_entity_state_table)) {
                    const char *v = (const char*)&kh_value(s_entity_state_table, ca);
                    __builtin_prefetch(v); __builtin_prefetch(v + 64); __builtin_prefetch(&kh_key(s_entity_state_table, ca));
                }
            }
        }
        if(i < lo) lo = i;
        if(i > hi) hi = i;
        s_hip_settle_chk[w].valid = false;
        /* (movestate_get through the bucket position the snapshot fill cached for this row: no probe sequence) */
        const khiter_t mk = HIP_IT(s_entity_state_table, state, s_hip_set.it_state[i], in->ent_uid);
        const struct movestate *ms = mk != kh_end(s_entity_state_table) ? &kh_value(s_entity_state_table, mk) : movestate_get(in->ent_uid);
        /* the inputs of the heading gate (:2319-2336), decided on the device for the slab at once */
        if(T->new_vel) {            /* (the resident pass reads the step's own outputs on the device) */
            T->new_vel[2 * i] = out->ent_vel.x; T->new_vel[2 * i + 1] = out->ent_vel.z;
            T->vdes[2 * i] = out->ent_des_v.x; T->vdes[2 * i + 1] = out->ent_des_v.z;
        }
        T->next_rot[4 * i] = ms->next_rot.x; T->next_rot[4 * i + 1] = ms->next_rot.y;
        T->next_rot[4 * i + 2] = ms->next_rot.z; T->next_rot[4 * i + 3] = ms->next_rot.w;
        /* the flag / counter arms (formation member on the move, ARRIVING_TO_CELL, the wait timer):
         * navhip_state_update_aux after the arrival arm, which a member falls through to (:2439) */
        T->fstate[i] = (uint8_t)((in->fstate.fid != NULL_FID ? NAVHIP_FS_MEMBER : 0) | (in->fstate.assignment_ready ? NAVHIP_FS_READY : 0)
                     | (in->fstate.assigned_to_cell ? NAVHIP_FS_ASSIGNED : 0) | (in->fstate.in_range_of_cell ? NAVHIP_FS_IN_RANGE : 0)
                     | (in->fstate.arrived_at_cell ? NAVHIP_FS_ARRIVED : 0));
        T->wait_ticks[i] = ms->wait_ticks_left; T->wait_prev[i] = (uint8_t)ms->wait_prev;
        /* the three arms with arrays of their own: a row per entity, or (sparse) the next row of the list */
        int row = i;
        bool listed = false;
        if(ms->state == STATE_ENTER_ENTITY_RANGE || ms->state == STATE_SURROUND_ENTITY || ms->state == STATE_TURNING) {
            if(T->sparse) {
                row = __atomic_fetch_add(&T->n_sparse, 1, __ATOMIC_RELAXED);
                listed = true;
                T->sparse_units[row] = i;
                T->r_target[row] = -2; T->r_row[row] = 0; T->s_target[row] = -2; T->s_query[row] = 0;
            }
            if(ms->state == STATE_ENTER_ENTITY_RANGE) T->range_items[__atomic_fetch_add(&T->n_range, 1, __ATOMIC_RELAXED)] = (struct hip_wrow){w, row};
            if(ms->state == STATE_SURROUND_ENTITY)    T->surround_items[__atomic_fetch_add(&T->n_surround, 1, __ATOMIC_RELAXED)] = (struct hip_wrow){w, row};
        }
        if(listed && ms->state != STATE_TURNING)                         /* (a listed row nobody turns in) */
            memset(T->ent_rot + 4 * row, 0, sizeof(float) * 4), memset(T->target_dir + 4 * row, 0, sizeof(float) * 4);
        if(ms->state == STATE_TURNING) {                      /* :2606-2628: the end of the turn is the device's to see */
            __atomic_store_n(&T->any_turning, 1, __ATOMIC_RELAXED);
            const quat_t rot = Entity_GetRot(in->ent_uid);
            memcpy(T->ent_rot + 4 * row, &rot, sizeof(float) * 4);
            memcpy(T->target_dir + 4 * row, &ms->target_dir, sizeof(float) * 4);
        }
        /* a rate below 20 Hz: the switch tests the first interpolated position of an accepted move (:2368-2377) -- the
         * device makes it from movestate.next_pos and .step */
        if(T->interp_from) {
            T->interp_from[2 * i] = ms->next_pos.x; T->interp_from[2 * i + 1] = ms->next_pos.z;
            T->interp_step[i] = ms->step;
        }
        /* an active arrival group (:2443): move_hip_settle_work's, after the pass */
        T->skip[i] = 0;
        if(S->flock[i] >= 0) {
            T->skip[i] = T->zone_active[(size_t)S->flock[i] * NAV_LAYER_MAX + Entity_NavLayerWithRadius(S->flags[i], S->radius[i])];
            T->zoned[i] = T->skip[i];       /* (:2443-2451: the settle rule's arm, navhip_arrival_settle below) */
            if(T->skip[i] && (S->state[i] == STATE_MOVING || S->state[i] == STATE_MOVING_IN_FORMATION))
                T->zoned_items[__atomic_fetch_add(&T->n_zoned, 1, __ATOMIC_RELAXED)] = w;
        }
    }
    /* (the slab of dense rows the pass covers: merged once per range) */
    int cur = __atomic_load_n(&T->lo, __ATOMIC_RELAXED);
    while(lo < cur && !__atomic_compare_exchange_n(&T->lo, &cur, lo, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    cur = __atomic_load_n(&T->hi, __ATOMIC_RELAXED);
    while(hi > cur && !__atomic_compare_exchange_n(&T->hi, &cur, hi, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

/* The arm of the state switch for a unit whose flock has an active arrival zone (:2443-2451, then the
 * no-guidance wait :2508): adjacent_settled_count and G_Arrival_ShouldSettle for all of them in two device calls
 * (navhip_settled_count, navhip_arrival_settle).  A zone = one struct arrival_state, handed over as it is: its
 * slots, their fill ranks and the sorted tile keys of its footprint.  st / fl (by dense index) are overwritten
 * for the units decided here. */
static bool s_hip_settle_resident;                   /* the pass in front of the settle work ran on the resident snapshot */
static bool move_hip_settle_work(navhip_ctx *ctx, struct hip_snap *S, const navhip_world *W, const int32_t *zoned_items, int n_zoned,
                                 const uint8_t *gate, const float *new_pos, uint8_t *st, uint8_t *fl)
{
    const struct move_gamestate *gs = &s_move_work.gamestate;
    if(n_zoned == 0)
        return true;
    const size_t F = S->nflocks;
    /* one block for the pass: the (flock, layer) -> zone table, the zones, and 17 per-unit arrays */
    const size_t Q = (size_t)n_zoned;
    size_t bytes = 0;
#define TAKE(n_bytes) (bytes += (((size_t)(n_bytes)) + 63) & ~(size_t)63, bytes - ((((size_t)(n_bytes)) + 63) & ~(size_t)63))
    const size_t o_zone_of = TAKE(sizeof(int32_t) * (F * NAV_LAYER_MAX + 1)), o_zones = TAKE(sizeof(navhip_arrival_zone) * (Q + 1)),
                 o_zone_as = TAKE(sizeof(void*) * (Q + 1)), o_uid = TAKE(4 * Q), o_zone = TAKE(4 * Q), o_witem = TAKE(4 * Q),
                 o_nsettled = TAKE(4 * Q), o_stuck = TAKE(4 * Q), o_ostuck = TAKE(4 * Q), o_qpos = TAKE(8 * Q), o_sink = TAKE(8 * Q),
                 o_order = TAKE(8 * Q), o_anchor = TAKE(8 * Q), o_oanchor = TAKE(8 * Q), o_sub = TAKE(Q), o_sv = TAKE(Q),
                 o_anchored = TAKE(Q), o_osettle = TAKE(Q), o_osub = TAKE(Q), o_oanchored = TAKE(Q);
#undef TAKE
    char *blk = malloc(bytes);
    if(!blk)
        return false;
    int32_t *zone_of = (int32_t*)(blk + o_zone_of);                         /* (flock, layer) -> zone */
    for(size_t k = 0; k < F * NAV_LAYER_MAX; k++) zone_of[k] = -1;
    navhip_arrival_zone *zones = (navhip_arrival_zone*)(blk + o_zones);
    const struct arrival_state **zone_as = (const struct arrival_state**)(blk + o_zone_as);
    int32_t *uid = (int32_t*)(blk + o_uid), *zone = (int32_t*)(blk + o_zone), *witem = (int32_t*)(blk + o_witem);
    int32_t *nsettled = (int32_t*)(blk + o_nsettled), *stuck = (int32_t*)(blk + o_stuck), *o_stuck_ = (int32_t*)(blk + o_ostuck);
    float *q_pos = (float*)(blk + o_qpos), *sink = (float*)(blk + o_sink), *order = (float*)(blk + o_order);
    float *anchor = (float*)(blk + o_anchor), *o_anchor_ = (float*)(blk + o_oanchor);
    uint8_t *substate = (uint8_t*)(blk + o_sub), *sink_valid = (uint8_t*)(blk + o_sv), *anchored = (uint8_t*)(blk + o_anchored);
    uint8_t *o_settle = (uint8_t*)(blk + o_osettle), *o_substate = (uint8_t*)(blk + o_osub), *o_anchored_ = (uint8_t*)(blk + o_oanchored);
    int nz = 0, q = 0, n_slots = 0, n_keys = 0;
    for(int z = 0; z < n_zoned; z++) {
        const int w = zoned_items[z], i = s_hip_witem.idx[w];
        /* (MOVING / MOVING_IN_FORMATION units of zoned flocks, listed by the fill; of those, the ones the gate decided
         * and the pass left to this arm) */
        if((gate[i] & NAVHIP_GATE_HOST) || !(fl[i] & NAVHIP_SU_HOST))
            continue;
        const int f = S->flock[i];
        const enum nav_layer layer = Entity_NavLayerWithRadius(S->flags[i], S->radius[i]);
        int32_t *zi = &zone_of[(size_t)f * NAV_LAYER_MAX + layer];
        if(*zi < 0) {
            const struct arrival_state *as = G_ArrivalGroup_ForLayer(&vec_AT(&s_flocks, f).arrival, layer);
            *zi = nz;
            zone_as[nz] = as;
            zones[nz] = (navhip_arrival_zone){as->centre.x, as->centre.z, as->unit_radius, as->fill_frac, as->radius,
                (int32_t)as->layer, as->active_row, as->num_rows, n_slots, n_slots + as->num_slots, n_keys, n_keys + as->num_region};
            n_slots += as->num_slots; n_keys += as->num_region;
            nz++;
        }
        const khiter_t mk = HIP_IT(s_entity_state_table, state, s_hip_set.it_state[i], S->uids[i]);
        const struct movestate *ms = mk != kh_end(s_entity_state_table) ? &kh_value(s_entity_state_table, mk) : movestate_get(S->uids[i]);
        const struct arrival_unit_state *us = &ms->arrival;
        uid[q] = i; zone[q] = *zi; witem[q] = w;
        q_pos[2 * q] = new_pos[2 * i]; q_pos[2 * q + 1] = new_pos[2 * i + 1];
        substate[q] = (uint8_t)us->substate; sink_valid[q] = us->sink_valid;
        sink[2 * q] = us->sink.x; sink[2 * q + 1] = us->sink.z;
        order[2 * q] = us->order_pos.x; order[2 * q + 1] = us->order_pos.z;
        anchor[2 * q] = us->progress_anchor.x; anchor[2 * q + 1] = us->progress_anchor.z;
        anchored[q] = us->progress_anchored; stuck[q] = us->stuck;
        q++;
    }
    const int nq = q;
    if(nq == 0) {
        free(blk);
        return true;
    }
    float *slots = malloc(sizeof(float) * 2 * (n_slots + 1));
    int32_t *ring = malloc(sizeof(int32_t) * (n_slots + 1));
    uint64_t *keys = malloc(sizeof(uint64_t) * (n_keys + 1));
    for(int z = 0; z < nz; z++) {
        const struct arrival_state *as = zone_as[z];
        memcpy(slots + 2 * zones[z].slot_begin, as->slots, sizeof(vec2_t) * as->num_slots);
        memcpy(ring + zones[z].slot_begin, as->slot_ring, sizeof(int) * as->num_slots);
        memcpy(keys + zones[z].key_begin, as->region_keys, sizeof(uint64_t) * as->num_region);
    }
    /* behind a state pass on the velocity pass's resident snapshot: the count and the rule in ONE call on that snapshot
     * (in.nsettled NULL: counted on the device, out.nsettled says which units stay the host's) */
    navhip_settle_in in = {nz, nq, zones, slots, ring, keys, uid, zone, q_pos, NULL, substate, sink_valid, sink, order,
                           anchor, anchored, stuck};
    navhip_settle_out out = {o_settle, o_substate, o_anchor_, o_anchored_, o_stuck_, nsettled};
    bool ok = s_hip_settle_resident && navhip_arrival_settle_resident(ctx, W, &in, &out) == NAVHIP_OK;
    if(!ok) {
        in.nsettled = nsettled; out.nsettled = NULL;
        ok = navhip_settled_count(ctx, W, nq, uid, nsettled) == NAVHIP_OK && navhip_arrival_settle(ctx, W, &in, &out) == NAVHIP_OK;
    }
    for(q = 0; ok && q < nq; q++) {
        const int i = uid[q];
        if(nsettled[q] < 0)
            continue;                                   /* (a unit wider than the device's query: the host's arm) */
        const enum nav_layer layer = Entity_NavLayerWithRadius(S->flags[i], S->radius[i]);
        const vec2_t np = {q_pos[2 * q], q_pos[2 * q + 1]}, vd = s_move_work.out[witem[q]].ent_des_v;
        s_hip_settle_stats[0]++;
        st[i] = S->state[i]; fl[i] = 0;
        if(!M_NavPositionPathable(gs->map, layer, np))
            continue;                                   /* :2437: stuck where it is, in the state it was */
        if(o_settle[q]) {
            st[i] = STATE_ARRIVED; fl[i] = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK;
            s_hip_settle_stats[1]++;
        }else if(PFM_Vec2_Len((vec2_t*)&vd) < EPSILON) {
            st[i] = STATE_WAITING; fl[i] = NAVHIP_SU_SET_STATE | NAVHIP_SU_BLOCK;       /* :2508 */
        }
        s_hip_settle_chk[witem[q]] = (struct hip_settle_chk){true, o_substate[q], o_anchored_[q], o_stuck_[q],
                                                            {o_anchor_[2 * q], o_anchor_[2 * q + 1]}};
    }
    free(blk); free(slots); free(ring); free(keys);
    return ok;
}

/* is p, to the bit, the centre the device computes for some nav tile: (map_x - col * 4, map_z + row * 4)? */
static bool hip_is_tile_centre(vec3_t map_pos, vec2_t p)
{
    const float c = roundf((map_pos.x - p.x) / 4.0f), r = roundf((p.z - map_pos.z) / 4.0f);
    return map_pos.x - c * 4.0f == p.x && map_pos.z + r * 4.0f == p.z;
}

/* the unit-query answers of the surround units, forked over the host threads like the reference forks
 * entity_compute_update -- the queries are the reference's own, read-only on the nav data */
struct hip_surround_q { const struct hip_snap *S; const struct hip_wrow *items; int32_t *s_target; uint8_t *s_query; float *s_tprev, *s_nprev, *s_dest; };
static void hip_surround_range(int begin, int end, void *arg)
{
    struct hip_surround_q *Q = arg;
    const struct hip_snap *S = Q->S;
    const struct move_gamestate *gs = &s_move_work.gamestate;
    for(int k = begin; k < end; k++) {
        const int w = Q->items[k].w, i = s_hip_witem.idx[w], r = Q->items[k].row;      /* (r: the unit's row in the arm's arrays) */
        const uint32_t uid = S->uids[i];
        const struct movestate *ms = movestate_get(uid);
        Q->s_tprev[2 * r] = ms->surround_target_prev.x; Q->s_tprev[2 * r + 1] = ms->surround_target_prev.z;
        Q->s_nprev[2 * r] = ms->surround_nearest_prev.x; Q->s_nprev[2 * r + 1] = ms->surround_nearest_prev.z;
        if(ms->surround_target_uid == NULL_UID) { Q->s_target[r] = -1; continue; }
        if(!entity_exists(ms->surround_target_uid)
        || M_NavObjAdjacentFrom(gs->map, uid, ms->surround_target_uid, &s_move_work.unit_query_ctx)) {
            Q->s_target[r] = -1; Q->s_query[r] = NAVHIP_SQ_ADJACENT;        /* (-> ARRIVED either way, :2518-2525) */
            continue;
        }
        khiter_t it = kh_get(id, S->dense, ms->surround_target_uid);
        if(it == kh_end(S->dense)) {
            Q->s_target[r] = -2;
            continue;                                                     /* (a target outside the snapshot: the host's) */
        }
        Q->s_target[r] = (int32_t)kh_value(S->dense, it);
        const vec2_t tp = {S->pos[2 * Q->s_target[r]], S->pos[2 * Q->s_target[r] + 1]};
        vec2_t delta, dest;
        PFM_Vec2_Sub((vec2_t*)&tp, (vec2_t*)&ms->surround_target_prev, &delta);
        if(!(PFM_Vec2_Len(&delta) > EPSILON || PFM_Vec2_Len(&ms->velocity) < EPSILON))
            continue;                                                     /* (the query does not run this tick) */
        const enum nav_layer layer = Entity_NavLayerWithRadius(S->flags[i], S->radius[i]);
        const vec2_t pos = {S->pos[2 * i], S->pos[2 * i + 1]}, vel = s_move_work.out[w].ent_vel;
        vec2_t from[2] = {pos, pos};
        PFM_Vec2_Add((vec2_t*)&pos, (vec2_t*)&vel, &from[0]);
        for(int c = 0; c < 2; c++) {
            if(c == 1 && from[0].x == from[1].x && from[0].z == from[1].z) {      /* (zero velocity: one query) */
                if(Q->s_query[r] & NAVHIP_SQ_HAS_DEST_0) {
                    Q->s_query[r] |= NAVHIP_SQ_HAS_DEST_1; Q->s_dest[4 * r + 2] = Q->s_dest[4 * r]; Q->s_dest[4 * r + 3] = Q->s_dest[4 * r + 1];
                }
                break;
            }
            if(M_NavClosestReachableAdjacentPosFrom(gs->map, layer, from[c], ms->surround_target_uid, &s_move_work.unit_query_ctx, &dest)) {
                Q->s_query[r] |= (uint8_t)(NAVHIP_SQ_HAS_DEST_0 << c);
                Q->s_dest[4 * r + 2 * c] = dest.x; Q->s_dest[4 * r + 2 * c + 1] = dest.z;
            }
        }
    }
}

/* the per-flock answers of arrived()'s two destination-only queries, kept between ticks */
struct hip_flock_q { bool valid, has_near; uint32_t epoch, blk_gen; const void *map; int layer, ntiles, per; float tx, tz, nx, nz; int16_t *tiles;
                     bool layer_valid; uint32_t layer_set_epoch, layer_attr_epoch; int layer_members, majority_layer; };
static struct hip_flock_q *s_hip_flock_q; static size_t s_hip_flock_q_cap;
static struct hip_flock_q *hip_flock_q_slot(size_t f, int per)
{
    if(f >= s_hip_flock_q_cap) {
        const size_t cap = f + 16;
        struct hip_flock_q *grown = realloc(s_hip_flock_q, sizeof(struct hip_flock_q) * cap);
        if(!grown)
            return NULL;                                /* (the caller asks the queries without keeping them) */
        s_hip_flock_q = grown;
        for(size_t k = s_hip_flock_q_cap; k < cap; k++) {
            s_hip_flock_q[k].valid = false; s_hip_flock_q[k].layer_valid = false;
            s_hip_flock_q[k].tiles = NULL; s_hip_flock_q[k].per = 0;
        }
        s_hip_flock_q_cap = cap;
    }
    struct hip_flock_q *C = &s_hip_flock_q[f];
    if(C->per < per) {                                  /* (the tile list's room follows the caller's `per`) */
        int16_t *t = realloc(C->tiles, sizeof(int16_t) * 2 * per);
        if(!t)
            return NULL;
        C->tiles = t; C->per = per; C->valid = false;
    }
    return C;
}

struct hip_state_scatter { int begin_idx; const uint8_t *st, *fl, *gate, *state; const int32_t *wait_after; const float *dense_dest; long host, gate_host; };
static void hip_state_scatter_range(int begin, int end, void *arg)
{
    struct hip_state_scatter *X = arg;
    long host = 0, gate_host = 0;
    for(int k = begin; k < end; k++) {
        const int w = X->begin_idx + k, i = s_hip_witem.idx[w];
        s_hip_su_state[w] = X->st[i];
        s_hip_su_flags[w] = X->fl[i];
        host += (X->fl[i] & NAVHIP_SU_HOST) != 0;
        /* a unit whose facing is within the device's margin of a tolerance came back NAVHIP_SU_HOST: the host's own
         * entity_compute_update answers for it (move_hip_update_work) */
        gate_host += (X->gate[i] & NAVHIP_GATE_HOST) != 0;
        s_hip_wait_chk[2 * w] = X->state[i] == STATE_WAITING;
        s_hip_wait_chk[2 * w + 1] = X->wait_after[i];
        /* (the surround arm's position: per entity from the host-buffer pass, per listed unit -- written afterwards --
         * from the resident one) */
        s_hip_su_dest[2 * w] = X->dense_dest ? X->dense_dest[2 * i] : 0.0f;
        s_hip_su_dest[2 * w + 1] = X->dense_dest ? X->dense_dest[2 * i + 1] : 0.0f;
    }
    __atomic_fetch_add(&X->host, host, __ATOMIC_RELAXED);
    __atomic_fetch_add(&X->gate_host, gate_host, __ATOMIC_RELAXED);
}

static int cmp_wrow(const void *a, const void *b)
{
    const int32_t x = ((const struct hip_wrow*)a)->w, y = ((const struct hip_wrow*)b)->w;
    return (x > y) - (x < y);
}

static bool move_hip_state_work(int begin_idx, int end_idx)
{
    navhip_ctx *ctx = N_HIP_Ctx();
    if(!ctx || end_idx < begin_idx)
        return false;
    const struct move_gamestate *gs = &s_move_work.gamestate;
    struct hip_snap S;
    double t_mark = hip_now(), t_now;
#define HIP_SU_LAP(k) (t_now = hip_now(), s_hip_su_times[k] = t_now - t_mark, t_mark = t_now)
#define HIP_SU_ADD(k) (t_now = hip_now(), s_hip_su_times[k] += t_now - t_mark, t_mark = t_now)
    /* resident: this tick's velocity pass has just run over the same work items -- its snapshot tables are still in the
     * arena and on the device, its outputs are the work items' velocities */
    bool resident = s_hip_state_resident && s_hip_last.valid && s_hip_last.begin_idx == begin_idx && s_hip_last.end_idx == end_idx
                 && s_hip_last.nwork == s_move_work.nwork && s_hip_last.S.n == (int)kh_size(gs->positions)
                 && hip_pin_reserve((size_t)s_hip_last.S.n);
    s_hip_last.valid = false;
    if(resident) S = s_hip_last.S;
    else         hip_snap_fill(&S);
    HIP_SU_LAP(0);
    const int n = S.n, nitems = end_idx - begin_idx + 1;
    if(s_hip_su_cap < s_move_work.nwork) {
        s_hip_su_cap = s_move_work.nwork;
        s_hip_su_state = realloc(s_hip_su_state, s_hip_su_cap);
        s_hip_su_flags = realloc(s_hip_su_flags, s_hip_su_cap);
        s_hip_settle_chk = realloc(s_hip_settle_chk, sizeof(struct hip_settle_chk) * (s_hip_su_cap + 1));
        s_hip_wait_chk = realloc(s_hip_wait_chk, sizeof(int32_t) * 2 * (s_hip_su_cap + 1));
        s_hip_su_dest = realloc(s_hip_su_dest, sizeof(float) * 2 * (s_hip_su_cap + 1));
    }
    /* (resident: the arrays the device reads or writes per unit live in page-locked memory and are transferred in
     * place; new_vel / vdes do not travel at all) */
    float *new_pos = resident ? s_hip_pin.new_pos : hip_arena(sizeof(float) * (2 * n + 2));
    float *vdes = resident ? NULL : hip_arena(sizeof(float) * (2 * n + 2)), *new_vel = resident ? NULL : hip_arena(sizeof(float) * (2 * n + 2));
    float *next_rot = resident ? s_hip_pin.next_rot : hip_arena(sizeof(float) * (4 * n + 4));
    float *gate_vel = resident ? NULL : hip_arena(sizeof(float) * (2 * n + 2));
    uint8_t *skip = resident ? s_hip_pin.skip : hip_arena(n + 1), *zoned = hip_arena(n + 1), *gate = resident ? s_hip_pin.gate : hip_arena(n + 1);
    uint8_t *fstate = resident ? s_hip_pin.fstate : hip_arena(n + 1), *wait_prev = resident ? s_hip_pin.wait_prev : hip_arena(n + 1);
    int32_t *wait_ticks = resident ? s_hip_pin.wait_ticks : hip_arena(sizeof(int32_t) * (n + 1));
    int32_t *wait_after = resident ? s_hip_pin.wait_after : hip_arena(sizeof(int32_t) * (n + 1));
    /* (rows of the slab without a work item are computed and dropped: they only need defined inputs -- whatever the
     * page-locked arrays of the resident pass held last tick will do, the pageable ones are cleared) */
    if(!resident) { memset(fstate, 0, n + 1); memset(wait_prev, 0, n + 1); memset(wait_ticks, 0, sizeof(int32_t) * (n + 1)); }
    if(!resident) { memset(skip, 0, n + 1); memset(next_rot, 0, sizeof(float) * (4 * n + 4)); }
    memset(zoned, 0, n + 1);
    if(!resident) {
        memset(new_pos, 0, sizeof(float) * (2 * n + 2)); memset(vdes, 0, sizeof(float) * (2 * n + 2));
        memset(new_vel, 0, sizeof(float) * (2 * n + 2));
    }
    const bool sub20 = (20 / hz_count(s_move_work.hz)) > 1;
    float *interp_from = sub20 ? hip_arena(sizeof(float) * (2 * n + 2)) : NULL, *interp_step = sub20 ? hip_arena(sizeof(float) * (n + 1)) : NULL;
    if(sub20) { memset(interp_from, 0, sizeof(float) * (2 * n + 2)); memset(interp_step, 0, sizeof(float) * (n + 1)); }
    /* The arrays of the three arms that few units take (TURNING, ENTER_ENTITY_RANGE, SURROUND_ENTITY), from the arena and
     * not cleared: the device reads the rows of such units only.  The resident pass hands them over SPARSE -- a row per
     * listed unit (navhip_state_aux_in.sparse_units) --, the host-buffer pass a row per entity. */
    const bool sparse = resident;
    const size_t R = sparse ? (size_t)nitems : (size_t)n;          /* rows an arm's array can need */
    float *ent_rot = hip_arena(sizeof(float) * (4 * R + 4)), *target_dir = hip_arena(sizeof(float) * (4 * R + 4));
    int32_t *r_target = hip_arena(sizeof(int32_t) * (R + 1)), *r_row = hip_arena(sizeof(int32_t) * (R + 1));
    float *r_range = hip_arena(sizeof(float) * (R + 1)), *r_prev = hip_arena(sizeof(float) * (2 * R + 2));
    int32_t *s_target = hip_arena(sizeof(int32_t) * (R + 1)); uint8_t *s_query = hip_arena(R + 1);
    float *s_tprev = hip_arena(sizeof(float) * (2 * R + 2)), *s_nprev = hip_arena(sizeof(float) * (2 * R + 2));
    float *s_dest = hip_arena(sizeof(float) * (4 * R + 4)), *s_out = hip_arena(sizeof(float) * (2 * R + 2));
    int32_t *sparse_units = sparse ? hip_arena(sizeof(int32_t) * (R + 1)) : NULL;
    /* which (flock, layer) has an active arrival zone (:2443): asked once per pass, not once per unit */
    const size_t F = S.nflocks;
    uint8_t *zone_active = hip_arena(F * NAV_LAYER_MAX + 1);
    for(size_t f = 0; f < F; f++)
        for(int l = 0; l < NAV_LAYER_MAX; l++) {
            struct arrival_state *as = G_ArrivalGroup_ForLayer(&vec_AT(&s_flocks, f).arrival, (enum nav_layer)l);
            zone_active[f * NAV_LAYER_MAX + l] = as && G_Arrival_IsActive(as);
        }
    hip_work_dense_prepare();
    struct hip_state_pass T = {&S, begin_idx, new_vel, vdes, next_rot, skip, zoned, fstate, wait_prev, wait_ticks, ent_rot, target_dir,
                               interp_from, interp_step, 0,
                               hip_arena(sizeof(struct hip_wrow) * (nitems + 1)), hip_arena(sizeof(struct hip_wrow) * (nitems + 1)),
                               hip_arena(sizeof(int32_t) * (nitems + 1)), 0, 0, 0,
                               sparse, sparse_units, 0, r_target, r_row, s_target, s_query, zone_active, INT32_MAX, -1};
    hip_for(hip_state_items_range, nitems, &T);
    const bool any_turning = T.any_turning != 0;
    /* (the fill's threads append in any order: work-item order again, so that a pass is reproducible) */
    qsort(T.range_items, T.n_range, sizeof(struct hip_wrow), cmp_wrow);
    qsort(T.surround_items, T.n_surround, sizeof(struct hip_wrow), cmp_wrow);
    qsort(T.zoned_items, T.n_zoned, sizeof(int32_t), cmp_i32);
    const int lo = T.lo, hi = T.hi;
    HIP_SU_LAP(1);
    /* the two destination-only queries of arrived() (:2170), once per flock for the nav layer most of its
     * members path on (units of another layer come back as NAVHIP_SU_HOST) */
    uint8_t *flayer = hip_arena(F + 1);
    float   *nearest = hip_arena(sizeof(float) * 2 * (F + 1));
    int32_t *toff = hip_arena(sizeof(int32_t) * (F + 2));
    memset(flayer, 0, F + 1); memset(toff, 0, sizeof(int32_t) * (F + 2));
    const int per = FIELD_RES_R * 2 + FIELD_RES_C * 2;
    int16_t *tiles = hip_arena(sizeof(int16_t) * 2 * per * (F + 1));
    vec3_t map_pos = move_hip_map_pos(gs->map);
    for(size_t f = 0; f < F; f++) {
        const struct flock *fl = &vec_AT(&s_flocks, f);
        nearest[2 * f] = NAN; nearest[2 * f + 1] = 0.0f;
        toff[f + 1] = toff[f];
        if(S.flock_offsets[f + 1] == S.flock_offsets[f])
            continue;
        /* (the layer most members path on: a property of the flock tables and the units' attributes, counted again when
         * either changes -- the entity set or move_hip_attrs_changed) */
        struct hip_flock_q scratch_q = {0}, *C = hip_flock_q_slot(f, per);
        int16_t scratch_tiles[2 * (FIELD_RES_R * 2 + FIELD_RES_C * 2)];
        if(!C) { C = &scratch_q; C->tiles = scratch_tiles; C->per = per; }      /* (no memory for the cache: ask, do not keep) */
        if(!(C->layer_valid && C->layer_set_epoch == s_hip_set_epoch && C->layer_attr_epoch == s_hip_attr_epoch
             && C->layer_members == S.flock_offsets[f + 1] - S.flock_offsets[f])) {
            int per_layer[NAV_LAYER_MAX] = {0}, best = 0;
            for(int m = S.flock_offsets[f]; m < S.flock_offsets[f + 1]; m++) {
                const int i = S.flock_members[m];
                per_layer[Entity_NavLayerWithRadius(S.flags[i], S.radius[i])]++;
            }
            for(int l = 1; l < NAV_LAYER_MAX; l++)
                if(per_layer[l] > per_layer[best]) best = l;
            C->layer_valid = true; C->layer_set_epoch = s_hip_set_epoch; C->layer_attr_epoch = s_hip_attr_epoch;
            C->layer_members = S.flock_offsets[f + 1] - S.flock_offsets[f]; C->majority_layer = best;
        }
        const enum nav_layer layer = (enum nav_layer)C->majority_layer;
        flayer[f] = (uint8_t)layer;
        /* The two queries depend on the destination, the layer, the terrain AND THE BLOCKERS: N_ClosestPathable (nav.c:4126)
         * goes through n_tile_blocked (:235: cost_base == COST_IMPASSABLE or blockers > 0), and the island tiles are taken
         * with ignore_blockers = false (:4725) -- a unit that settles on or next to the destination changes both, which
         * is the arrival case itself.  So an answer is kept only while the flock's target, the layer, the map's nav data
         * (move_hip_attrs_changed / N_HIP_SyncLayer bump s_hip_attr_epoch) and the BLOCKER GENERATION stand
         * (N_HIP_BlockersGeneration: bumped by every N_BlockersIncref / N_BlockersDecref, nav_hip.c): on a tick in which
         * nothing was blocked or unblocked anywhere the answers are the last tick's, otherwise they are asked again. */
        const uint32_t blk_gen = N_HIP_BlockersGeneration();
        if(!(C->valid && C->epoch == s_hip_attr_epoch && C->blk_gen == blk_gen && C->map == (const void*)gs->map && C->layer == (int)layer
             && C->tx == fl->target_xz.x && C->tz == fl->target_xz.z)) {
            vec2_t near_xz;
            C->has_near = M_NavClosestPathable(gs->map, layer, fl->target_xz, &near_xz);
            C->nx = C->has_near ? near_xz.x : 0.0f; C->nz = C->has_near ? near_xz.z : 0.0f;
            C->ntiles = N_HIP_ClosestIslandTiles(move_hip_nav_private(gs->map), layer, map_pos, fl->target_xz, C->tiles, per);
            C->valid = true; C->epoch = s_hip_attr_epoch; C->blk_gen = blk_gen; C->map = (const void*)gs->map; C->layer = (int)layer;
            C->tx = fl->target_xz.x; C->tz = fl->target_xz.z;
        }
        if(C->has_near) { nearest[2 * f] = C->nx; nearest[2 * f + 1] = C->nz; }
        memcpy(tiles + 2 * toff[f], C->tiles, sizeof(int16_t) * 2 * C->ntiles);
        toff[f + 1] += C->ntiles;
    }
    HIP_SU_LAP(2);
    navhip_world W;
    hip_snap_world(&S, &W);
    W.work_begin = lo; W.work_end = hi + 1;
    uint8_t *st = resident ? s_hip_pin.st : hip_arena(n + 1), *fl = resident ? s_hip_pin.fl : hip_arena(n + 1);
    /* ONE call for the pass (navhip_state_pass): the heading gate of every unit (:2319-2336) -> the arrival arm on the
     * positions the gate leaves (navhip_state_update) -> the arms that flags, the wait counter, the angle to target_dir and
     * the distance to the target decide (navhip_state_update_aux); the snapshot travels once */
    navhip_state_pass_in pin;
    memset(&pin, 0, sizeof(pin));
    pin.gate = (navhip_gate_in){next_rot, new_vel, vdes, interp_from, interp_step};
    pin.state = (navhip_state_in){NULL, NULL, skip, flayer, nearest, toff, tiles};
    int32_t *r_off = NULL; int16_t *r_tiles = NULL;
    pin.aux.fstate = fstate; pin.aux.wait_ticks_left = wait_ticks; pin.aux.wait_prev = wait_prev;
    if(any_turning) {                                     /* (else 32 bytes per row that nobody would read) */
        pin.aux.ent_rot = ent_rot; pin.aux.target_dir = target_dir;
    }
    if(sparse && T.n_sparse > 0) { pin.aux.sparse_units = sparse_units; pin.aux.n_sparse = T.n_sparse; }
    /* STATE_ENTER_ENTITY_RANGE (:2569-2604): the target's row in the snapshot, the range, where the target stood when
     * the path was requested, and -- per such unit -- the closest island tiles of the target's position on the unit's
     * layer (the first half of N_IsMaximallyClose, as for the flocks' destinations above) */
    const int n_range = T.n_range;
    if(n_range > 0) {
        r_off = calloc(n_range + 1, sizeof(int32_t));
        r_tiles = malloc(sizeof(int16_t) * 2 * per * n_range);
        if(!sparse) {                 /* (a row per entity: the library checks every row's target, hence "the host's") */
            memset(r_target, 0xfe, sizeof(int32_t) * n);            /* (any value below -1 = "the host's") */
            memset(r_row, 0, sizeof(int32_t) * n);
        }
        int row = 0;
        for(int k = 0; k < n_range; k++) {
            const int w = T.range_items[k].w, i = s_hip_witem.idx[w], r = T.range_items[k].row;
            const struct movestate *ms = movestate_get(S.uids[i]);
            r_off[row + 1] = r_off[row];
            r_row[r] = row;
            r_range[r] = ms->target_range;
            r_prev[2 * r] = ms->target_prev_pos.x; r_prev[2 * r + 1] = ms->target_prev_pos.z;
            if(ms->surround_target_uid == NULL_UID) {
                r_target[r] = -1;
            }else{
                khiter_t kt = kh_get(id, S.dense, ms->surround_target_uid);
                if(kt != kh_end(S.dense)) {
                    r_target[r] = (int32_t)kh_value(S.dense, kt);
                    /* N_IsMaximallyClose(new_pos, target, 0.0f) (:2585) holds only where the tested position IS the
                     * centre of one of the target's closest island tiles: the 31-us island query is only made for a
                     * unit one of whose candidate positions -- pos + new velocity, pos (halted by the gate) -- is a
                     * tile centre to the bit; any other unit gets an empty row (at rates below 20 Hz the position is
                     * the device's interpolation: every unit is asked for) */
                    const vec2_t pos = {S.pos[2 * i], S.pos[2 * i + 1]}, vel = s_move_work.out[w].ent_vel;
                    vec2_t moved_to;
                    PFM_Vec2_Add((vec2_t*)&pos, (vec2_t*)&vel, &moved_to);
                    if(sub20 || hip_is_tile_centre(map_pos, pos) || hip_is_tile_centre(map_pos, moved_to)) {
                        const vec2_t tp = {S.pos[2 * r_target[r]], S.pos[2 * r_target[r] + 1]};
                        r_off[row + 1] += N_HIP_ClosestIslandTiles(move_hip_nav_private(gs->map),
                            Entity_NavLayerWithRadius(S.flags[i], S.radius[i]), map_pos, tp, r_tiles + 2 * r_off[row], per);
                    }
                }else
                    r_target[r] = -2;                               /* (a target outside the snapshot: the host's) */
            }
            row++;
        }
        pin.aux.range_target = r_target; pin.aux.target_range = r_range; pin.aux.target_prev_xz = r_prev;
        pin.aux.range_tiles_row = r_row; pin.aux.range_tiles_off = r_off; pin.aux.range_tiles = r_tiles;
        pin.aux.n_range_rows = n_range;
    }
    /* STATE_SURROUND_ENTITY (:2509-2567) at 20 Hz: the switch runs on the device, the two queries on the unit-query
     * context stay here -- whether the unit already touches its target (or the target is gone), and, for the units that
     * reach the query (:2532-2534), the closest reachable position next to the target from both positions the tick can
     * test: pos + new velocity, and pos (the heading gate halts the unit) */
    const int n_surround = sub20 ? 0 : T.n_surround;
    if(n_surround > 0) {
        if(!sparse) {
            memset(s_target, 0xfe, sizeof(int32_t) * n);              /* (any value below -1 = "the host's") */
            memset(s_query, 0, n + 1); memset(s_out, 0, sizeof(float) * (2 * n + 2));
        }
        struct hip_surround_q Q = {&S, T.surround_items, s_target, s_query, s_tprev, s_nprev, s_dest};
        hip_for_min(hip_surround_range, n_surround, &Q, 16);
        pin.aux.surround_target = s_target; pin.aux.surround_query = s_query; pin.aux.surround_target_prev_xz = s_tprev;
        pin.aux.surround_nearest_prev_xz = s_nprev; pin.aux.surround_dest_xz = s_dest; pin.aux.out_surround_dest_xz = s_out;
    }
    navhip_state_pass_out pout = {st, fl, gate, new_pos, gate_vel, wait_after};
    HIP_SU_ADD(2);                  /* (the enter-range and surround inputs count as queries too) */
    bool ok = hi >= lo;
    if(ok && resident && navhip_state_pass_resident(ctx, &pin, &pout) != NAVHIP_OK) {
        /* (the step's arrays are gone -- somebody used the context in between: the host-buffer pass, from the start) */
        free(r_off); free(r_tiles);
        hip_snap_free(&S);
        s_hip_state_resident = false;
        ok = move_hip_state_work(begin_idx, end_idx);
        s_hip_state_resident = true;
        return ok;
    }else if(ok && !resident)
        ok = navhip_state_pass(ctx, &W, &pin, &pout) == NAVHIP_OK;
    s_hip_resident_passes += ok && resident;
    s_hip_settle_resident = ok && resident;
    HIP_SU_LAP(3);
    free(r_off); free(r_tiles);
    /* units of flocks with an active arrival zone, skipped above: the settle rule on the positions the gate left */
    if(ok)
        ok = move_hip_settle_work(ctx, &S, &W, T.zoned_items, T.n_zoned, gate, new_pos, st, fl);
    HIP_SU_LAP(4);
    if(ok) {
        s_hip_su_stats[2]++;
        struct hip_state_scatter X = {begin_idx, st, fl, gate, S.state, wait_after, (n_surround > 0 && !sparse) ? s_out : NULL, 0, 0};
        hip_for(hip_state_scatter_range, nitems, &X);
        s_hip_su_stats[1] += X.host; s_hip_su_stats[0] += nitems - X.host;
        s_hip_settle_stats[3] += X.gate_host;
        if(sparse)                                             /* (the surround positions came back per listed unit) */
            for(int k = 0; k < n_surround; k++) {
                const struct hip_wrow it = T.surround_items[k];
                s_hip_su_dest[2 * it.w] = s_out[2 * it.row]; s_hip_su_dest[2 * it.w + 1] = s_out[2 * it.row + 1];
            }
    }else
        for(int w = begin_idx; w <= end_idx; w++) {
            s_hip_wait_chk[2 * w] = 0; s_hip_su_flags[w] = NAVHIP_SU_HOST; s_hip_su_state[w] = 0;
        }
    hip_snap_free(&S);
    HIP_SU_LAP(5);
#undef HIP_SU_LAP
#undef HIP_SU_ADD
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
        /* (the reference's own switch has just counted the wait down in movestate: the device's count is held against it;
         * a maintainer stores the device's) */
        if(s_hip_wait_chk && s_hip_wait_chk[2 * w] && movestate_get(out->ent_uid)->wait_ticks_left != s_hip_wait_chk[2 * w + 1])
            s_hip_wait_differ++;
        if(s_hip_settle_chk && s_hip_settle_chk[w].valid) {
            /* the reference's G_Arrival_ShouldSettle has just run inside entity_compute_update: what it left in
             * the unit's arrival state is what the device's rule returned for it */
            const struct arrival_unit_state *us = &movestate_get(out->ent_uid)->arrival;
            const struct hip_settle_chk *c = &s_hip_settle_chk[w];
            if((uint8_t)us->substate != c->substate || (uint8_t)us->progress_anchored != c->anchored || us->stuck != c->stuck
            || us->progress_anchor.x != c->anchor.x || us->progress_anchor.z != c->anchor.z)
                s_hip_settle_stats[2]++;
        }
        if(s_hip_su_flags[w] & NAVHIP_SU_HOST)
            continue;
        if(s_hip_su_flags[w] & NAVHIP_SU_SURROUND_PREV) {
            /* (the reference's own switch has just stored surround_target_prev / surround_nearest_prev in movestate, :2551:
             * the device's position is held against it; a maintainer stores the device's) */
            const struct movestate *ms = movestate_get(out->ent_uid);
            if(ms->surround_nearest_prev.x != s_hip_su_dest[2 * w] || ms->surround_nearest_prev.z != s_hip_su_dest[2 * w + 1])
                s_hip_surround_differ++;
        }
        out->patch.flags = (enum movestate_flags)(out->patch.flags & ~(UPDATE_SET_STATE | UPDATE_SET_MOVING | UPDATE_SET_TARGET_DIR
                                                                        | UPDATE_SET_DEST | UPDATE_SET_TARGET_PREV));
        if(s_hip_su_flags[w] & NAVHIP_SU_SURROUND_DEST) {               /* a new position next to the target, :2555-2560 */
            out->patch.flags = (enum movestate_flags)(out->patch.flags | UPDATE_SET_DEST);
            out->patch.next_dest = (vec2_t){s_hip_su_dest[2 * w], s_hip_su_dest[2 * w + 1]};
            out->patch.next_attack = false;
        }
        if(s_hip_su_flags[w] & NAVHIP_SU_SET_STATE) {
            out->patch.flags = (enum movestate_flags)(out->patch.flags | UPDATE_SET_STATE);
            out->patch.next_state = (enum move_state)s_hip_su_state[w];
            out->patch.next_block = (s_hip_su_flags[w] & NAVHIP_SU_BLOCK) != 0;
        }
        if(s_hip_su_flags[w] & NAVHIP_SU_SET_MOVING) {                  /* the wait ran out, :2641 */
            out->patch.flags = (enum movestate_flags)(out->patch.flags | UPDATE_SET_MOVING);
            out->patch.next_state = (enum move_state)s_hip_su_state[w];
        }
        if(s_hip_su_flags[w] & NAVHIP_SU_SET_DEST) {                    /* the target has moved on, :2597-2602 */
            const vec2_t xz_target = G_Pos_GetXZFrom(s_move_work.gamestate.positions, movestate_get(out->ent_uid)->surround_target_uid);
            out->patch.flags = (enum movestate_flags)(out->patch.flags | UPDATE_SET_DEST | UPDATE_SET_TARGET_PREV);
            out->patch.next_dest = xz_target;
            out->patch.next_attack = false;
            out->patch.next_target_prev = xz_target;
        }
        if(s_hip_su_flags[w] & NAVHIP_SU_TARGET_DIR) {                  /* arrived at the cell, :2663 */
            out->patch.flags = (enum movestate_flags)(out->patch.flags | UPDATE_SET_TARGET_DIR);
            out->patch.next_target_dir = s_move_work.in[w].fstate.target_orientation;
        }
    }
}
#undef DENSE
