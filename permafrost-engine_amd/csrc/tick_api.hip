// tick_api.hip -- navhip_tick_*: the whole navigation tick of a device-resident world behind ONE call.
//
// The reference drives its tick from one place -- move_do_tick (movement.c:4312) -> navigation_tick_task (:4263-4280):
// field work, the velocity fork-join (:4182), the snapshot for the next tick -- and so does a C host of this library:
// navhip_tick_run enqueues the field builds, the blocker batch, the velocity step, the slab exchange and the ping-pong
// of the snapshot buffers for n ticks and returns.  The schedule is the one the Python driver (tick.py) measured its
// way to in rounds 2-6 (profiles/HISTORY.md 3.7): the narrow front of the step on the main stream, the
// cohesion term on a side stream, the fields of tick t+1 built during tick t on a CU-masked stream behind the neighbour
// walk (from a jam's worth of workgroup searches on: with the tick), the exchange on a stream only the next tick's
// snapshot consumers wait for -- every one of them a stream of the process's own set, each on a pipe of the command
// processor to itself, handing over through words in device memory (csrc/stream_set.hip).  It calls the SAME entry
// points tick.py calls, in the same order per stream: results are bit-identical by construction (tests/test_tick_gpu.py).
//
// A captured HIP graph per tick was built and measured in round 5 (0.448 against 0.342 ms per tick at configs[2]: a
// barrier packet per cross-stream edge, and streams inside a graph carry neither CU mask nor queue of their own) and
// removed in round 6: profiles/r05_host_overhead_c.txt.
#include "navhip_internal.h"
#include <cstring>
#include <ctime>
#include <new>

#define HIPCHK(ctx, expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if(_e != hipSuccess) {                                                              \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);          \
            return NAVHIP_ERR_DEVICE;                                                       \
        }                                                                                   \
    } while(0)
#define RCCHK(expr) do { int _rc = (expr); if(_rc) return _rc; } while(0)

struct navhip_tick {
    navhip_ctx      *ctx;
    navhip_tick_desc d;
    navhip_world     W[2];            // the snapshot as tick parity p reads it
    navhip_step_out  O[2];            // ... and writes it
    uint8_t         *pool[2];
    std::vector<int32_t> bounds;
    hipStream_t      s, f, comm;      // agent chain | field builds ahead | exchange
    hipEvent_t       ev_fields[2], ev_step, ev_comm, ev_tmp;
    bool             ahead, pipelined, comm_pending, computed, serial;
    bool             follows;         // nothing came between the last tick's step and this tick on T->s
    bool             owns, stepped;   // NAVHIP_TICK_OWNS_SNAPSHOT; a tick has been computed
    bool             step_flagged;    // the last step stored its end in device memory (no ev_step was recorded)
    int64_t          ticks;
    double           enqueue_ms;
    // NAVHIP_TICK_TIME_FIELDS: event pairs on the field stream around the builds of every fourth tick
    bool             time_fields;
    hipEvent_t       ft[8][2];
    bool             ft_used[8];
    int              ft_next;
    double           fields_ms_sum; int fields_samples;
};

static double now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// where in tick t the builds of tick t+1 start: behind the neighbour walk -- unless the workgroup ClearPath searches of
// a jam would then starve them of registers for milliseconds (tick.py, round 4: 5.3 -> 6.8 ms per tick)
static int fields_stage_now(navhip_tick *T)
{
    int stage = T->d.fields_stage == NAVHIP_STAGE_START ? NAVHIP_STAGE_START : NAVHIP_STAGE_NEIGHBOURS;
    int32_t lists[6];
    if(stage == NAVHIP_STAGE_NEIGHBOURS && navhip_step_lists_peek(T->ctx, lists) == NAVHIP_OK && lists[4] >= 8192)
        stage = NAVHIP_STAGE_START;
    return stage;
}

static int build_fields(navhip_tick *T, uint8_t *pool, hipStream_t st)
{
    if(T->d.n_reqs <= 0) return NAVHIP_OK;
    return navhip_build_fields_dev(T->ctx, T->d.dev_reqs, T->d.n_reqs, pool + (size_t)T->d.req_slot0 * NAVHIP_FIELD_CELLS,
                                   nullptr, (void*)st);
}

// pairs whose second event has completed are read and freed (no waiting)
static void harvest_field_times(navhip_tick *T, bool wait)
{
    for(int k = 0; k < 8; k++) {
        if(!T->ft_used[k]) continue;
        if(wait ? hipEventSynchronize(T->ft[k][1]) != hipSuccess : hipEventQuery(T->ft[k][1]) != hipSuccess) { (void)hipGetLastError(); continue; }
        float ms = 0.0f;
        if(hipEventElapsedTime(&ms, T->ft[k][0], T->ft[k][1]) == hipSuccess) { T->fields_ms_sum += ms; T->fields_samples++; }
        T->ft_used[k] = false;
    }
}

static int build_fields_timed(navhip_tick *T, uint8_t *pool, hipStream_t st)
{
    int slot = -1;
    if(T->time_fields && (T->ticks & 3) == 0 && T->d.n_reqs > 0) {
        harvest_field_times(T, false);
        for(int k = 0; k < 8 && slot < 0; k++) if(!T->ft_used[(T->ft_next + k) & 7]) slot = (T->ft_next + k) & 7;
    }
    if(slot >= 0) HIPCHK(T->ctx, hipEventRecord(T->ft[slot][0], st));
    int rc = build_fields(T, pool, st);
    if(slot >= 0 && !rc) { HIPCHK(T->ctx, hipEventRecord(T->ft[slot][1], st)); T->ft_used[slot] = true; T->ft_next = (slot + 1) & 7; }
    return rc;
}

// ---- one tick, plain launches: tick.py's compute() / _compute_pipelined(), call for call ---------------------------
static int compute_plain(navhip_tick *T)
{
    navhip_ctx *ctx = T->ctx;
    const int p = (int)(T->ticks & 1);
    const navhip_world *w = &T->W[p];
    if(T->serial) {
        // one stream, no events: blockers -> fields -> spatial hash -> neighbour walk -> cohesion -> the rest of the step
        if(T->d.dev_moves) {
            const navhip_circle *mv = T->d.dev_moves + (size_t)((T->d.move_tick0 + T->ticks) % T->d.n_move_ticks) * T->d.n_moves;
            RCCHK(navhip_blockers_circles_dev(ctx, mv, T->d.n_moves, w->map_pos_x, w->map_pos_z, (void*)T->s));
        }
        RCCHK(build_fields(T, T->pool[0], T->s));
        if(T->d.dev_moves) RCCHK(navhip_clear_changed(ctx, (void*)T->s));
        if(T->comm_pending) HIPCHK(ctx, hipStreamWaitEvent(T->s, T->ev_comm, 0));
        ctx->serial_step = true;
        int rc = navhip_agent_step_dev(ctx, w, &T->O[p], (void*)T->s);
        ctx->serial_step = false;
        if(rc) return rc;
        T->step_flagged = false;
        if(T->pipelined) HIPCHK(ctx, hipEventRecord(T->ev_step, T->s));
        return NAVHIP_OK;
    }
    if(T->pipelined)            // behind the previous tick's all-gather, beside the field builds below
        RCCHK(navhip_agent_prefetch_dev(ctx, w, (void*)T->comm));
    if(T->ahead) {
        const int stage = fields_stage_now(T);
        if(stage == NAVHIP_STAGE_START && T->pipelined) {          // (the end of the previous tick)
            HIPCHK(ctx, hipEventRecord(T->ev_tmp, T->s));
            HIPCHK(ctx, hipStreamWaitEvent(T->f, T->ev_tmp, 0));
        }
        // (inside one navhip_tick_run nothing comes between a step and the next tick's prefetch on T->s, and the snapshot
        // of a tick is the output of the last one)
        if(!T->pipelined)
            RCCHK(navhip_agent_prefetch_dev_ex(ctx, w, (void*)T->s, NAVHIP_PREFETCH_FRONT_INLINE | NAVHIP_PREFETCH_SNAPSHOT_HELD |
                                               (T->follows ? NAVHIP_PREFETCH_FOLLOWS_STEP : 0u)));
        HIPCHK(ctx, hipStreamWaitEvent(T->s, T->ev_fields[p], 0));       // this tick's fields (built during the last one)
        if(T->comm_pending) HIPCHK(ctx, hipStreamWaitEvent(T->s, T->ev_comm, 0));   // the other ranks' rows of the snapshot
        RCCHK(navhip_agent_step_dev(ctx, w, &T->O[p], (void*)T->s));
        // (a step that forked has said in device memory that it ended: the exchange waits for that word, not for an event)
        T->step_flagged = ctx->step_end_signalled;
        if(T->pipelined && !T->step_flagged) HIPCHK(ctx, hipEventRecord(T->ev_step, T->s));
        // the fields of the NEXT tick: enqueued behind the step -- the step's own wait for the cohesion term is the
        // launch that says "the neighbour walk is done" -- and started by the device as soon as that is so
        if(stage == NAVHIP_STAGE_NEIGHBOURS) RCCHK(navhip_stream_wait_stage(ctx, (void*)T->f, NAVHIP_STAGE_NEIGHBOURS));
        else if(!T->pipelined)               RCCHK(navhip_stream_wait_stage(ctx, (void*)T->f, NAVHIP_STAGE_START));
        RCCHK(build_fields_timed(T, T->pool[p ^ 1], T->f));
        HIPCHK(ctx, hipEventRecord(T->ev_fields[p ^ 1], T->f));
        return NAVHIP_OK;
    }else{
        if(!T->pipelined) RCCHK(navhip_agent_prefetch_dev(ctx, w, (void*)T->s));
        if(T->d.dev_moves) {
            const navhip_circle *mv = T->d.dev_moves + (size_t)((T->d.move_tick0 + T->ticks) % T->d.n_move_ticks) * T->d.n_moves;
            RCCHK(navhip_blockers_circles_dev(ctx, mv, T->d.n_moves, w->map_pos_x, w->map_pos_z, (void*)T->s));
        }
        RCCHK(build_fields(T, T->pool[0], T->s));
        if(T->d.dev_moves) RCCHK(navhip_clear_changed(ctx, (void*)T->s));
    }
    if(T->comm_pending) HIPCHK(ctx, hipStreamWaitEvent(T->s, T->ev_comm, 0));   // the other ranks' rows of the snapshot
    RCCHK(navhip_agent_step_dev(ctx, w, &T->O[p], (void*)T->s));
    T->step_flagged = false;
    if(T->pipelined) HIPCHK(ctx, hipEventRecord(T->ev_step, T->s));
    return NAVHIP_OK;
}

static int tick_compute(navhip_tick *T)
{
    navhip_ctx *ctx = T->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if(T->computed) { ctx->last_error = "navhip_tick_compute: the previous tick has not been advanced"; return NAVHIP_ERR_INVALID; }
    int rc = compute_plain(T);
    if(rc == NAVHIP_OK) T->computed = T->stepped = true;
    return rc;
}

static int tick_exchange(navhip_tick *T)
{
    navhip_ctx *ctx = T->ctx;
    if(!T->pipelined) return NAVHIP_OK;
    const int p = (int)(T->ticks & 1);
    if(T->step_flagged) RCCHK(navhip_stream_wait_stage(ctx, (void*)T->comm, NAVHIP_STAGE_END));
    else                HIPCHK(ctx, hipStreamWaitEvent(T->comm, T->ev_step, 0));
    RCCHK(navhip_comm_allgather_step_dev(ctx, T->O[p].new_pos_xz, T->O[p].vel_xz, T->bounds.data(), (void*)T->comm));
    HIPCHK(ctx, hipEventRecord(T->ev_comm, T->comm));
    T->comm_pending = true;
    return NAVHIP_OK;
}

extern "C" {

int navhip_tick_create(navhip_ctx *ctx, const navhip_tick_desc *desc, navhip_tick **out)
{
    if(!ctx || !desc || !out) return NAVHIP_ERR_INVALID;
    const navhip_world &w = desc->world;
    if(w.n_ents <= 0 || !desc->pos_xz_1 || !desc->vel_xz_1 || !w.pos_xz || !w.vel_xz || desc->n_reqs < 0
    || (desc->n_reqs > 0 && (!desc->dev_reqs || !w.field_pool)) || desc->req_slot0 < 0
    || (desc->dev_moves && (desc->n_moves <= 0 || desc->n_move_ticks <= 0 || desc->move_tick0 < 0 || desc->field_pool_1))) {
        ctx->last_error = "navhip_tick_create: malformed description";
        return NAVHIP_ERR_INVALID;
    }
    if(desc->bounds && !ctx->comm) {
        ctx->last_error = "navhip_tick_create: slab bounds without a communicator (navhip_comm_init)";
        return NAVHIP_ERR_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    navhip_tick *T = new (std::nothrow) navhip_tick();
    if(!T) return NAVHIP_ERR_NOMEM;
    T->ctx = ctx; T->d = *desc;
    T->serial = (desc->flags & NAVHIP_TICK_SERIAL) != 0;
    T->owns = (desc->flags & NAVHIP_TICK_OWNS_SNAPSHOT) != 0;
    T->time_fields = (desc->flags & NAVHIP_TICK_TIME_FIELDS) != 0;
    T->ahead = desc->field_pool_1 != nullptr && !T->serial;
    T->pipelined = desc->bounds != nullptr;
    if(T->pipelined) {
        const int world = navhip_comm_world(ctx);
        T->bounds.assign(desc->bounds, desc->bounds + world + 1);
        T->d.bounds = T->bounds.data();
    }
    T->pool[0] = const_cast<uint8_t*>(w.field_pool);
    T->pool[1] = T->ahead ? desc->field_pool_1 : T->pool[0];
    for(int p = 0; p < 2; p++) {
        T->W[p] = w;
        T->W[p].pos_xz = p ? desc->pos_xz_1 : w.pos_xz;
        T->W[p].vel_xz = p ? desc->vel_xz_1 : w.vel_xz;
        T->W[p].field_pool = T->pool[p];
        T->O[p] = navhip_step_out{const_cast<float*>(p ? w.vel_xz : desc->vel_xz_1), const_cast<float*>(p ? w.pos_xz : desc->pos_xz_1),
                                  desc->vdes_xz, desc->vpref_xz, desc->status};
    }
    auto fail = [&](const char *what) { ctx->last_error = what; navhip_tick_destroy(T); return NAVHIP_ERR_DEVICE; };
    // (streams the caller did not give: the process's own, each with a hardware queue to itself on a pipe of the command
    // processor the others do not use -- nh_streams_for)
    hipStream_t own[NH_STREAM_FIXED];
    if(nh_streams_for(ctx, (hipStream_t)desc->stream, own) != NAVHIP_OK) { navhip_tick_destroy(T); return NAVHIP_ERR_DEVICE; }
    T->s = own[NH_STREAM_MAIN];
    T->f = (hipStream_t)desc->field_stream;
    if(!T->f && T->ahead) {
        hipDeviceProp_t prop;
        int ncu = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 0;
        T->f = (desc->field_cus > 0 && desc->field_cus < ncu) ? nh_stream_partial_for(ctx, T->s, ncu - desc->field_cus, desc->field_cus)
                                                               : own[NH_STREAM_FIELDS];
        if(!T->f) return fail("navhip_tick_create: field stream");
    }
    T->comm = (hipStream_t)desc->comm_stream;
    if(!T->comm && T->pipelined) T->comm = own[NH_STREAM_COMM];
    // (the step's side streams are chosen for THIS stream, whatever stream a prefetch runs on)
    if(nh_prepare_step_streams(ctx, T->s) != NAVHIP_OK) { navhip_tick_destroy(T); return NAVHIP_ERR_DEVICE; }
    hipEvent_t *evs[] = {&T->ev_fields[0], &T->ev_fields[1], &T->ev_step, &T->ev_comm, &T->ev_tmp};
    for(hipEvent_t *e : evs) if(hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return fail("navhip_tick_create: event");
    if(T->time_fields)
        for(auto &pair : T->ft) for(auto &e : pair) if(hipEventCreate(&e) != hipSuccess) return fail("navhip_tick_create: event");
    if(T->ahead) {
        // the fields tick 0 samples: start-up, on the agent stream
        int rc = build_fields(T, T->pool[0], T->s);
        if(rc) { navhip_tick_destroy(T); return rc; }
        if(hipEventRecord(T->ev_fields[0], T->s) != hipSuccess) return fail("navhip_tick_create: event record");
    }
    *out = T;
    return NAVHIP_OK;
}

int navhip_tick_compute(navhip_tick *T)
{
    if(!T) return NAVHIP_ERR_INVALID;
    const double t0 = now_ms();
    T->follows = T->owns && T->stepped;
    int rc = tick_compute(T);
    T->follows = false;
    T->enqueue_ms += now_ms() - t0;
    return rc;
}

int navhip_tick_advance(navhip_tick *T)
{
    if(!T || !T->computed) return NAVHIP_ERR_INVALID;
    T->computed = false;
    T->ticks++;
    return NAVHIP_OK;
}

int navhip_tick_run(navhip_tick *T, int n)
{
    if(!T || n < 0) return NAVHIP_ERR_INVALID;
    const double t0 = now_ms();
    int rc = NAVHIP_OK;
    for(int i = 0; i < n && !rc; i++) {
        T->follows = i > 0 || (T->owns && T->stepped);
        rc = tick_compute(T);
        if(!rc) rc = tick_exchange(T);
        if(!rc) { T->computed = false; T->ticks++; }
    }
    T->follows = false;
    T->enqueue_ms += now_ms() - t0;
    return rc;
}

int navhip_tick_sync(navhip_tick *T)
{
    if(!T) return NAVHIP_ERR_INVALID;
    navhip_ctx *ctx = T->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if(T->comm) HIPCHK(ctx, hipStreamSynchronize(T->comm));
    if(T->f)    HIPCHK(ctx, hipStreamSynchronize(T->f));
    HIPCHK(ctx, hipStreamSynchronize(T->s));
    return navhip_sync(ctx);
}

int navhip_tick_get_info(const navhip_tick *T, navhip_tick_info *out)
{
    if(!T || !out) return NAVHIP_ERR_INVALID;
    out->ticks = T->ticks;
    harvest_field_times(const_cast<navhip_tick*>(T), false);
    out->fields_ms = T->fields_samples ? T->fields_ms_sum / T->fields_samples : 0.0; out->fields_samples = T->fields_samples; out->_pad = 0;
    out->host_enqueue_ms = T->enqueue_ms;
    out->stream = (void*)T->s; out->field_stream = (void*)T->f; out->comm_stream = (void*)T->comm;
    return NAVHIP_OK;
}

void navhip_tick_destroy(navhip_tick *T)
{
    if(!T) return;
    hipSetDevice(T->ctx->device);
    if(nh_streams_alive(T->ctx->device)) {       // (not while the process exits: the library's streams are gone by then)
        if(T->comm) hipStreamSynchronize(T->comm);
        if(T->f) hipStreamSynchronize(T->f);
        if(T->s) hipStreamSynchronize(T->s);
        navhip_sync(T->ctx);
    }
    hipEvent_t evs[] = {T->ev_fields[0], T->ev_fields[1], T->ev_step, T->ev_comm, T->ev_tmp};
    for(hipEvent_t e : evs) if(e) hipEventDestroy(e);
    for(auto &pair : T->ft) for(auto &e : pair) if(e) hipEventDestroy(e);
    delete T;
}

}  // extern "C"
