// tick_api.hip -- navhip_tick_*: the whole navigation tick of a device-resident world behind ONE call.
//
// The reference drives its tick from one place -- move_do_tick (movement.c:4312) -> navigation_tick_task (:4263-4280):
// field work, the velocity fork-join (:4182), the snapshot for the next tick -- and so does a C host of this library:
// navhip_tick_run enqueues the field builds, the blocker batch, the velocity step, the slab exchange and the ping-pong
// of the snapshot buffers for n ticks and returns.  The schedule is the one the Python driver (tick.py) measured its
// way to in rounds 2-4: the narrow front of the step on a high-priority stream, the cohesion
// term on a side stream, the fields of tick t+1 built during tick t on a CU-masked stream behind the neighbour walk
// (from a jam's worth of workgroup searches on: with the tick), the exchange on a stream only the next tick's
// snapshot consumers wait for.  It calls the SAME entry points tick.py calls, in the same order per stream: results
// are bit-identical by construction (tests/test_tick_gpu.py).
//
// NAVHIP_TICK_GRAPH: the enqueue cost of a tick is ~25 HIP calls (15 kernels, events, one copy): 0.1 ms from C,
// 0.15 ms from Python -- configs[0]'s whole tick and the floor of strong scaling.  A tick's launches only depend on a
// handful of host-side parities (ping-pong buffers, work-list counter set, cohesion permutation buffer, whether the
// lane regrouping is due, where the field builds start): the tick is captured ONCE per combination into a HIP graph
// (all side streams joined back at the end of the tick, the cross-tick event waits dropped: graph launches on one
// stream are ordered) and replayed with one hipGraphLaunch.  Single-process worlds without moving obstacles.
#include "navhip_internal.h"
#include <cstring>
#include <ctime>
#include <map>
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

#ifdef NH_HOSTSIM
#define NH_TICK_GRAPHS 0          /* (the host emulator's stand-in runtime has no graphs: plain launches) */
#else
#define NH_TICK_GRAPHS 1
#endif

struct nh_graph_entry {
#if NH_TICK_GRAPHS
    hipGraphExec_t exec;
#endif
    int wl_parity_after, coh_parity_after;
    unsigned gen_launches_delta;      // (the field kernels' list counters alternate per launch)
    unsigned sp_builds_delta;         // (a slab step's spatial-hash builds alternate the slab box)
};

struct navhip_tick {
    navhip_ctx      *ctx;
    navhip_tick_desc d;
    navhip_world     W[2];            // the snapshot as tick parity p reads it
    navhip_step_out  O[2];            // ... and writes it
    uint8_t         *pool[2];
    std::vector<int32_t> bounds;
    hipStream_t      s, f, comm;      // agent chain | field builds ahead | exchange
    hipEvent_t       ev_fields[2], ev_step, ev_comm, ev_side, ev_tmp;
    bool             ahead, pipelined, comm_pending, computed, serial, split_mid;
    int64_t          ticks;
    int              regroup_age;
    bool             graph;
    int              graphs_captured;
    std::map<uint32_t, nh_graph_entry> execs;
    double           enqueue_ms;
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
        if(!T->pipelined) {
            // (NAVHIP_TICK_SPLIT_MID: this tick's fields were built during the last one -- with them final, the front of
            // the step also runs the sampling half of the per-agent chain, NAVHIP_PREFETCH_FIELDS_READY; otherwise the
            // front does not wait for them, only the step does, below)
            if(T->split_mid) HIPCHK(ctx, hipStreamWaitEvent(T->s, T->ev_fields[p], 0));
            RCCHK(navhip_agent_prefetch_dev_ex(ctx, w, (void*)T->s, NAVHIP_PREFETCH_FRONT_INLINE | NAVHIP_PREFETCH_SNAPSHOT_HELD
                                                                     | (T->split_mid ? NAVHIP_PREFETCH_FIELDS_READY : 0u)));
        }
        // the fields of the NEXT tick
        if(stage == NAVHIP_STAGE_NEIGHBOURS) RCCHK(navhip_stream_wait_stage(ctx, (void*)T->f, NAVHIP_STAGE_NEIGHBOURS));
        else if(!T->pipelined)               RCCHK(navhip_stream_wait_stage(ctx, (void*)T->f, NAVHIP_STAGE_START));
        RCCHK(build_fields(T, T->pool[p ^ 1], T->f));
        HIPCHK(ctx, hipEventRecord(T->ev_fields[p ^ 1], T->f));
        if(T->pipelined || !T->split_mid)
            HIPCHK(ctx, hipStreamWaitEvent(T->s, T->ev_fields[p], 0));   // this tick's fields (built during the last one)
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
    if(T->pipelined) HIPCHK(ctx, hipEventRecord(T->ev_step, T->s));
    return NAVHIP_OK;
}

#if NH_TICK_GRAPHS
// ---- one tick as a graph: captured once per combination of the host-side parities its launches depend on -----------
static int compute_graph(navhip_tick *T)
{
    navhip_ctx *ctx = T->ctx;
    const int p = (int)(T->ticks & 1);
    const navhip_world *w = &T->W[p];
    // the regrouping cadence of navhip_api.hip's coh_regroup_due, decided HERE (the decision is part of the graph's key)
    int32_t lists[6];
    const bool jam = navhip_step_lists_peek(ctx, lists) == NAVHIP_OK && lists[4] >= 8192;
    const int age = T->regroup_age++;
    const bool regroup = jam || age < 2 || age % 8 == 0;
    const int stage = T->ahead ? fields_stage_now(T) : NAVHIP_STAGE_START;
    const uint32_t key = (uint32_t)p | (uint32_t)(ctx->wl_parity & 1) << 1 | (uint32_t)(ctx->coh_parity & 1) << 2
                       | (uint32_t)regroup << 3 | (uint32_t)(stage == NAVHIP_STAGE_START) << 4
                       | (uint32_t)(ctx->gen_launches & 1u) << 5 | (uint32_t)(ctx->sp_builds & 1u) << 6;
    auto it = T->execs.find(key);
    if(it == T->execs.end()) {
        hipGraph_t g = nullptr;
        const unsigned gen_before = ctx->gen_launches, sp_before = ctx->sp_builds;
        HIPCHK(ctx, hipStreamBeginCapture(T->s, hipStreamCaptureModeRelaxed));
        ctx->regroup_override = regroup ? 1 : 2;
        int rc = NAVHIP_OK;
        if(T->serial) {
            rc = build_fields(T, T->pool[0], T->s);
            ctx->serial_step = true;
            if(!rc) rc = navhip_agent_step_dev(ctx, w, &T->O[p], (void*)T->s);
            ctx->serial_step = false;
        }else{
            rc = navhip_agent_prefetch_dev_ex(ctx, w, (void*)T->s, NAVHIP_PREFETCH_FRONT_INLINE
                                              | ((T->ahead && T->split_mid) ? NAVHIP_PREFETCH_FIELDS_READY : 0u));
            if(!rc && T->ahead) {
                rc = navhip_stream_wait_stage(ctx, (void*)T->f, stage);
                if(!rc) rc = build_fields(T, T->pool[p ^ 1], T->f);
                if(!rc && hipEventRecord(T->ev_fields[0], T->f) != hipSuccess) rc = NAVHIP_ERR_DEVICE;
            }else if(!rc) {
                rc = build_fields(T, T->pool[0], T->s);
            }
            if(!rc) rc = navhip_agent_step_dev(ctx, w, &T->O[p], (void*)T->s);
            // every stream the capture forked into comes back to the origin: the copy of the list counters on the
            // library's side stream, the field builds
            if(!rc && ctx->aux[0]
            && (hipEventRecord(T->ev_side, ctx->aux[0]) != hipSuccess || hipStreamWaitEvent(T->s, T->ev_side, 0) != hipSuccess))
                rc = NAVHIP_ERR_DEVICE;
            if(!rc && T->ahead && hipStreamWaitEvent(T->s, T->ev_fields[0], 0) != hipSuccess) rc = NAVHIP_ERR_DEVICE;
        }
        ctx->regroup_override = 0;
        hipError_t e = hipStreamEndCapture(T->s, &g);
        if(rc || e != hipSuccess || !g) {
            if(g) hipGraphDestroy(g);
            if(!rc) ctx->last_error = std::string("hipStreamEndCapture: ") + hipGetErrorString(e);
            (void)hipGetLastError();
            return rc ? rc : NAVHIP_ERR_DEVICE;
        }
        nh_graph_entry ent;
        e = hipGraphInstantiate(&ent.exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if(e != hipSuccess) {
            ctx->last_error = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
            return NAVHIP_ERR_DEVICE;
        }
        ent.wl_parity_after = ctx->wl_parity; ent.coh_parity_after = ctx->coh_parity;
        ent.gen_launches_delta = ctx->gen_launches - gen_before;
        ent.sp_builds_delta = ctx->sp_builds - sp_before;
        it = T->execs.emplace(key, ent).first;
        T->graphs_captured++;
    }else{
        // what the captured calls did to the library's host-side state
        ctx->wl_parity = it->second.wl_parity_after; ctx->coh_parity = it->second.coh_parity_after;
        ctx->gen_launches += it->second.gen_launches_delta;
        ctx->sp_builds += it->second.sp_builds_delta;
        ctx->counters.step_calls++; ctx->counters.agent_steps += (uint64_t)(w->work_end - w->work_begin);
        if(T->d.n_reqs > 0) { ctx->counters.field_calls++; ctx->counters.chunk_fields += (uint64_t)T->d.n_reqs; }
    }
    ctx->pre.valid = false; ctx->regroup_pending = false;
    HIPCHK(ctx, hipGraphLaunch(it->second.exec, T->s));
    return NAVHIP_OK;
}
#endif

static int tick_compute(navhip_tick *T)
{
    navhip_ctx *ctx = T->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if(T->computed) { ctx->last_error = "navhip_tick_compute: the previous tick has not been advanced"; return NAVHIP_ERR_INVALID; }
#if NH_TICK_GRAPHS
    // (the first ticks run plain: allocations, side streams, derived planes -- nothing of that may happen in a capture)
    if(T->graph && T->ticks >= 2) {
        int rc = compute_graph(T);
        if(rc == NAVHIP_OK) { T->computed = true; return rc; }
        // a capture that failed: this runtime cannot replay the tick -- plain launches from here on
        T->graph = false;
        for(auto &kv : T->execs) hipGraphExecDestroy(kv.second.exec);
        T->execs.clear();
        hipStreamCaptureStatus st;
        if(hipStreamIsCapturing(T->s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            hipGraph_t g = nullptr; hipStreamEndCapture(T->s, &g); if(g) hipGraphDestroy(g);
        }
        (void)hipGetLastError();
        ctx->regroup_override = 0;
        HIPCHK(ctx, hipDeviceSynchronize());
        ctx->pre.valid = false; ctx->regroup_pending = false;
    }
#endif
    int rc = compute_plain(T);
    if(rc == NAVHIP_OK) T->computed = true;
    return rc;
}

static int tick_exchange(navhip_tick *T)
{
    navhip_ctx *ctx = T->ctx;
    if(!T->pipelined) return NAVHIP_OK;
    const int p = (int)(T->ticks & 1);
    HIPCHK(ctx, hipStreamWaitEvent(T->comm, T->ev_step, 0));
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
    T->ahead = desc->field_pool_1 != nullptr && !T->serial;
    T->split_mid = (desc->flags & NAVHIP_TICK_SPLIT_MID) != 0;
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
    hipEvent_t *evs[] = {&T->ev_fields[0], &T->ev_fields[1], &T->ev_step, &T->ev_comm, &T->ev_side, &T->ev_tmp};
    for(hipEvent_t *e : evs) if(hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return fail("navhip_tick_create: event");
    // (a slab step without a static_epoch carries a never-repeating membership key: nothing to replay)
    const bool whole = (w.work_begin == 0 && w.work_end == 0) || (w.work_begin == 0 && w.work_end == w.n_ents);
    T->graph = NH_TICK_GRAPHS && (desc->flags & NAVHIP_TICK_GRAPH) && !T->pipelined && !desc->dev_moves
            && (whole || w.static_epoch != 0);
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
    int rc = tick_compute(T);
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
        rc = tick_compute(T);
        if(!rc) rc = tick_exchange(T);
        if(!rc) { T->computed = false; T->ticks++; }
    }
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
    out->ticks = T->ticks; out->graph = T->graph ? 1 : 0; out->graphs_captured = T->graphs_captured;
    out->host_enqueue_ms = T->enqueue_ms;
    out->stream = (void*)T->s; out->field_stream = (void*)T->f; out->comm_stream = (void*)T->comm;
    return NAVHIP_OK;
}

void navhip_tick_destroy(navhip_tick *T)
{
    if(!T) return;
    hipSetDevice(T->ctx->device);
    if(T->comm) hipStreamSynchronize(T->comm);
    if(T->f) hipStreamSynchronize(T->f);
    if(T->s) hipStreamSynchronize(T->s);
    navhip_sync(T->ctx);
#if NH_TICK_GRAPHS
    for(auto &kv : T->execs) hipGraphExecDestroy(kv.second.exec);
#endif
    hipEvent_t evs[] = {T->ev_fields[0], T->ev_fields[1], T->ev_step, T->ev_comm, T->ev_side, T->ev_tmp};
    for(hipEvent_t e : evs) if(e) hipEventDestroy(e);
    delete T;
}

}  // extern "C"
