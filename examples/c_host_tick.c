/* examples/c_host_tick.c -- a plain C99 host driving the navigation tick through libnavhip.so.
 *
 * "Host code stays in C": this is what the reference's game loop looks like when its navigation tick
 * (move_do_tick -> navigation_tick_task, movement.c:4312, 4263-4280) runs on the device with a resident world:
 * upload the nav planes once (N_CopyCostBasePacked layouts, nav.c:2408-2490), put the snapshot tables and the planner's
 * chunk-field requests into device memory, and call navhip_tick_run once per tick batch.  No Python, no C++: the header
 * is C99, the only other dependency is the HIP runtime's C API for device memory.
 *
 *     c_host_tick <world.bin> <out.bin> <ticks>
 *
 * world.bin (written by tests/test_c_host_gpu.py from a tick.NavTick world): 8 int32 {chunk_w, chunk_h, n_ents, n_flocks,
 * n_reqs, hz, fields_ahead, reserved}, then the arrays in the order read below.  out.bin: positions and velocities after
 * the last tick, the status bytes, the field pool the last tick sampled. */
#include <navhip.h>
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call) do { int rc_ = (call); if(rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? navhip_last_error(ctx) : ""); return 1; } } while(0)
#define HIPOK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while(0)

static void *read_array(FILE *f, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if(bytes && fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

static void *to_device(const void *host, size_t bytes)
{
    void *d = NULL;
    if(hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) { fprintf(stderr, "hipMalloc\n"); exit(2); }
    if(bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "hipMemcpy\n"); exit(2); }
    return d;
}

int main(int argc, char **argv)
{
    navhip_ctx *ctx = NULL;
    if(argc != 4) { fprintf(stderr, "usage: %s world.bin out.bin ticks\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if(!f) { perror(argv[1]); return 2; }
    int32_t h[8];
    if(fread(h, 4, 8, f) != 8) return 2;
    const int W = h[0], H = h[1], n = h[2], F = h[3], nreq = h[4], hz = h[5], ahead = h[6];
    const size_t chunks = (size_t)W * H, cells = chunks * 4096;
    const int ticks = atoi(argv[3]);

    /* ---- the map: three planes of layer 0 (NAV_LAYER_GROUND_1X1) */
    uint8_t  *cost = read_array(f, cells);
    uint16_t *blockers = read_array(f, cells * 2), *liid = read_array(f, cells * 2);
    CHECK(navhip_ctx_create(&ctx, W, H, 0));
    CHECK(navhip_upload_plane(ctx, 0, NAVHIP_PLANE_COST_BASE, cost, cells));
    CHECK(navhip_upload_plane(ctx, 0, NAVHIP_PLANE_BLOCKERS, blockers, cells * 2));
    CHECK(navhip_upload_plane(ctx, 0, NAVHIP_PLANE_LOCAL_ISLANDS, liid, cells * 2));

    /* ---- the planner's request stream and the (destination, chunk) -> slot table (N_FC_GetDestFFMapping) */
    navhip_field_req *reqs = read_array(f, (size_t)nreq * sizeof(navhip_field_req));
    int32_t *slot_tbl = read_array(f, (size_t)F * chunks * 4);
    /* ---- the snapshot: struct move_gamestate + move_work_in + flock tables as structure-of-arrays */
    float *pos = read_array(f, (size_t)n * 8), *vel = read_array(f, (size_t)n * 8), *radius = read_array(f, (size_t)n * 4);
    float *max_speed = read_array(f, (size_t)n * 4), *speed = read_array(f, (size_t)n * 4);
    uint32_t *flags = read_array(f, (size_t)n * 4);
    uint8_t *state = read_array(f, (size_t)n), *los = read_array(f, (size_t)n);
    int32_t *flock = read_array(f, (size_t)n * 4);
    float *ftarget = read_array(f, (size_t)F * 8);
    int32_t *foff = read_array(f, (size_t)(F + 1) * 4), *fmem = read_array(f, (size_t)n * 4);
    fclose(f);

    navhip_tick_desc d;
    memset(&d, 0, sizeof(d));
    navhip_world *w = &d.world;
    w->n_ents = n; w->n_flocks = F; w->hz = hz; w->n_field_slots = nreq;
    w->pos_xz = to_device(pos, (size_t)n * 8); w->vel_xz = to_device(vel, (size_t)n * 8);
    w->radius = to_device(radius, (size_t)n * 4); w->max_speed = to_device(max_speed, (size_t)n * 4);
    w->speed = to_device(speed, (size_t)n * 4); w->flags = to_device(flags, (size_t)n * 4);
    w->state = to_device(state, (size_t)n); w->has_dest_los = to_device(los, (size_t)n);
    w->flock = to_device(flock, (size_t)n * 4); w->flock_target_xz = to_device(ftarget, (size_t)F * 8);
    w->flock_offsets = to_device(foff, (size_t)(F + 1) * 4); w->flock_members = to_device(fmem, (size_t)n * 4);
    w->flock_field_slot = to_device(slot_tbl, (size_t)F * chunks * 4);
    uint8_t *pool0 = NULL, *pool1 = NULL, *status = NULL;
    HIPOK(hipMalloc((void**)&pool0, (size_t)nreq * 4096)); HIPOK(hipMemset(pool0, 0, (size_t)nreq * 4096));
    HIPOK(hipMalloc((void**)&status, (size_t)n)); HIPOK(hipMemset(status, 0, (size_t)n));
    w->field_pool = pool0;
    /* vec3 map_pos and the bg_ent_init bounds of the position snapshot (position.c:276-283): the map spans
     * [-W*128, W*128] x [-H*128, H*128] world units */
    w->map_pos_x = W * 128.0f; w->map_pos_z = -H * 128.0f;
    w->grid_xmin = -W * 128.0f; w->grid_xmax = W * 128.0f; w->grid_zmin = -H * 128.0f; w->grid_zmax = H * 128.0f;
    d.pos_xz_1 = to_device(pos, (size_t)n * 8); d.vel_xz_1 = to_device(vel, (size_t)n * 8);
    d.status = status;
    d.dev_reqs = to_device(reqs, (size_t)nreq * sizeof(navhip_field_req)); d.n_reqs = nreq; d.req_slot0 = 0;
    if(ahead) {                       /* the fields of tick t+1 built during tick t, into the other pool */
        HIPOK(hipMalloc((void**)&pool1, (size_t)nreq * 4096)); HIPOK(hipMemset(pool1, 0, (size_t)nreq * 4096));
        d.field_pool_1 = pool1;
        d.fields_stage = NAVHIP_STAGE_NEIGHBOURS;
    }
    d.flags |= NAVHIP_TICK_OWNS_SNAPSHOT;   /* nothing but the tick writes the snapshot buffers or uses its stream between ticks */

    navhip_tick *tick = NULL;
    CHECK(navhip_tick_create(ctx, &d, &tick));
    CHECK(navhip_tick_run(tick, ticks));          /* asynchronous: the game loop would go on with its frame here */
    CHECK(navhip_tick_sync(tick));
    navhip_tick_info info;
    CHECK(navhip_tick_get_info(tick, &info));
    const int cur = (int)(info.ticks & 1);        /* the buffer set that holds the current snapshot */
    fprintf(stderr, "c_host_tick: %lld ticks, %.3f ms of host time per tick to enqueue them\n", (long long)info.ticks,
            info.ticks ? info.host_enqueue_ms / (double)info.ticks : 0.0);

    HIPOK(hipMemcpy(pos, cur ? d.pos_xz_1 : (const float*)w->pos_xz, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(vel, cur ? d.vel_xz_1 : (const float*)w->vel_xz, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(state, status, (size_t)n, hipMemcpyDeviceToHost));
    uint8_t *pool_host = malloc((size_t)nreq * 4096 + 1);
    /* (the pool the NEXT tick samples; the static map makes both pools equal once two ticks have run) */
    HIPOK(hipMemcpy(pool_host, (ahead && cur) ? pool1 : pool0, (size_t)nreq * 4096, hipMemcpyDeviceToHost));
    FILE *o = fopen(argv[2], "wb");
    if(!o) { perror(argv[2]); return 2; }
    fwrite(pos, 8, (size_t)n, o); fwrite(vel, 8, (size_t)n, o); fwrite(state, 1, (size_t)n, o);
    fwrite(pool_host, 4096, (size_t)nreq, o);
    fclose(o);
    navhip_tick_destroy(tick);
    navhip_ctx_destroy(ctx);
    return 0;
}
