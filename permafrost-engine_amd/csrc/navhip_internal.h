// navhip_internal.h -- shared by the HIP translation units of libnavhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "navhip.h"
#include "map_view.h"

static_assert(sizeof(navhip_field_req) == 32, "navhip_field_req must stay 32 bytes");
static_assert(sizeof(navhip_los_req) == 16, "navhip_los_req must stay 16 bytes");

#define NH_CELLS   4096
#define NH_RES     64
#define NH_INF_U32 0x3fffffffu      /* "unreached" integration value (float INFINITY on output) */

struct navhip_layer {
    uint8_t  *cost;            // [nchunks][64][64]
    uint16_t *blockers;        // [nchunks][64][64]
    uint16_t *local_islands;   // [nchunks][64][64]
    uint8_t  *factions;        // [nchunks][15][64][64]
    uint16_t *islands;         // [nchunks][64][64]  global island ids (ISLAND_NEAREST repair only)
    // derived (rebuilt lazily for dirty chunks):
    uint64_t *passmask;        // [nchunks][64]  bit c of word r = cell (r,c) passable, faction NONE
    uint64_t *probemask;       // [nchunks][64][2] bit c of word (r, 0) = cost_base != 0xff, of (r, 1) = blockers > 0:
                               // what the agent kernels' tile probes read (16 bytes per tile row instead of
                               // a byte and a 16-bit word per tile in two planes)
                               //                (field_tile_passable, field.c:117)
    uint8_t  *unit_cost;       // [nchunks]      1 when every cost != 0xff cell has cost 1
    uint8_t  *touched;         // [nchunks]      device: blockers modified since the last refresh
    uint8_t  *changed;         // [nchunks]      device: passability changed since navhip_clear_changed
    uint8_t  *dirty;           // host side: [nchunks] derived state stale
    bool      any_dirty;
    bool      nonunit_costs;   // host side: a cost other than 1 / 0xff was uploaded for this layer
};

struct navhip_ctx {
    int          device;
    int          w, h, nchunks;
    hipStream_t  stream;
    navhip_layer layers[NAVHIP_NAV_LAYER_MAX];
    int          field_kernel_mode;
    // scratch for the host-buffer entry points
    void        *d_reqs;      size_t d_reqs_cap;
    uint8_t     *d_dirs;      size_t d_dirs_cap;
    float       *d_integ;     size_t d_integ_cap;
    uint64_t    *d_reqmask;   size_t d_reqmask_cap;   // per-request passability rows
    uint32_t    *d_dirty_list; size_t d_dirty_cap;
    // agent-step scratch (grown on demand, reused every tick)
    struct buf { void *p; size_t cap; };
    buf          sp[10];       // spatial hash: ent_cell, ent_rank, cell_count, cell_start, tmp_id,
                               //               block_sum, slab box, recA, recV, pool_of
    unsigned     sp_builds;    // spatial-hash builds so far: parity selects the slab box of a build
    buf          nbr[3];       // neighbour walk: separation force, counts, neighbour lists
    buf          arrived[2];   // state update: {x, z, radius, uid} of every flock's ARRIVED members, compacted; counts
    buf          midrec;       // per-entity record k_agent_mid leaves for the work-list consumers
    buf          wl[2];        // work lists: 2 x NH_WL_COUNT counters (alternating), ids
    int          wl_parity;
    int32_t     *lists_pinned;       // pinned host copy of a step's list counters (navhip_step_lists_peek)
    buf          coh;          // cohesion force per entity
    buf          coh_plan;     // [n_flocks + 1] wave prefix of the cohesion launch
    buf          gen_list;     // [2 + n] requests the BFS kernel left to k_field_generic: 2 counters, ids
    unsigned     gen_launches; // parity selects the counter of a launch
    int          coh_flocks, coh_members, coh_parity;   // layout of coh_plan + which perm buffer is next
    unsigned     coh_unique;   // membership keys of slab steps whose caller gave no static_epoch: never equal
    buf          stage[48];    // device copies of host buffers for the host-pointer entry points
    // side streams for navhip_agent_prefetch_dev (spatial hash | cohesion) + fork/join events; BORROWED from the process's
    // set (nh_device_stream)
    hipStream_t  aux[2];
    hipStream_t  aux_main;          // the main stream the side streams were chosen for
    hipEvent_t   ev_regroup;        // side stream 1 has finished the lane regrouping (the only event of the step: rare, off the critical path)
    navhip_counters counters;       // navhip_get_counters
    bool         snapshot_held;     // NAVHIP_PREFETCH_SNAPSHOT_HELD of the last prefetch
    bool         join0_signalled;   // NH_HO_NBR has been (or: is going to be, by a launch already enqueued) stored behind the front of the last prefetch
    bool         lists_signalled;   // the last step forked: NH_HO_MID says when its work lists were complete
    int          start_flag;        // what the side streams of the last prefetch wait for: NH_HO_START, or NH_HO_END of the step it follows,
    int32_t      start_seq;         // ... and the word's number then (later steps advance it)
    bool         step_end_signalled; // the last step forked: NH_HO_END (word or event) says when it had ended
    hipStream_t  step_end_on;       // the stream on which the last step stored NH_HO_START behind its last kernel, or NULL
    hipStream_t  front_stream;      // the stream the last prefetch ran the front of the step on
    bool         regroup_pending;   // a lane regrouping launched by the prefetch has not been joined yet
    bool         serial_step;       // navhip_agent_step_dev runs EVERYTHING on the caller's stream (no side streams, no events):
                                    // the tick of a small world is a chain of dependent launches, and every cross-stream
                                    // edge costs more than the overlap it buys (tick_api.hip, NAVHIP_TICK_SERIAL)
    int64_t      coh_regroup_key[4]; int coh_regroup_age;   // what the last regrouping was built for, ticks since
    // the snapshot a prefetch was started for: everything the side streams baked into their results
    struct { bool valid; const float *pos_xz, *vel_xz, *radius, *arrival_sink_xz; const uint32_t *flags;
             const uint8_t *state, *arrival_flags; const int32_t *flock_members, *flock_offsets;
             int n_ents, n_flocks, hz, work_begin, work_end;
             struct nh_grid_store { int32_t origin_x, origin_y; int grid_w, grid_h; } g;
             } pre;
    // optional per-kernel-group timing of the agent step (navhip_set_profiling)
    bool         profiling;
    hipEvent_t   ev[6];        // start | hash built | neighbour walk | cohesion | regroup | finish
    bool         ev_valid;
    std::string  last_error;
    struct nh_handover *ho;    // hand-overs between the step's streams through device memory (stream_set.hip) or NULL
    struct nh_pool  *pool;     // resident flow-field pool (navhip_pool_*, pool_api.hip) or NULL
    struct nh_async *async;    // state of navhip_agent_step_submit / _poll
    struct nh_comm  *comm;     // RCCL communicator of navhip_comm_* (comm_api.hip) or NULL
};

// The library's streams: one set per process and device, each with a hardware queue of its own, chosen per caller stream
// so that streams which hand over to each other sit on different pipes of the command processor (navhip_api.hip has the
// measurements behind this).  Borrowed by contexts and ticks; never destroyed.
enum { NH_STREAM_SIDE0 = 0,     // the ClearPath side chain of the agent step
       NH_STREAM_SIDE1,         // the cohesion term
       NH_STREAM_MAIN,          // the agent chain (the caller's stream, or the set's own when the caller gave none)
       NH_STREAM_FIELDS,        // field builds beside the step, when they may use every compute unit
       NH_STREAM_COMM,          // the slab exchange (shares the field builds' pipe: one hand-over per tick each)
       NH_STREAM_FIXED };
int         nh_streams_for(navhip_ctx *ctx, hipStream_t main, hipStream_t out[NH_STREAM_FIXED]);
hipStream_t nh_stream_partial_for(navhip_ctx *ctx, hipStream_t main, int cu_begin, int cu_count);   // nullptr: ctx->last_error says why
bool        nh_streams_alive(int device);
void        nh_streams_forget(int device, hipStream_t s);
int         nh_prepare_step_streams(navhip_ctx *ctx, hipStream_t main);      // the side streams of steps whose main chain runs on `main`

// Hand-overs between the library's streams through a word in device memory instead of a queue barrier: the producer's
// stream stores a sequence number behind its last kernel (a one-lane launch, or the last workgroup of the kernel
// itself: nh_signal), the consumer's stream holds a one-lane kernel that ends when the number has arrived.  Measured
// (scripts/stream_flag_probe.hip, profiles/r06_stream_flag_probe.txt): 2-3 us per hand-over against 12 us for an event
// record + event wait between two hardware queues.  The host enqueues the producer first, always: any order of
// execution that respects the order of submission -- the emulator's, a profiler that serialises kernels -- terminates.
enum { NH_HO_COH = 0,           // the cohesion term            (side stream 1 -> the agent chain)
       NH_HO_MID,               // k_agent_mid's work lists      (the agent chain -> side stream 0), stored by the kernel itself
       NH_HO_CP,                // the ClearPath side chain      (side stream 0 -> the agent chain)
       NH_HO_NBR,               // spatial hash + neighbour walk (the agent chain -> whoever waits for NAVHIP_STAGE_NEIGHBOURS)
       NH_HO_START,             // what the caller's stream had reached at a prefetch (-> its side streams)
       NH_HO_END,               // the end of a step on the agent chain (-> the side streams of a prefetch that follows it directly, the exchange)
       NH_HO_FLAGS };
#define NH_HO_STRIDE 32         /* one 128-byte line per word; the ticket of a kernel that signals itself is the word + 1 */
// a number for a word, as a kernel argument: stored by a kernel that FOLLOWS the producer on its stream (its first
// workgroup, when it starts).  flag == nullptr: nobody waits.
struct nh_signal { int32_t *flag; int32_t seq; };
struct nh_handover {
    int32_t *flags;             // device
    int32_t *status;            // pinned host word (device pointer: status_dev): a wait that gave up after two seconds stores 1
    int32_t *status_dev;
    int32_t  seq[NH_HO_FLAGS];  // the number the last producer of a flag stores
    // NAVHIP_HANDOVER=events (read when the context gets its side streams; the default under rocprofv3 --pmc, which says so
    // in ROCPROF_COUNTER_COLLECTION; NAVHIP_HANDOVER=words overrides): every word is an event instead -- record on the
    // producer's stream, event wait on the consumer's; 12 us per hand-over, as before round 6.  For a profiler that
    // SERIALISES kernels (counter collection: one kernel on the device at a time, and not in the order of submission): a
    // kernel that waits for a kernel of another queue never ends there.
    // ... and in a JAM (8 192 workgroup searches and more in the last step: navhip_step_lists_peek, the predicate the field
    // builds and the regrouping cadence already follow): the crowded world runs 5.0-5.1 ms per tick with words and 4.7-4.8
    // with events on the same box, every kernel taking the same time under the tracer (profiles/r06_ab_handover_vs_events.txt)
    // -- three one-lane kernels resident beside milliseconds of persistent searches lose more than five hand-overs of 12 us
    // cost a 5-ms tick.  The mode is chosen at every prefetch (nh_handover_mode) and holds until the next.
    bool       forced_events, never_events, by_events;     // NAVHIP_HANDOVER=events / =words / the mode now
    hipEvent_t ev[NH_HO_FLAGS];
};
void     nh_handover_mode(navhip_ctx *ctx, bool jam);                            // the mode of the hand-overs from here to the next call
int      nh_handover_ensure(navhip_ctx *ctx);                                   // NAVHIP_OK, or an error with ctx->last_error
void     nh_handover_destroy(navhip_ctx *ctx);
// for a kernel that FOLLOWS the producer on `producer` and stores the number itself when it starts: its argument
// ({nullptr, 0} with events: the event is recorded here, in front of that kernel)
nh_signal nh_handover_by_kernel(navhip_ctx *ctx, int flag, hipStream_t producer);
void     nh_handover_signal(navhip_ctx *ctx, int flag, hipStream_t producer);   // behind everything enqueued on `producer` so far
// `consumer` continues when the flag's last producer has stored.  before / after (-1: none): words the waiting kernel
// itself stores when it starts -- it follows their producer on `consumer` -- and when its wait is over
void     nh_handover_wait(navhip_ctx *ctx, int flag, hipStream_t consumer, int before = -1, int after = -1);
void     nh_handover_wait2(navhip_ctx *ctx, int flag_a, int flag_b, hipStream_t consumer);   // both, in one launch
int32_t  nh_handover_seq(navhip_ctx *ctx, int flag);                                         // the number of the flag's last producer
void     nh_handover_wait_for(navhip_ctx *ctx, int flag, int32_t want, hipStream_t consumer); // ... of an earlier one (nh_handover_seq then)
bool     nh_handover_failed(navhip_ctx *ctx);                                   // a wait gave up: ctx->last_error says so

// pool_api.hip <-> navhip_api.hip
extern "C" int nh_validate_field_reqs(navhip_ctx *ctx, const navhip_field_req *reqs, int n);   /* (library internal) */
int  navhip_build_fields_slots_dev(navhip_ctx *ctx, const navhip_field_req *dev_reqs, int n, uint8_t *dev_fields,
                                   const int32_t *dev_slots, hipStream_t s);
int  navhip_stage_reserve(navhip_ctx *ctx, int slot, size_t bytes, void **dev);
void nh_async_destroy(navhip_ctx *ctx);
void nh_async_invalidate_static(navhip_ctx *ctx);   // the staging buffers were used by someone else
bool nh_async_resident(navhip_ctx *ctx, navhip_world *w, navhip_step_out *o);   // snapshot + outputs the last completed submit left on the device
int  nh_async_slabs(navhip_ctx *ctx, size_t in_bytes, size_t out_bytes, char **h_in, char **h_out);
bool nh_is_pinned(const void *p);
int  nh_refresh_derived(navhip_ctx *ctx, hipStream_t s);    // the derived row masks (passmask / probemask) of dirty chunks, rebuilt
int  nh_spatial_query_dev(navhip_ctx *ctx, const navhip_world *dev_w, const float *d_query, int nq, float range, int maxout,
                          int32_t *d_counts, uint32_t *d_ids, hipStream_t s);
const uint8_t *nh_pool_fields(const navhip_ctx *ctx);
const int32_t *nh_pool_map(const navhip_ctx *ctx);
int nh_pool_dests(const navhip_ctx *ctx);
int nh_pool_slots(const navhip_ctx *ctx);

// launched by navhip_api.hip
void nh_launch_derive(navhip_ctx *ctx, int layer, const uint32_t *d_chunk_list, int n,
                      hipStream_t s);
void nh_launch_fields(navhip_ctx *ctx, const navhip_field_req *d_reqs, int n, uint8_t *d_dirs,
                      float *d_integ, int32_t *d_gen_list, hipStream_t s,
                      const int32_t *d_out_slot = nullptr);

void nh_launch_blockers_circles(navhip_ctx *ctx, const navhip_circle *d_circles, int n, float map_x,
                                float map_z, hipStream_t s);
void nh_launch_local_islands(navhip_ctx *ctx, int layer, hipStream_t s);
void nh_launch_region_fields(navhip_ctx *ctx, const navhip_region_req *d_reqs, int n, int max_dim,
                             const int16_t *d_seeds, const int16_t *d_overlay, uint8_t *d_out,
                             size_t out_stride, hipStream_t s);
void nh_launch_los(navhip_ctx *ctx, const navhip_los_req *d_reqs, int n, const uint8_t *d_prev,
                   uint8_t *d_out, float map_x, float map_z, hipStream_t s);

