// stream_set.hip -- the streams the library runs its side chains on (nh_streams_for, nh_stream_partial_for).
#include "navhip_internal.h"
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

// ---------------------------------------------------------------------------------------------------------------
// The library's streams: ONE set per process and device, each with a hardware queue of its own, on a pipe of the
// command processor that the streams it exchanges events with do not use.
//
// Measured on the MI355X (scripts/stream_queue_probe.hip, scripts/stream_pingpong_probe.hip; profiles/r06_stream_*.txt):
//  * HIP maps the streams of one priority onto at most four hardware queues (GPU_MAX_HW_QUEUES); the fifth stream of
//    a priority shares the queue of the fourth, and kernels on streams that share a queue run one after the other.  A
//    host that holds a pool of streams (torch: 32 per priority) has every pooled queue in use: a pooled side stream of
//    this library landed on the queue of the caller's own stream once in four contexts.
//  * A stream created with a CU mask never shares its queue.  But hardware queues sit on the FOUR PIPES of the command
//    processor's compute engine, queue k on pipe k mod 4 in creation order, and a pipe serves one of its queues at a
//    time: a queue that waits for an event (a barrier packet polling a signal) keeps its pipe until the arbiter's
//    time slice ends, and when the queue that is to RECORD that event sits on the same pipe, the hand-over takes
//    100-200 us instead of 12.  The agent step is four streams that hand over to each other four to six times per
//    tick: the tick of one and the same world cost 0.18 or 0.37-0.66 ms depending on how many streams the process had
//    created before (profiles/r06_queue_probe_*.txt).
//  * From about twenty live masked streams on (beside a host's pooled ones) queues are time-sliced, milliseconds at a time.
// So: four masked streams per device, created ONCE, never destroyed, borrowed by every context and tick; which of them
// share a pipe -- with each other and with the caller's stream -- is MEASURED (a ping-pong of two 10-us kernels, 2-5 ms
// per pair, once per process and once per caller stream), and a step's side streams are the ones on the pipes its
// caller's stream does not use.  Contexts on one device that are driven from several threads at once share these
// queues: their work interleaves in stream order, which adds ordering, never removes any.
// ---------------------------------------------------------------------------------------------------------------
#define NH_PIPES 4
__global__ void k_nh_spin(long long ticks, int32_t *word, int32_t seq)
{
#ifndef NH_HOSTSIM
    const long long t0 = wall_clock64();
    while(wall_clock64() - t0 < ticks) { }
    if(word) __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    if(word) *word = seq;
#endif
}

// ---- hand-overs through device memory (navhip_internal.h: nh_handover) ------------------------------------------------
__global__ void k_ho_signal(int32_t *flag, int32_t seq)
{
#ifdef NH_HOSTSIM
    *flag = seq;
#else
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// stores `before` (a word nobody has to wait for any longer once this kernel runs: it follows their producer on its
// stream), waits for `flag`, stores `after` (what the stream has reached once the wait is over)
__global__ void k_ho_wait(const int32_t *flag, int32_t want, int32_t *status, nh_signal before, nh_signal after)
{
#ifdef NH_HOSTSIM
    if(before.flag) *before.flag = before.seq;
    // (launches run to completion in the order of submission here: the producer has run)
    if(*flag - want < 0) { fprintf(stderr, "k_ho_wait: the producer of a hand-over was enqueued behind its consumer\n"); abort(); }
    if(after.flag) *after.flag = after.seq;
#else
    if(before.flag) __hip_atomic_store(before.flag, before.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // (relaxed polls -- a load that goes to the L2, no cache invalidate on this compute unit while others work on it --
    // and ONE acquire when the number has arrived)
    const long long t0 = wall_clock64();
    while(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
        __builtin_amdgcn_s_sleep(8);
        if(wall_clock64() - t0 > 200000000LL) {                   // two seconds at 100 MHz: never hang the device
            __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if(after.flag) __hip_atomic_store(after.flag, after.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

__global__ void k_ho_wait2(const int32_t *flag_a, int32_t want_a, const int32_t *flag_b, int32_t want_b, int32_t *status)
{
#ifdef NH_HOSTSIM
    if(*flag_a - want_a < 0 || *flag_b - want_b < 0) { fprintf(stderr, "k_ho_wait2: the producer of a hand-over was enqueued behind its consumer\n"); abort(); }
#else
    const long long t0 = wall_clock64();
    while(__hip_atomic_load(flag_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want_a < 0 ||
          __hip_atomic_load(flag_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want_b < 0) {
        __builtin_amdgcn_s_sleep(8);
        if(wall_clock64() - t0 > 200000000LL) {
            __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
#endif
}

int nh_handover_ensure(navhip_ctx *ctx)
{
    if(ctx->ho) return NAVHIP_OK;
    nh_handover *H = new nh_handover();
    memset(H, 0, sizeof(*H));
    if(hipMalloc((void**)&H->flags, sizeof(int32_t) * NH_HO_FLAGS * NH_HO_STRIDE) != hipSuccess ||
       hipMemset(H->flags, 0, sizeof(int32_t) * NH_HO_FLAGS * NH_HO_STRIDE) != hipSuccess ||
       hipHostMalloc((void**)&H->status, sizeof(int32_t), hipHostMallocMapped) != hipSuccess ||
       hipHostGetDevicePointer((void**)&H->status_dev, H->status, 0) != hipSuccess) {
        (void)hipGetLastError();
        if(H->flags) hipFree(H->flags);
        if(H->status) hipHostFree(H->status);
        delete H;
        ctx->last_error = "hand-over words: out of memory";
        return NAVHIP_ERR_NOMEM;
    }
    *H->status = 0;
    // (rocprofv3 --pmc exports ROCPROF_COUNTER_COLLECTION to the process it profiles: counter collection serialises kernels)
    const char *mode = getenv("NAVHIP_HANDOVER"), *pmc = getenv("ROCPROF_COUNTER_COLLECTION");
    const bool serialised = pmc && *pmc && strcmp(pmc, "0") && strcmp(pmc, "False") && strcmp(pmc, "false");
    H->forced_events = H->by_events = mode ? !strcmp(mode, "events") : serialised;
    H->never_events = mode && !strcmp(mode, "words");
    if(!H->never_events)
        for(auto &e : H->ev)
            if(hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ctx->last_error = "hand-over events"; ctx->ho = H; nh_handover_destroy(ctx); return NAVHIP_ERR_DEVICE; }
    ctx->ho = H;
    return NAVHIP_OK;
}

void nh_handover_destroy(navhip_ctx *ctx)
{
    if(!ctx->ho) return;
    for(auto &e : ctx->ho->ev) if(e) hipEventDestroy(e);
    hipFree(ctx->ho->flags);
    hipHostFree(ctx->ho->status);
    delete ctx->ho;
    ctx->ho = nullptr;
}

void nh_handover_mode(navhip_ctx *ctx, bool jam)
{
    nh_handover *H = ctx->ho;
    H->by_events = H->forced_events || (jam && !H->never_events);
}

nh_signal nh_handover_by_kernel(navhip_ctx *ctx, int flag, hipStream_t producer)
{
    nh_handover *H = ctx->ho;
    ++H->seq[flag];
    if(H->by_events) { hipEventRecord(H->ev[flag], producer); return nh_signal{nullptr, 0}; }
    return nh_signal{H->flags + flag * NH_HO_STRIDE, H->seq[flag]};
}

void nh_handover_signal(navhip_ctx *ctx, int flag, hipStream_t producer)
{
    nh_handover *H = ctx->ho;
    if(H->by_events) { ++H->seq[flag]; hipEventRecord(H->ev[flag], producer); return; }
    hipLaunchKernelGGL(k_ho_signal, dim3(1), dim3(1), 0, producer, H->flags + flag * NH_HO_STRIDE, ++H->seq[flag]);
}

void nh_handover_wait(navhip_ctx *ctx, int flag, hipStream_t consumer, int before, int after)
{
    nh_handover *H = ctx->ho;
    if(H->by_events) {
        if(before >= 0) { ++H->seq[before]; hipEventRecord(H->ev[before], consumer); }
        hipStreamWaitEvent(consumer, H->ev[flag], 0);
        if(after >= 0) { ++H->seq[after]; hipEventRecord(H->ev[after], consumer); }
        return;
    }
    const int32_t want = H->seq[flag];
    nh_signal b = {nullptr, 0}, a = {nullptr, 0};
    if(before >= 0) { b.flag = H->flags + before * NH_HO_STRIDE; b.seq = ++H->seq[before]; }
    if(after >= 0)  { a.flag = H->flags + after * NH_HO_STRIDE;  a.seq = ++H->seq[after]; }
    hipLaunchKernelGGL(k_ho_wait, dim3(1), dim3(1), 0, consumer, (const int32_t*)(H->flags + flag * NH_HO_STRIDE), want,
                       H->status_dev, b, a);
}

// ... for an earlier number of the word than its last producer's (what nh_handover_seq said then)
void nh_handover_wait_for(navhip_ctx *ctx, int flag, int32_t want, hipStream_t consumer)
{
    nh_handover *H = ctx->ho;
    // (an event has no history: the callers keep to words whose event is not recorded again in between -- NH_HO_START)
    if(H->by_events) { hipStreamWaitEvent(consumer, H->ev[flag], 0); return; }
    hipLaunchKernelGGL(k_ho_wait, dim3(1), dim3(1), 0, consumer, (const int32_t*)(H->flags + flag * NH_HO_STRIDE), want,
                       H->status_dev, nh_signal{nullptr, 0}, nh_signal{nullptr, 0});
}

int32_t nh_handover_seq(navhip_ctx *ctx, int flag) { return ctx->ho->seq[flag]; }

void nh_handover_wait2(navhip_ctx *ctx, int flag_a, int flag_b, hipStream_t consumer)
{
    nh_handover *H = ctx->ho;
    if(H->by_events) { hipStreamWaitEvent(consumer, H->ev[flag_a], 0); hipStreamWaitEvent(consumer, H->ev[flag_b], 0); return; }
    hipLaunchKernelGGL(k_ho_wait2, dim3(1), dim3(1), 0, consumer, (const int32_t*)(H->flags + flag_a * NH_HO_STRIDE), H->seq[flag_a],
                       (const int32_t*)(H->flags + flag_b * NH_HO_STRIDE), H->seq[flag_b], H->status_dev);
}

bool nh_handover_failed(navhip_ctx *ctx)
{
    if(!ctx->ho || !*(volatile int32_t*)ctx->ho->status) return false;
    ctx->last_error = "a hand-over between two streams of the agent step was not signalled within two seconds (under a profiler "
                      "that serialises kernels -- rocprofv3 --pmc -- run with NAVHIP_HANDOVER=events)";
    return true;
}

namespace {
struct nh_dev_streams {
    bool        ready = false;
    hipStream_t full[NH_PIPES] = {};                         // pairwise on different pipes (as measured)
    std::map<std::pair<int, int>, std::array<hipStream_t, NH_PIPES>> partial;   // [k]: on the pipe of full[k]
    std::map<hipStream_t, int> caller_pipe;                  // caller's stream -> k of the full stream whose pipe it shares, -1: none
    int32_t    *words = nullptr, *status = nullptr, *status_dev = nullptr;   // two words in device memory + a pinned status word
    int32_t     seq = 0;
    bool        unmeasured = false;                          // under a profiler that serialises kernels nothing is timed
    double      base_us = 0.0;                               // round trip between two streams on different pipes
    int         measured_pairs = 0, collisions_seen = 0;
};
// same pipe: the round trip of the probe takes 58-69 us instead of 23-24 (profiles/r06_stream_flag_probe.txt)
#define NH_SAME_PIPE 1.6
std::mutex g_streams_mu;
std::map<int, nh_dev_streams> g_streams;

hipStream_t masked_stream(navhip_ctx *ctx, int cu_begin, int cu_count)
{
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) { ctx->last_error = "hipGetDeviceProperties failed"; return nullptr; }
    const int ncu = prop.multiProcessorCount;
    if(cu_count <= 0 || cu_begin + cu_count > ncu) cu_count = ncu - cu_begin;
    uint32_t mask[32] = {0};
    for(int c = cu_begin; c < cu_begin + cu_count && c < 1024; c++) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((ncu + 31) / 32), mask);
    if(e != hipSuccess) { ctx->last_error = std::string("hipExtStreamCreateWithCUMask: ") + hipGetErrorString(e); return nullptr; }
    return st;
}

// microseconds per round trip a -> b -> a of two 10-us kernels that hand over the way the step's streams do -- the first
// stores a word when it ends, a one-lane kernel on the other stream waits for it --, everything enqueued up front
double pingpong_us(nh_dev_streams &D, hipStream_t a, hipStream_t b, int rounds)
{
#ifdef NH_HOSTSIM
    return 1.0;
#else
    if(hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1.0;
    const auto t0 = std::chrono::steady_clock::now();
    for(int r = 0; r < rounds; r++) {
        const int32_t n = ++D.seq;
        hipLaunchKernelGGL(k_nh_spin, dim3(1), dim3(64), 0, a, 1000LL, D.words, n);        // (wall_clock64: 100 MHz)
        hipLaunchKernelGGL(k_ho_wait, dim3(1), dim3(1), 0, b, (const int32_t*)D.words, n, D.status_dev, nh_signal{nullptr, 0}, nh_signal{nullptr, 0});
        hipLaunchKernelGGL(k_nh_spin, dim3(1), dim3(64), 0, b, 1000LL, D.words + NH_HO_STRIDE, n);
        hipLaunchKernelGGL(k_ho_wait, dim3(1), dim3(1), 0, a, (const int32_t*)(D.words + NH_HO_STRIDE), n, D.status_dev, nh_signal{nullptr, 0},
                           nh_signal{nullptr, 0});
    }
    if(hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1.0;
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
#endif
}

// the slower direction of the hand-over between a and b
double handover_us(nh_dev_streams &D, hipStream_t a, hipStream_t b)
{
    if(D.unmeasured) return 0.0;
    pingpong_us(D, a, b, 4);                                           // (first use of a stream creates its queue)
    const double ab = pingpong_us(D, a, b, 24), ba = pingpong_us(D, b, a, 24);
    const double worst = ab > ba ? ab : ba;
    D.measured_pairs++;
    if(D.base_us <= 0.0 || (worst > 0.0 && worst < D.base_us)) D.base_us = worst;      // (the fastest pair seen so far)
    static const bool dbg = getenv("NAVHIP_STREAM_DEBUG") != nullptr;
    if(dbg) fprintf(stderr, "navhip streams: %p <-> %p  %.0f / %.0f us per round trip (fastest pair %.0f)\n", (void*)a, (void*)b, ab, ba, D.base_us);
    return worst;
}

// which of the four full streams shares its pipe with s: all four are measured and the slowest hand-over names it; -1:
// none stands out (s has a pipe to itself, or nothing is being measured)
int pipe_among_full(nh_dev_streams &D, hipStream_t s)
{
    if(D.unmeasured) return -1;
    double us[NH_PIPES];
    int best = 0;
    for(int k = 0; k < NH_PIPES; k++) { us[k] = handover_us(D, s, D.full[k]); if(us[k] > us[best]) best = k; }
    if(!(us[best] > NH_SAME_PIPE * D.base_us)) return -1;
    D.collisions_seen++;
    return best;
}

// At process exit the set is destroyed while the HIP runtime is still up: an atexit handler registered at the set's first
// use runs before the runtime's own (registered earlier, at its initialisation).  Left alive, the streams crashed
// rocprofv3's finalisation (a core dump behind every profiled run: profiles/r06_pytest_gpu_l.log's session).
void streams_atexit()
{
    std::lock_guard<std::mutex> lock(g_streams_mu);
    for(auto &kv : g_streams) {
        nh_dev_streams &D = kv.second;
        if(hipSetDevice(kv.first) != hipSuccess) continue;
        for(auto &st : D.full) if(st) { hipStreamSynchronize(st); hipStreamDestroy(st); st = nullptr; }
        for(auto &pk : D.partial) for(auto &st : pk.second) if(st) { hipStreamSynchronize(st); hipStreamDestroy(st); st = nullptr; }
        if(D.words) hipFree(D.words);
        if(D.status) hipHostFree(D.status);
        D.words = D.status = D.status_dev = nullptr; D.ready = false;
        D.caller_pipe.clear(); D.partial.clear();
    }
}

// Four masked streams on four different pipes.  Consecutive creations land on consecutive pipes; that is verified: every
// pair of the four is timed, and while a pair hands over like two queues on one pipe its later stream is replaced (the
// rejected candidates stay alive until the set is complete -- their queues keep their pipe slots --; at most eight
// replacements, then the set stays as it is: results never depend on it).
bool streams_init(navhip_ctx *ctx, nh_dev_streams &D)
{
    if(D.ready) return true;
    // (rocprofv3 --pmc: kernels are serialised, a kernel that waits for another never ends, and no timing means anything)
    const char *pmc = getenv("ROCPROF_COUNTER_COLLECTION");
    D.unmeasured = pmc && *pmc && strcmp(pmc, "0") && strcmp(pmc, "False") && strcmp(pmc, "false");
    if(!D.unmeasured) {
        if(hipMalloc((void**)&D.words, sizeof(int32_t) * 2 * NH_HO_STRIDE) != hipSuccess ||
           hipMemset(D.words, 0, sizeof(int32_t) * 2 * NH_HO_STRIDE) != hipSuccess ||
           hipHostMalloc((void**)&D.status, sizeof(int32_t), hipHostMallocMapped) != hipSuccess ||
           hipHostGetDevicePointer((void**)&D.status_dev, D.status, 0) != hipSuccess) {
            ctx->last_error = "stream set: out of memory";
            return false;
        }
        *D.status = 0;
    }
    for(int k = 0; k < NH_PIPES; k++) if(!(D.full[k] = masked_stream(ctx, 0, 0))) return false;
    // (registered behind the first allocations of every kind the handler frees: whatever the runtime -- or its stand-in on
    // the host emulator -- set up for them at their first use is torn down after the handler has run, not before)
    static const bool registered = (atexit(streams_atexit), true);
    (void)registered;
    std::vector<hipStream_t> rejected;
    for(int tries = 0; tries < 8 && !D.unmeasured; tries++) {
        double us[NH_PIPES][NH_PIPES];
        for(int i = 0; i < NH_PIPES; i++) for(int j = i + 1; j < NH_PIPES; j++) us[i][j] = handover_us(D, D.full[i], D.full[j]);
        int bad = -1;
        for(int i = 0; i < NH_PIPES && bad < 0; i++) for(int j = i + 1; j < NH_PIPES; j++) if(us[i][j] > NH_SAME_PIPE * D.base_us) { bad = j; break; }
        if(bad < 0) break;
        D.collisions_seen++;
        rejected.push_back(D.full[bad]);
        if(!(D.full[bad] = masked_stream(ctx, 0, 0))) return false;
    }
    for(hipStream_t r : rejected) hipStreamDestroy(r);
    D.ready = true;
    return true;
}

// k of the full stream whose pipe stream s shares (-1: none of them, or s is one of the set)
int pipe_of(navhip_ctx *ctx, nh_dev_streams &D, hipStream_t s)
{
    for(int k = 0; k < NH_PIPES; k++) if(D.full[k] == s) return k;
    for(auto &kv : D.partial) for(int k = 0; k < NH_PIPES; k++) if(kv.second[k] == s) return k;
    auto it = D.caller_pipe.find(s);
    if(it != D.caller_pipe.end()) return it->second;
    int found = -1;
    found = pipe_among_full(D, s);
    D.caller_pipe[s] = found;
    return found;
}
}  // namespace

// a stream the library is about to destroy: what was measured for its handle must not be answered for the next stream
// that gets the same handle
void nh_streams_forget(int device, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(g_streams_mu);
    auto it = g_streams.find(device);
    if(it != g_streams.end()) it->second.caller_pipe.erase(s);
}

// false once the process is exiting and the set is gone (a context destroyed that late must not touch its borrowed streams)
bool nh_streams_alive(int device)
{
    std::lock_guard<std::mutex> lock(g_streams_mu);
    auto it = g_streams.find(device);
    return it != g_streams.end() && it->second.ready;
}

// The streams of a step whose main chain runs on `main` (the caller's stream, or nullptr: the set's own main stream is
// returned in out[NH_STREAM_MAIN]): side0, side1, fields / comm on the other three pipes.
int nh_streams_for(navhip_ctx *ctx, hipStream_t main, hipStream_t out[NH_STREAM_FIXED])
{
    if(hipSetDevice(ctx->device) != hipSuccess) { ctx->last_error = "hipSetDevice failed"; return NAVHIP_ERR_DEVICE; }
    std::lock_guard<std::mutex> lock(g_streams_mu);
    nh_dev_streams &D = g_streams[ctx->device];
    if(!streams_init(ctx, D)) return NAVHIP_ERR_DEVICE;
    int p = main ? pipe_of(ctx, D, main) : 0;
    out[NH_STREAM_MAIN] = main ? main : D.full[0];
    if(p < 0) p = 0;                        // (a caller's stream on a pipe of its own: any three)
    out[NH_STREAM_SIDE0] = D.full[(p + 1) % NH_PIPES];
    out[NH_STREAM_SIDE1] = D.full[(p + 2) % NH_PIPES];
    out[NH_STREAM_FIELDS] = out[NH_STREAM_COMM] = D.full[(p + 3) % NH_PIPES];
    return NAVHIP_OK;
}

// a stream limited to the compute units [cu_begin, cu_begin + cu_count) on the pipe nh_streams_for(main) leaves for the
// field builds (a CU mask is a property of the queue: one queue per pipe and mask, made on first use)
hipStream_t nh_stream_partial_for(navhip_ctx *ctx, hipStream_t main, int cu_begin, int cu_count)
{
    hipDeviceProp_t prop;
    if(hipSetDevice(ctx->device) != hipSuccess || hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) {
        ctx->last_error = "hipGetDeviceProperties failed";
        return nullptr;
    }
    ctx->last_error.clear();
    if(cu_begin < 0 || cu_count <= 0 || cu_begin >= prop.multiProcessorCount) return nullptr;
    if(cu_begin + cu_count > prop.multiProcessorCount) cu_count = prop.multiProcessorCount - cu_begin;
    std::lock_guard<std::mutex> lock(g_streams_mu);
    nh_dev_streams &D = g_streams[ctx->device];
    if(!streams_init(ctx, D)) return nullptr;
    int p = main ? pipe_of(ctx, D, main) : 0;
    if(p < 0) p = 0;
    const int want = (p + 3) % NH_PIPES;
    auto &P = D.partial[std::make_pair(cu_begin, cu_count)];
    if(P[want]) return P[want];
    // candidates until one lands on the wanted pipe; what lands elsewhere is kept for the caller whose stream needs it
    std::vector<hipStream_t> surplus;
    for(int tries = 0; tries < 8 && !P[want]; tries++) {
        hipStream_t c = masked_stream(ctx, cu_begin, cu_count);
        if(!c) break;
        const int k = pipe_among_full(D, c);
        if(k >= 0 && !P[k]) P[k] = c; else surplus.push_back(c);
    }
    if(!P[want] && !surplus.empty()) { P[want] = surplus.back(); surplus.pop_back(); }     // (unplaced: still a queue of its own)
    for(hipStream_t r : surplus) hipStreamDestroy(r);
    return P[want];
}
