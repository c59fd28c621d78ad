// tests/hostsim/wave_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A lockstep emulator for the GROUP code of the device step (permafrost-engine_amd/csrc/agent_group.h): the lanes of
// one group run as fibers (ucontext) on one host thread; every cross-lane operation -- ballot, shuffle, xor shuffle,
// readfirstlane, DPP row shift, wave barrier -- is a rendezvous: a lane posts its operand and yields, and once every
// live lane of the group has arrived at the same operation the results are computed and the lanes resumed.  That is
// the execution model the group code is written for ("everything a group does is group-uniform control flow").  Where
// an operation sits inside a branch that only some lanes take, those lanes are served first and alone (see run()),
// the way EXEC masking runs a branch body before the code behind it.  Lanes that have returned count as inactive.
//
// A wave of G lanes is emulated for a group of G (lane ids 0 .. G-1, so grp<G>::base() is 0 and a ballot carries G
// bits); a workgroup is up to four waves of 64 whose wave-level operations stay inside their wave and whose
// __syncthreads() is a rendezvous of all of them.  Float arithmetic is the host forms of agent_math.h (NH_HOSTSIM): IEEE everywhere, no contraction.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

namespace emu {

enum { K_NONE = 0, K_BALLOT, K_SHFL, K_SHFL_XOR, K_SHFL_UP, K_FIRST, K_SYNC, K_DPP, K_WGSYNC, K_WAVE_ALL };
enum { MAXL = 256 };                  // lanes of a workgroup (up to four waves)

struct Wave {
    int          n;                   // lanes (a group of 16, a wave of 64, or a workgroup of waves)
    ucontext_t   sched;
    ucontext_t   ctx[MAXL];
    char        *stack[MAXL];
    bool         done[MAXL], waiting[MAXL];
    int          kind[MAXL], arg[MAXL];
    uintptr_t    site[MAXL];          // where in the code the lane waits (the call site of the operation)
    uint64_t     val[MAXL], val2[MAXL], res[MAXL];
    int          cur;
    void       (*body)(void *);
    void        *user;
    long         collectives;
    const char  *error;
};

static thread_local Wave *W = nullptr;
struct { unsigned x; } static thread_local threadIdx_emu;

static __attribute__((noinline)) uint64_t collective(int kind, uint64_t v, int arg, uint64_t v2 = 0)
{
    Wave *w = W;
    const int l = w->cur;
    w->kind[l] = kind; w->val[l] = v; w->val2[l] = v2; w->arg[l] = arg; w->waiting[l] = true;
    w->site[l] = (uintptr_t)__builtin_return_address(0);
    swapcontext(&w->ctx[l], &w->sched);
    return w->res[l];
}

static void trampoline()
{
    Wave *w = W;
    const int l = w->cur;
    w->body(w->user);
    w->done[l] = true;
    swapcontext(&w->ctx[l], &w->sched);
}

// run body(user) on n lanes in lockstep; returns nullptr or a description of what went wrong
static const char *run(int n, void (*body)(void *), void *user, long *n_collectives = nullptr)
{
    static thread_local Wave wave;
    Wave *w = &wave;
    W = w;
    w->n = n; w->body = body; w->user = user; w->collectives = 0; w->error = nullptr;
    const size_t STK = 512 * 1024;
    for(int l = 0; l < n; l++) {
        if(!w->stack[l]) w->stack[l] = (char*)malloc(STK);
        w->done[l] = w->waiting[l] = false;
        getcontext(&w->ctx[l]);
        w->ctx[l].uc_stack.ss_sp = w->stack[l];
        w->ctx[l].uc_stack.ss_size = STK;
        w->ctx[l].uc_link = &w->sched;
        makecontext(&w->ctx[l], (void (*)())trampoline, 0);
    }
    const int WS = n < 64 ? n : 64;                 // lanes per wave
    for(;;) {
        int live = 0;
        for(int l = 0; l < n; l++) {
            if(w->done[l] || w->waiting[l]) continue;
            w->cur = l; threadIdx_emu.x = (unsigned)l;
            swapcontext(&w->sched, &w->ctx[l]);
        }
        for(int l = 0; l < n; l++) if(!w->done[l]) live++;
        if(!live) break;
        // Every live lane waits now.  Wave by wave: usually all lanes of a wave at the same operation; lanes can differ
        // where an operation sits inside a branch only some of them took (`gl < n && shfl(...)`): the hardware runs
        // the branch body for those lanes first -- the others are masked off -- and reconverges behind it.  The call
        // site tells who is behind: the lanes at the lowest code address are served, alone; the rest keep waiting.
        // A wave whose lanes all wait at the workgroup barrier is parked until every wave has arrived.
        bool progressed = false;
        int at_barrier = 0;
        for(int w0 = 0; w0 < n; w0 += WS) {
            // readfirstlane marks a value the whole wave agrees on (uni<64>, a ticket drawn by lane 0 at the top of a
            // persistent loop): it is a point where the hardware has the whole wave reconverged.  Groups of 16 lanes that
            // finished their share of a loop iteration early wait there for the others, even though the loop top has
            // the lowest address: such a site is only served once every live lane of the wave stands at it.
            uintptr_t site = ~(uintptr_t)0;
            int first = -1, live_w = 0;
            for(int l = w0; l < w0 + WS; l++) if(!w->done[l]) live_w++;
            for(int pass = 0; pass < 2 && first < 0; pass++)
                for(int l = w0; l < w0 + WS; l++) {
                    if(w->done[l] || w->kind[l] == K_WGSYNC || !(w->site[l] < site)) continue;
                    if(pass == 0 && (w->kind[l] == K_FIRST || w->kind[l] == K_WAVE_ALL)) {
                        int there = 0;
                        for(int k = w0; k < w0 + WS; k++) if(!w->done[k] && w->kind[k] == w->kind[l] && w->site[k] == w->site[l]) there++;
                        if(there != live_w) continue;
                    }
                    site = w->site[l]; first = l;
                }
            if(first < 0) {
                for(int l = w0; l < w0 + WS; l++) if(!w->done[l]) at_barrier++;
                continue;
            }
            const int kind = w->kind[first];
            if((kind == K_FIRST || kind == K_WAVE_ALL) && getenv("WAVE_EMU_DEBUG")) {
                int there = 0;
                for(int k = w0; k < w0 + WS; k++) if(!w->done[k] && w->site[k] == site) there++;
                if(there != live_w) {
                    fprintf(stderr, "wave_emu: wave %d: kind %d served with %d of %d live lanes:", w0 / WS, kind, there, live_w);
                    for(int k = w0; k < w0 + WS; k += 16) fprintf(stderr, " [%d: kind %d site %lx%s]", k, w->kind[k], (unsigned long)(w->site[k] & 0xfffff), w->done[k] ? " done" : "");
                    fprintf(stderr, "\n");
                }
            }
            bool at[MAXL];
            for(int l = w0; l < w0 + WS; l++) at[l] = !w->done[l] && w->kind[l] != K_WGSYNC && w->site[l] == site;
            w->collectives++;
            uint64_t mask = 0;
            switch(kind) {
            case K_BALLOT:
                for(int l = w0; l < w0 + WS; l++) if(at[l] && w->val[l]) mask |= 1ull << (l - w0);
                for(int l = w0; l < w0 + WS; l++) if(at[l]) w->res[l] = mask;
                break;
            case K_SHFL:
                for(int l = w0; l < w0 + WS; l++) if(at[l]) { const int s = w0 + (w->arg[l] & (WS - 1)); w->res[l] = at[s] ? w->val[s] : w->val[l]; }
                break;
            case K_SHFL_XOR:
                for(int l = w0; l < w0 + WS; l++) if(at[l]) { const int s = w0 + (((l - w0) ^ w->arg[l]) & (WS - 1)); w->res[l] = at[s] ? w->val[s] : w->val[l]; }
                break;
            case K_FIRST:
                for(int l = w0; l < w0 + WS; l++) if(at[l]) w->res[l] = w->val[first];
                break;
            case K_SHFL_UP:
                for(int l = w0; l < w0 + WS; l++) if(at[l]) { const int s = l - w->arg[l]; w->res[l] = (s >= w0 && at[s]) ? w->val[s] : w->val[l]; }
                break;
            case K_DPP: {
                // v_mov_b32_dpp: arg = dpp_ctrl | row_mask << 12 | bound_ctrl << 16; val = source, val2 = old.
                // A lane whose row is not in row_mask keeps old; a lane whose source lane does not exist (or is
                // masked off) gets 0 with bound_ctrl, else old.
                for(int l = w0; l < w0 + WS; l++) if(at[l]) {
                    const int ctrl = w->arg[l] & 0xfff, row_mask = (w->arg[l] >> 12) & 0xf, bc = (w->arg[l] >> 16) & 1;
                    const int i = l - w0, row = i >> 4;
                    int src = -1;
                    bool none = false;                               // the control names no source for this lane
                    if(ctrl < 0x100)        src = (i & ~3) + ((ctrl >> (2 * (i & 3))) & 3);           // quad_perm
                    else if(ctrl >= 0x111 && ctrl <= 0x11f) { const int k = ctrl - 0x110; if((i & 15) >= k) src = i - k; else none = true; }   // row_shr
                    else if(ctrl >= 0x101 && ctrl <= 0x10f) { const int k = ctrl - 0x100; if((i & 15) + k <= 15) src = i + k; else none = true; }  // row_shl
                    else if(ctrl == 0x138) { if(i >= 1) src = i - 1; else none = true; }                // wave_shr:1
                    else if(ctrl == 0x130) { if(i + 1 < WS) src = i + 1; else none = true; }            // wave_shl:1
                    else if(ctrl == 0x142) { if(row >= 1) src = (row - 1) * 16 + 15; else none = true; } // row_bcast:15
                    else if(ctrl == 0x143) { if(row >= 2) src = 31; else none = true; }                  // row_bcast:31
                    else { w->error = "DPP control not emulated"; return w->error; }
                    const bool enabled = (row_mask >> row) & 1;
                    uint64_t r = w->val2[l];
                    if(enabled) {
                        if(!none && src >= 0 && src < WS && at[w0 + src]) r = w->val[w0 + src];
                        else if(bc) r = 0;
                    }
                    w->res[l] = r;
                }
                break; }
            case K_SYNC:
                break;
            case K_WAVE_ALL: {
                // one read for the whole wave, at the moment all of it is here (val = address, arg = size)
                uint64_t v = 0;
                memcpy(&v, (const void*)(uintptr_t)w->val[first], (size_t)w->arg[first]);
                for(int l = w0; l < w0 + WS; l++) if(at[l]) w->res[l] = v;
                break; }
            default:
                w->error = "unknown cross-lane operation"; return w->error;
            }
            for(int l = w0; l < w0 + WS; l++) if(at[l]) w->waiting[l] = false;
            progressed = true;
        }
        if(!progressed) {
            if(at_barrier != live) { w->error = "deadlock: nothing to serve and not every lane at the workgroup barrier"; return w->error; }
            w->collectives++;
            uint64_t any = 0, all = 1;
            for(int l = 0; l < n; l++) if(!w->done[l]) { any |= w->val[l] ? 1 : 0; all &= w->val[l] ? 1 : 0; }
            for(int l = 0; l < n; l++) if(!w->done[l]) { w->res[l] = any | (all << 1); w->waiting[l] = false; }   // __syncthreads(_or/_and): everybody is here
        }
    }
    if(n_collectives) *n_collectives = w->collectives;
    return nullptr;
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

}  // namespace emu

// ---- the device vocabulary agent_group.h uses, on top of the rendezvous ---------------------------------------
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define threadIdx emu::threadIdx_emu
static inline unsigned long long __ballot(bool p) { return emu::collective(emu::K_BALLOT, p ? 1 : 0, 0); }
static inline int   __shfl(int v, int src)     { return (int)(uint32_t)emu::collective(emu::K_SHFL, (uint32_t)v, src); }
static inline float __shfl(float v, int src)   { return emu::u2f((uint32_t)emu::collective(emu::K_SHFL, emu::f2u(v), src)); }
static inline uint64_t __shfl(uint64_t v, int src) { return emu::collective(emu::K_SHFL, v, src); }
static inline unsigned long long __shfl(unsigned long long v, int src) { return emu::collective(emu::K_SHFL, v, src); }
static inline unsigned __shfl(unsigned v, int src) { return (unsigned)emu::collective(emu::K_SHFL, v, src); }
static inline int   __shfl_xor(int v, int m)   { return (int)(uint32_t)emu::collective(emu::K_SHFL_XOR, (uint32_t)v, m); }
static inline float __shfl_xor(float v, int m) { return emu::u2f((uint32_t)emu::collective(emu::K_SHFL_XOR, emu::f2u(v), m)); }
static inline int   emu_readfirstlane(int v)   { return (int)(uint32_t)emu::collective(emu::K_FIRST, (uint32_t)v, 0); }
static inline int   emu_update_dpp(int old, int v, int ctrl, int row_mask, int, bool bound_ctrl)
{
    return (int)(uint32_t)emu::collective(emu::K_DPP, (uint32_t)v, ctrl | (row_mask << 12) | ((bound_ctrl ? 1 : 0) << 16), (uint32_t)old);
}
static inline int   __shfl_up(int v, int d)    { return (int)(uint32_t)emu::collective(emu::K_SHFL_UP, (uint32_t)v, d); }
static inline bool  __any(bool p)              { return __ballot(p) != 0ull; }
static inline bool  __all(bool p)              { return __ballot(!p) == 0ull; }
static inline int   __syncthreads_or(int p)    { return (int)(emu::collective(emu::K_WGSYNC, p ? 1 : 0, 0) & 1); }
static inline int   __syncthreads_and(int p)   { return (int)((emu::collective(emu::K_WGSYNC, p ? 1 : 0, 0) >> 1) & 1); }
static inline unsigned emu_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
// A wave-uniform read of a counter that one lane is about to bump (the ticket draw of the persistent kernels: every lane
// checks the counter, lane 0 takes the ticket): in lockstep every lane has read before any lane writes.  Lanes run ahead
// of each other here, so the read is a rendezvous of the whole wave: ONE read when all of them are there, the same
// value for every lane.
template <typename T> static inline T emu_atomic_load(const T *p) { const uint64_t v = emu::collective(emu::K_WAVE_ALL, (uint64_t)(uintptr_t)p, (int)sizeof(T)); T r; memcpy(&r, &v, sizeof(T)); return r; }
#define __atomic_load_n(p, order) emu_atomic_load(p)
#define __builtin_amdgcn_update_dpp(o, v, c, r, b, bc) emu_update_dpp(o, v, c, r, b, bc)
#define __builtin_amdgcn_alignbit(hi, lo, sh) emu_alignbit(hi, lo, sh)
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)emu::collective(emu::K_SYNC, 0, 0))
static inline void __syncthreads() { (void)emu::collective(emu::K_WGSYNC, 0, 0); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int   __float_as_int(float f)   { return (int)emu::f2u(f); }
static inline unsigned __float_as_uint(float f) { return emu::f2u(f); }
static inline float __int_as_float(int i)     { return emu::u2f((uint32_t)i); }
template <typename T, typename U> static inline T atomicAdd(T *p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <typename T, typename U> static inline T atomicMax(T *p, U v) { T o = *p; if((T)v > o) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicMin(T *p, U v) { T o = *p; if((T)v < o) *p = (T)v; return o; }
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }
