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

enum { K_NONE = 0, K_BALLOT, K_SHFL, K_SHFL_XOR, K_FIRST, K_SYNC, K_DPP_SHR, K_WGSYNC };
enum { MAXL = 256 };                  // lanes of a workgroup (up to four waves)

struct Wave {
    int          n;                   // lanes (a group of 16, a wave of 64, or a workgroup of waves)
    ucontext_t   sched;
    ucontext_t   ctx[MAXL];
    char        *stack[MAXL];
    bool         done[MAXL], waiting[MAXL];
    int          kind[MAXL], arg[MAXL];
    uintptr_t    site[MAXL];          // where in the code the lane waits (the call site of the operation)
    uint64_t     val[MAXL], res[MAXL];
    int          cur;
    void       (*body)(void *);
    void        *user;
    long         collectives;
    const char  *error;
};

static thread_local Wave *W = nullptr;
struct { unsigned x; } static thread_local threadIdx_emu;

static __attribute__((noinline)) uint64_t collective(int kind, uint64_t v, int arg)
{
    Wave *w = W;
    const int l = w->cur;
    w->kind[l] = kind; w->val[l] = v; w->arg[l] = arg; w->waiting[l] = true;
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
            uintptr_t site = ~(uintptr_t)0;
            int first = -1;
            for(int l = w0; l < w0 + WS; l++)
                if(!w->done[l] && w->kind[l] != K_WGSYNC && w->site[l] < site) { site = w->site[l]; first = l; }
            if(first < 0) {
                for(int l = w0; l < w0 + WS; l++) if(!w->done[l]) at_barrier++;
                continue;
            }
            const int kind = w->kind[first];
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
            case K_DPP_SHR:       // row_shr:k inside rows of 16 lanes, bound_ctrl: 0 shifted in
                for(int l = w0; l < w0 + WS; l++) if(at[l]) { const int k = w->arg[l]; w->res[l] = ((l & 15) >= k && at[l - k]) ? w->val[l - k] : 0; }
                break;
            case K_SYNC:
                break;
            default:
                w->error = "unknown cross-lane operation"; return w->error;
            }
            for(int l = w0; l < w0 + WS; l++) if(at[l]) w->waiting[l] = false;
            progressed = true;
        }
        if(!progressed) {
            if(at_barrier != live) { w->error = "deadlock: nothing to serve and not every lane at the workgroup barrier"; return w->error; }
            w->collectives++;
            for(int l = 0; l < n; l++) if(!w->done[l]) w->waiting[l] = false;      // __syncthreads: everybody is here
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
static inline int   __shfl_xor(int v, int m)   { return (int)(uint32_t)emu::collective(emu::K_SHFL_XOR, (uint32_t)v, m); }
static inline float __shfl_xor(float v, int m) { return emu::u2f((uint32_t)emu::collective(emu::K_SHFL_XOR, emu::f2u(v), m)); }
static inline int   emu_readfirstlane(int v)   { return (int)(uint32_t)emu::collective(emu::K_FIRST, (uint32_t)v, 0); }
static inline int   emu_update_dpp(int old, int v, int ctrl, int, int, bool)
{
    if(ctrl >= 0x111 && ctrl <= 0x11f) return (int)(uint32_t)emu::collective(emu::K_DPP_SHR, (uint32_t)v, ctrl - 0x110);
    fprintf(stderr, "wave_emu: DPP control 0x%x not emulated\n", ctrl); abort();
    return old;
}
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
#define __builtin_amdgcn_update_dpp(o, v, c, r, b, bc) emu_update_dpp(o, v, c, r, b, bc)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)emu::collective(emu::K_SYNC, 0, 0))
static inline void __syncthreads() { (void)emu::collective(emu::K_WGSYNC, 0, 0); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int   __float_as_int(float f)   { return (int)emu::f2u(f); }
static inline unsigned __float_as_uint(float f) { return emu::f2u(f); }
static inline float __int_as_float(int i)     { return emu::u2f((uint32_t)i); }
template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }
