/* oracle/ref/ref_support.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Engine-runtime services that the reference's navigation / movement
 * translation units expect from the rest of permafrost-engine (allocator,
 * profiler, fiber scheduler, string helpers, SDL atomics, shared khash
 * instantiations).  They are given trivial single-threaded bodies so that
 * the reference sources can be linked, unmodified and in place from
 * /root/reference/src, into oracle/_ref/libpfref.so.
 *
 * Nothing here is shipped or measured as product code.
 */
#include <stdarg.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <SDL_atomic.h>
#include <SDL_thread.h>
#include "lib/public/khash.h"
#include "game/public/game.h"   /* KHASH_DECLARE(entity, ...) */
#include "game/gamestate.h"     /* KHASH_DECLARE(id / range)  */

/* game.c:116-118 instantiates these three tables for the whole engine */
__KHASH_IMPL(entity,  extern, khint32_t, uint32_t, 0, kh_int_hash_func, kh_int_hash_equal)
__KHASH_IMPL(id,      extern, khint32_t, int,      1, kh_int_hash_func, kh_int_hash_equal)
__KHASH_IMPL(range,   extern, khint32_t, float,    1, kh_int_hash_func, kh_int_hash_equal)

unsigned long g_frame_idx = 0;
SDL_threadID  g_main_thread_id = 1;

SDL_threadID SDL_ThreadID(void) { return g_main_thread_id; }

int SDL_AtomicSet(SDL_atomic_t *a, int v) { int o = a->value; a->value = v; return o; }
int SDL_AtomicGet(SDL_atomic_t *a)        { return a->value; }
SDL_bool SDL_AtomicCAS(SDL_atomic_t *a, int oldv, int newv)
{
    if(a->value != oldv) return SDL_FALSE;
    a->value = newv;
    return SDL_TRUE;
}
int    SDL_GetCPUCount(void) { return 1; }
Uint32 SDL_GetTicks(void)    { return 0; }

void *Mem_Malloc(size_t n)                                         { return malloc(n); }
void *Mem_Calloc(size_t c, size_t n)                               { return calloc(c, n); }
void *Mem_Realloc(void *p, size_t n)                               { return realloc(p, n); }
void *Mem_MallocTagged(size_t n, uint16_t sys, uint16_t sub)       { (void)sys; (void)sub; return malloc(n); }
void *Mem_CallocTagged(size_t c, size_t n, uint16_t s, uint16_t b) { (void)s; (void)b; return calloc(c, n); }
void *Mem_ReallocTagged(void *p, size_t n, uint16_t s, uint16_t b) { (void)s; (void)b; return realloc(p, n); }
void  Mem_Free(void *p)                                            { free(p); }
void  Mem_PushScope(uint16_t sys, uint16_t sub)                    { (void)sys; (void)sub; }
void  Mem_PopScope(void)                                           { }

void Perf_Push(const char *name)  { (void)name; }
void Perf_Pop(const char **out)   { if(out) *out = NULL; }

/* sched.c: the harness has no fibers.  A task "created" here runs to completion at once on the caller's
 * stack (what Sched_RunSync would make of it) and its future is complete -- the all-CPU control of the
 * asynchronous field batch (nav.c:3824: Sched_Create(field_task); :2062 field_join_work). */
#include "sched.h"
uint32_t Sched_Create(int prio, task_func_t code, void *arg, const char *name, struct future *result, int flags)
{
    (void)prio; (void)name; (void)flags;
    struct result r = code(arg);
    if(result) {
        result->res = r;
        SDL_AtomicSet(&result->status, FUTURE_COMPLETE);
    }
    return 1;
}
bool     Sched_FutureIsReady(const struct future *future) { return ((SDL_atomic_t*)&future->status)->value == FUTURE_COMPLETE; }
bool     Sched_RunSync(uint32_t tid) { (void)tid; return true; }
bool     Sched_UsingBigStack(void) { return true; }
void     Sched_TryYield(void)      { }
uint32_t Sched_ActiveTID(void)     { return 0; /* NULL_TID: satisfies FC_ASSERT_NAV_TASK */ }

int pf_snprintf(char *str, size_t size, const char *format, ...)
{
    va_list ap;
    va_start(ap, format);
    int r = vsnprintf(str, size, format, ap);
    va_end(ap);
    return r;
}

size_t pf_strlcpy(char *dest, const char *src, size_t size)
{
    size_t n = strlen(src);
    if(size) {
        size_t c = n < size - 1 ? n : size - 1;
        memcpy(dest, src, c);
        dest[c] = '\0';
    }
    return n;
}

char *pf_strlcat(char *dest, const char *src, size_t size)
{
    size_t dl = strlen(dest);
    if(dl < size)
        pf_strlcpy(dest + dl, src, size - dl);
    return dest;
}

/* game.c:2744 -- the diplomacy table is game state; the harness makes it an explicit input so that
 * attacking paths (field_tile_passable_no_enemies, field.c:179) can be exercised */
static uint16_t s_enemy_masks[16];

uint16_t G_GetEnemyFactions(int faction_id)
{
    return s_enemy_masks[faction_id & 15];
}

void pfref_set_enemy_factions(int faction_id, unsigned mask)
{
    s_enemy_masks[faction_id & 15] = (uint16_t)mask;
}

void pfref_stub_abort(const char *name)
{
    fprintf(stderr, "pfref: engine symbol '%s' is stubbed out in the oracle harness "
                    "and must not be reached\n", name);
    abort();
}
