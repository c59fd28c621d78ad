/* tests/tools/mathsweep_host.c -- TEST INFRASTRUCTURE: what the REFERENCE computes for every float argument of the
 * sweeps of tests/tools/mathsweep.hip, folded into the same per-chunk checksums (see there).  Plain C, libm, threads.
 *   0 MS_EXP        (float)exp((double)a)                       movement.c:1671, :1731 (libm's double exp on a float)
 *   1 MS_SQRT_RN    (float)sqrt((double)s) == sqrtf(s)          PFM_Vec2_Len, pf_math.c:84
 *   2 MS_COH_T_F32  (float)(((double)len - 50.0f*0.75) / 50.0f) movement.c:1668
 *   3 MS_COH_T_F64  the same expression
 *   6 MS_VLEN       sqrtf(a * a)  (PFM_Vec2_Len of (a, 0))
 *   7 MS_FDIV       a / y(bits of a)   (C's float division; y = the hash of mathsweep.hip's ms_divisor_bits)
 * gcc -O2 -ffp-contract=off -shared -fPIC mathsweep_host.c -o _mathsweep_host.so -lm -lpthread */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static uint32_t divisor_bits(uint32_t b)
{
    uint32_t h = b * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    if((h & 0x7f800000u) == 0x7f800000u) h ^= 0x40000000u;
    return h;
}

static uint32_t ref_eval(int which, uint32_t b)
{
    const float a = u2f(b);
    switch(which) {
    case 0: return f2u((float)exp((double)a));
    case 1: return f2u((float)sqrt((double)a));
    case 2:
    case 3: { float t = (a - 50.0f * 0.75) / 50.0f; return f2u(t); }
    case 6: { float s = a * a + 0.0f * 0.0f; return f2u((float)sqrt((double)s)); }
    case 7: { volatile float y = u2f(divisor_bits(b)); float q = a / y; return q != q ? 0x7fc00000u : f2u(q); }
    default: return 0;
    }
}

struct job { int which, chunk_log2; uint32_t lo, hi; uint32_t c0, c1; unsigned long long *out; };

static void *run(void *p)
{
    struct job *j = (struct job*)p;
    for(uint32_t c = j->c0; c < j->c1; c++) {
        uint64_t b0 = (uint64_t)j->lo + ((uint64_t)c << j->chunk_log2), b1 = b0 + (1ull << j->chunk_log2);
        if(b1 > j->hi) b1 = j->hi;
        unsigned long long acc = 0;
        for(uint64_t b = b0; b < b1; b++)
            acc += (2ull * b + 1ull) * ((unsigned long long)ref_eval(j->which, (uint32_t)b) + 1ull);
        j->out[c] = acc;
    }
    return 0;
}

int mathsweep_host(int which, uint32_t lo, uint32_t hi, int chunk_log2, unsigned long long *out, int nthreads)
{
    if(hi <= lo || chunk_log2 < 8 || chunk_log2 > 31) return -1;
    const uint64_t n = (uint64_t)hi - lo;
    const uint32_t nchunks = (uint32_t)((n + (1ull << chunk_log2) - 1) >> chunk_log2);
    if(nthreads < 1) nthreads = 1;
    if(nthreads > 64) nthreads = 64;
    pthread_t th[64];
    struct job jb[64];
    const uint32_t per = (nchunks + nthreads - 1) / nthreads;
    int started = 0;
    for(int t = 0; t < nthreads; t++) {
        uint32_t c0 = (uint32_t)t * per, c1 = c0 + per > nchunks ? nchunks : c0 + per;
        if(c0 >= nchunks) break;
        jb[t] = (struct job){which, chunk_log2, lo, hi, c0, c1, out};
        pthread_create(&th[t], 0, run, &jb[t]);
        started++;
    }
    for(int t = 0; t < started; t++) pthread_join(th[t], 0);
    return 0;
}

uint32_t mathsweep_host_one(int which, uint32_t b) { return ref_eval(which, b); }
