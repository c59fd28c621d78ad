// tests/tools/mathsweep_emul.cpp -- TEST INFRASTRUCTURE: csrc/agent_math.h's own algorithms compiled for the host
// (-DNH_HOSTSIM: IEEE fma / sqrt of libm instead of the device instructions), folded into the per-chunk checksums of
// tests/tools/mathsweep.hip.  What the CPU suite can say about exp_f32_magic and cohesion_t_*: their f64 / f32
// operation sequences reproduce the reference's libm results for EVERY float argument, given IEEE arithmetic.  The
// device's own instructions are swept on the device (tests/test_mathsweep_gpu.py).
#include <stdint.h>
#include <pthread.h>
#include "agent_math.h"

static const double g_tab[64] = { NH_EXP2_64_TABLE };

static uint32_t emul_eval(int which, uint32_t b)
{
    const float a = nh_u2f(b);
    switch(which) {
    case 0: return nh_f2u(exp_f32_magic(a, g_tab));
    case 1: return nh_f2u(sqrt_rn_normal(a));
    case 2: return nh_f2u(cohesion_t_f32(a));
    case 3: return nh_f2u(cohesion_t_f64(a));
    case 6: return nh_f2u(vlen(mkv(a, 0.0f)));
    default: return 0;
    }
}

struct job { int which, chunk_log2; uint32_t lo, hi, c0, c1; unsigned long long *out; };

static void *run(void *p)
{
    job *j = (job*)p;
    for(uint32_t c = j->c0; c < j->c1; c++) {
        uint64_t b0 = (uint64_t)j->lo + ((uint64_t)c << j->chunk_log2), b1 = b0 + (1ull << j->chunk_log2);
        if(b1 > j->hi) b1 = j->hi;
        unsigned long long acc = 0;
        for(uint64_t b = b0; b < b1; b++)
            acc += (2ull * b + 1ull) * ((unsigned long long)emul_eval(j->which, (uint32_t)b) + 1ull);
        j->out[c] = acc;
    }
    return 0;
}

extern "C" int mathsweep_emul(int which, uint32_t lo, uint32_t hi, int chunk_log2, unsigned long long *out, int nthreads)
{
    if(hi <= lo || chunk_log2 < 8 || chunk_log2 > 31) return -1;
    const uint64_t n = (uint64_t)hi - lo;
    const uint32_t nchunks = (uint32_t)((n + (1ull << chunk_log2) - 1) >> chunk_log2);
    if(nthreads < 1) nthreads = 1;
    if(nthreads > 64) nthreads = 64;
    pthread_t th[64];
    job jb[64];
    const uint32_t per = (nchunks + nthreads - 1) / nthreads;
    int started = 0;
    for(int t = 0; t < nthreads; t++) {
        uint32_t c0 = (uint32_t)t * per, c1 = c0 + per > nchunks ? nchunks : c0 + per;
        if(c0 >= nchunks) break;
        jb[t] = job{which, chunk_log2, lo, hi, c0, c1, out};
        pthread_create(&th[t], 0, run, &jb[t]);
        started++;
    }
    for(int t = 0; t < started; t++) pthread_join(th[t], 0);
    return 0;
}

extern "C" uint32_t mathsweep_emul_one(int which, uint32_t b) { return emul_eval(which, b); }
