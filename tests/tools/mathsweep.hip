// tests/tools/mathsweep.hip -- TEST INFRASTRUCTURE (not part of libnavhip.so).
//
// Exhaustive sweeps of the one-argument exact-arithmetic building blocks of csrc/agent_math.h ON THE DEVICE, over
// every float of their domain: the functions whose bit-exactness rests on what the hardware's own instructions
// return (v_sqrt_f32, v_rsq_f32, the f64 FMA / conversion sequence of exp_f32_magic).  The host emulator of
// tests/hostsim cannot see these: it substitutes IEEE sqrtf for the native instructions.
//
// For function `which`, every argument with bit pattern in [lo, hi) is evaluated; the results of one chunk of
// 2^chunk_log2 consecutive bit patterns fold into a 64-bit checksum  sum_b (2 b + 1) * (result_bits(b) + 1)
// (mod 2^64: order independent, every argument weighted differently).  tests/tools/mathsweep_host.c folds what the
// REFERENCE computes for the same arguments -- glibc's (float)exp((double)a), sqrtf, the double-then-float
// expression of movement.c:1668 -- the same way; tests/test_mathsweep_gpu.py compares chunk by chunk and, on a
// difference, fetches the chunk's raw results (mathsweep_raw) to name the first argument.
//
// MS_FDIV is a two-operand function: the dividend sweeps every finite float, the divisor is a hash of its bits.
//
// For the two native approximations used behind safety margins (v_rsq_f32 in cone_contains_fast / cone_test_bf,
// v_sqrt_f32 in front of sqrt_rn_normal's fix-up) the sweep returns the largest error in units of the last place
// against the f64 evaluation instead (MS_RSQ_ULP, MS_SQRT_ULP; x 2^-16 fixed point).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "agent_math.h"

enum { MS_EXP = 0, MS_SQRT_RN = 1, MS_COH_T_F32 = 2, MS_COH_T_F64 = 3, MS_RSQ_ULP = 4, MS_SQRT_ULP = 5, MS_VLEN = 6, MS_FDIV = 7 };

__constant__ double c_ms_exp2_64[64] = { NH_EXP2_64_TABLE };

// the second operand of the division sweep: every exponent and sign, never Inf / NaN
__host__ __device__ static inline uint32_t ms_divisor_bits(uint32_t b)
{
    uint32_t h = b * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    if((h & 0x7f800000u) == 0x7f800000u) h ^= 0x40000000u;
    return h;
}

__device__ __forceinline__ uint32_t ms_eval(int which, uint32_t b, const double *tab)
{
    const float a = __uint_as_float(b);
    switch(which) {
    case MS_EXP:       return __float_as_uint(exp_f32_magic(a, tab));
    case MS_SQRT_RN:   return __float_as_uint(sqrt_rn_normal(a));
    case MS_COH_T_F32: return __float_as_uint(cohesion_t_f32(a));
    case MS_COH_T_F64: return __float_as_uint(cohesion_t_f64(a));
    case MS_VLEN:      return __float_as_uint(vlen(mkv(a, 0.0f)));      // (|a| through the product path incl. the cold IEEE branch)
    case MS_FDIV: {    // nh_fdiv(a, y(b)): the compiler's IEEE division (denormals kept), y a hash of the bit pattern
        const float q = nh_fdiv(a, __uint_as_float(ms_divisor_bits(b)));
        return q != q ? 0x7fc00000u : __float_as_uint(q);               // (one NaN: payloads are not the reference's business)
    }
    default:           return 0;
    }
}

// error of a native approximation in ulps of the exact result, x 65536, saturating
__device__ __forceinline__ uint32_t ms_ulp_err(int which, uint32_t b)
{
    const float s = __uint_as_float(b);
    const double exact = which == MS_RSQ_ULP ? 1.0 / __builtin_sqrt((double)s) : __builtin_sqrt((double)s);
    const float got = which == MS_RSQ_ULP ? nh_rsq_native(s) : nh_sqrt_native(s);
    const float ex32 = (float)exact;
    // ulp of the exact value's binade
    const float ulp = __uint_as_float((__float_as_uint(ex32) & 0x7f800000u)) * 0x1p-23f;
    const double e = __builtin_fabs((double)got - exact) / (double)ulp * 65536.0;
    return e >= 4294967295.0 ? 0xffffffffu : (uint32_t)e;
}

__global__ __launch_bounds__(256) void k_mathsweep(int which, uint32_t lo, uint32_t hi, int chunk_log2, unsigned long long *out)
{
    __shared__ double tab[64];
    __shared__ unsigned long long red[4];
    if(threadIdx.x < 64) tab[threadIdx.x] = c_ms_exp2_64[threadIdx.x];
    __syncthreads();
    const uint64_t c0 = (uint64_t)lo + ((uint64_t)blockIdx.x << chunk_log2);
    uint64_t c1 = c0 + (1ull << chunk_log2);
    if(c1 > hi) c1 = hi;
    unsigned long long acc = 0;
    if(which == MS_RSQ_ULP || which == MS_SQRT_ULP) {
        for(uint64_t b = c0 + threadIdx.x; b < c1; b += 256) {
            const unsigned long long e = ms_ulp_err(which, (uint32_t)b);
            acc = e > acc ? e : acc;
        }
#pragma unroll
        for(int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(acc, d); acc = o > acc ? o : acc; }
    }else{
        for(uint64_t b = c0 + threadIdx.x; b < c1; b += 256)
            acc += (2ull * b + 1ull) * ((unsigned long long)ms_eval(which, (uint32_t)b, tab) + 1ull);
#pragma unroll
        for(int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    }
    if((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if(threadIdx.x == 0) {
        unsigned long long r = red[0];
        for(int w = 1; w < 4; w++)
            r = (which == MS_RSQ_ULP || which == MS_SQRT_ULP) ? (red[w] > r ? red[w] : r) : r + red[w];
        out[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(256) void k_mathsweep_raw(int which, uint32_t lo, uint32_t n, uint32_t *out)
{
    __shared__ double tab[64];
    if(threadIdx.x < 64) tab[threadIdx.x] = c_ms_exp2_64[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if(i < n) out[i] = (which == MS_RSQ_ULP || which == MS_SQRT_ULP) ? ms_ulp_err(which, lo + i) : ms_eval(which, lo + i, tab);
}

extern "C" {

// checksums (or largest errors) of the chunks of [lo, hi): out[ceil((hi - lo) / 2^chunk_log2)], host memory.
// Returns 0, or the HIP error code.
int mathsweep_run(int which, uint32_t lo, uint32_t hi, int chunk_log2, unsigned long long *out)
{
    if(hi <= lo || chunk_log2 < 8 || chunk_log2 > 31) return -1;
    const uint64_t n = (uint64_t)hi - lo;
    const uint32_t nchunks = (uint32_t)((n + (1ull << chunk_log2) - 1) >> chunk_log2);
    unsigned long long *d = nullptr;
    hipError_t e = hipMalloc((void**)&d, (size_t)nchunks * 8);
    if(e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_mathsweep, dim3(nchunks), dim3(256), 0, 0, which, lo, hi, chunk_log2, d);
    e = hipGetLastError();
    if(e == hipSuccess) e = hipMemcpy(out, d, (size_t)nchunks * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return (int)e;
}

// the raw results of the n arguments from bit pattern lo on: out[n], host memory
int mathsweep_raw(int which, uint32_t lo, uint32_t n, uint32_t *out)
{
    if(n == 0) return -1;
    uint32_t *d = nullptr;
    hipError_t e = hipMalloc((void**)&d, (size_t)n * 4);
    if(e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_mathsweep_raw, dim3((n + 255) / 256), dim3(256), 0, 0, which, lo, n, d);
    e = hipGetLastError();
    if(e == hipSuccess) e = hipMemcpy(out, d, (size_t)n * 4, hipMemcpyDeviceToHost);
    hipFree(d);
    return (int)e;
}

}
