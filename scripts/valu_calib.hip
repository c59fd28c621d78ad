// scripts/valu_calib.hip -- what does one wave64 VALU instruction cost on an MI355X SIMD?
//
// DESIGN.md prices the kernels of this library against an instruction-issue floor; round 1 assumed
// 4 cycles per wave64 instruction, the MI355X guide says 2 (SIMD-32).  This measures it: every SIMD
// gets `waves` resident waves, each running `iters` x 64 independent instructions of one kind on 8
// separate register chains (no dependency stalls), and the kernel time gives
//     cycles per instruction per SIMD = time x clock x SIMDs / (waves x instructions per wave).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/valu_calib.hip -o scripts/valu_calib.bin
// Run on the GPU box: prints one JSON object (committed as profiles/archive/r02_valu_calib.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAINS 8
#define UNROLL 8            // 64 instructions per loop body

enum { K_FMA_F32 = 0, K_XOR_B32, K_PK_FMA_F32, K_FMA_F64, K_SHL_B64, K_MUL_LO_U32, K_DPP_MOV, K_SQRT_F32,
       K_CNDMASK, K_ADD_F64, K_MAD_U64_U32, K_COUNT };
static const char *k_names[K_COUNT] = {"v_fma_f32", "v_xor_b32", "v_pk_fma_f32", "v_fma_f64", "v_lshlrev_b64",
                                       "v_mul_lo_u32", "v_mov_b32_dpp(row_shr)", "v_sqrt_f32", "v_cndmask_b32",
                                       "v_add_f64", "v_mad_u64_u32"};

template <int KIND>
__global__ __launch_bounds__(256) void k_calib(int iters, float seed, float *sink)
{
    float a[CHAINS]; double d[CHAINS]; uint32_t u[CHAINS]; uint64_t q[CHAINS];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CHAINS];
    for(int c = 0; c < CHAINS; c++) {
        a[c] = seed + c + threadIdx.x; d[c] = a[c]; u[c] = (uint32_t)(a[c] * 977.0f) | 1u; q[c] = u[c] * 0x9E3779B97F4A7C15ull;
        p[c] = f2{a[c], a[c] + 0.5f};
    }
    const float m = 0.999f, b = 0.001f;
    for(int it = 0; it < iters; it++) {
#pragma unroll
        for(int r = 0; r < UNROLL; r++) {
#pragma unroll
            for(int c = 0; c < CHAINS; c++) {
                if(KIND == K_FMA_F32)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
                if(KIND == K_XOR_B32)    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS] | 1u));
                if(KIND == K_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(f2{m, m}), "v"(f2{b, b}));
                if(KIND == K_FMA_F64)    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"((double)m), "v"((double)b));
                if(KIND == K_ADD_F64)    asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"((double)b));
                if(KIND == K_SHL_B64)    asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[c]));
                if(KIND == K_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS] | 1u));
                if(KIND == K_DPP_MOV)    asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[c]));
                if(KIND == K_SQRT_F32)   asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[c]));
                if(KIND == K_CNDMASK)    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
                if(KIND == K_MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[c]) : "v"(u[c]), "v"(u[(c + 1) % CHAINS]) : "vcc");
            }
        }
    }
    float s = 0;
    for(int c = 0; c < CHAINS; c++) s += a[c] + (float)d[c] + (float)u[c] + (float)q[c] + p[c].x + p[c].y;
    if(s == 12345.678f) sink[0] = s;          // keeps everything alive
}

template <int KIND>
static double run(int grid, int iters, float *sink, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_calib<KIND>, dim3(grid), dim3(256), 0, 0, iters, 1.0f, sink);
    hipDeviceSynchronize();
    // the fastest of three timed batches: a box whose clocks have not ramped up yet (or that is power
    // capped for a moment) reports 7 cycles where every other session reports 2.6
    double best = 1e30;
    for(int b = 0; b < 3; b++) {
        hipEventRecord(e0, 0);
        for(int r = 0; r < reps; r++)
            hipLaunchKernelGGL(k_calib<KIND>, dim3(grid), dim3(256), 0, 0, iters, 1.0f, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if(ms * 1e-3 / reps < best) best = ms * 1e-3 / reps;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no GPU\n"); return 1; }
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    const double clock_hz = prop.clockRate * 1e3;               // kHz -> Hz (peak engine clock)
    float *sink; hipMalloc(&sink, 4);
    // (warm the clocks up)
    for(int r = 0; r < 40; r++) hipLaunchKernelGGL(k_calib<K_FMA_F32>, dim3(simds * 2), dim3(256), 0, 0, 2000, 1.0f, sink);
    hipDeviceSynchronize();
    const int iters = 2000, reps = 5;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f, \"insts_per_wave\": %d, \"results\": {",
           prop.gcnArchName, cus, clock_hz / 1e6, iters * UNROLL * CHAINS);
    for(int wps = 1; wps <= 8; wps *= 2) {                       // resident waves per SIMD
        const int grid = cus * wps;                              // 256 threads = 4 waves = one per SIMD of a CU
        double t[K_COUNT];
        t[K_FMA_F32] = run<K_FMA_F32>(grid, iters, sink, reps);
        t[K_XOR_B32] = run<K_XOR_B32>(grid, iters, sink, reps);
        t[K_PK_FMA_F32] = run<K_PK_FMA_F32>(grid, iters, sink, reps);
        t[K_FMA_F64] = run<K_FMA_F64>(grid, iters, sink, reps);
        t[K_SHL_B64] = run<K_SHL_B64>(grid, iters, sink, reps);
        t[K_MUL_LO_U32] = run<K_MUL_LO_U32>(grid, iters, sink, reps);
        t[K_DPP_MOV] = run<K_DPP_MOV>(grid, iters, sink, reps);
        t[K_SQRT_F32] = run<K_SQRT_F32>(grid, iters, sink, reps);
        t[K_CNDMASK] = run<K_CNDMASK>(grid, iters, sink, reps);
        t[K_ADD_F64] = run<K_ADD_F64>(grid, iters, sink, reps);
        t[K_MAD_U64_U32] = run<K_MAD_U64_U32>(grid, iters, sink, reps);
        printf("%s\"waves_per_simd_%d\": {", wps == 1 ? "" : ", ", wps);
        for(int k = 0; k < K_COUNT; k++) {
            const double insts_per_simd = (double)wps * iters * UNROLL * CHAINS;
            printf("%s\"%s\": %.3f", k ? ", " : "", k_names[k], t[k] * clock_hz / insts_per_simd);
        }
        printf("}");
    }
    printf("}, \"unit\": \"cycles per wave64 instruction per SIMD at the reported peak clock\"}\n");
    return 0;
}
