// stream_flag_probe.hip -- what a hand-over between two streams costs when it is a queue barrier (event record on A,
// event wait on B) and when it is a word in device memory (a one-lane kernel on B that ends when A's kernel has
// stored a sequence number).  Four masked streams; every pair; the round trip A -> B -> A of two 10-us kernels.
//   hipcc --offload-arch=gfx950 -O2 scripts/stream_flag_probe.hip -o scripts/stream_flag_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void k_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while(wall_clock64() - t0 < ticks) { }
}
// the producer's last kernel stores the number itself
__global__ void k_spin_signal(long long ticks, int *flag, int seq)
{
    const long long t0 = wall_clock64();
    while(wall_clock64() - t0 < ticks) { }
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_signal(int *flag, int seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(const int *flag, int want, int *status)
{
    const long long t0 = wall_clock64();
    while(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
        __builtin_amdgcn_s_sleep(4);
        if(wall_clock64() - t0 > 200000000LL) { *status = 1; break; }          // 2 s (100 MHz)
    }
}

static hipStream_t mk()
{
    hipStream_t s = nullptr;
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    uint32_t mask[32] = {0};
    for(int c = 0; c < p.multiProcessorCount; c++) mask[c >> 5] |= 1u << (c & 31);
    CHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)((p.multiProcessorCount + 31) / 32), mask));
    return s;
}

static int *g_flags, *g_status, g_seq;

// mode 0: events; 1: k_signal + k_wait (two more launches per hand-over); 2: the producer stores, k_wait
static double pingpong_us(hipStream_t a, hipStream_t b, int rounds, int mode)
{
    static hipEvent_t ea = nullptr, eb = nullptr;
    if(!ea) { CHK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&eb, hipEventDisableTiming)); }
    CHK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for(int r = 0; r < rounds; r++) {
        const int n = ++g_seq;
        if(mode == 0) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 1000LL);
            CHK(hipEventRecord(ea, a));
            CHK(hipStreamWaitEvent(b, ea, 0));
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, 1000LL);
            CHK(hipEventRecord(eb, b));
            CHK(hipStreamWaitEvent(a, eb, 0));
        }else if(mode == 1) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 1000LL);
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, a, g_flags, n);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, b, (const int*)g_flags, n, g_status);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, 1000LL);
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, b, g_flags + 32, n);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, a, (const int*)(g_flags + 32), n, g_status);
        }else{
            hipLaunchKernelGGL(k_spin_signal, dim3(1), dim3(64), 0, a, 1000LL, g_flags, n);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, b, (const int*)g_flags, n, g_status);
            hipLaunchKernelGGL(k_spin_signal, dim3(1), dim3(64), 0, b, 1000LL, g_flags + 32, n);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, a, (const int*)(g_flags + 32), n, g_status);
        }
    }
    CHK(hipStreamSynchronize(a)); CHK(hipStreamSynchronize(b));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
}

int main()
{
    CHK(hipSetDevice(0));
    CHK(hipMalloc((void**)&g_flags, 64 * sizeof(int)));
    CHK(hipMemset(g_flags, 0, 64 * sizeof(int)));
    CHK(hipHostMalloc((void**)&g_status, sizeof(int), hipHostMallocMapped));
    *g_status = 0;
    std::vector<hipStream_t> S;
    for(int i = 0; i < 4; i++) S.push_back(mk());
    const char *names[3] = {"event record + event wait", "k_signal + k_wait (two launches more)", "the producer stores + k_wait"};
    printf("round trip A -> B -> A of two 10-us kernels, us (20 = no hand-over cost at all)\n");
    // a chain on ONE stream for scale: what two kernels and two one-lane kernels cost back to back
    {
        CHK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for(int r = 0; r < 40; r++) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, S[0], 1000LL); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, S[0], 1000LL); }
        CHK(hipStreamSynchronize(S[0]));
        printf("  one stream, two 10-us kernels back to back: %.1f\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 40);
    }
    for(int mode = 0; mode < 3; mode++) {
        printf("  %-40s", names[mode]);
        for(int i = 0; i < 4; i++) for(int j = i + 1; j < 4; j++) {
            pingpong_us(S[i], S[j], 5, mode);
            printf(" m%d-m%d %5.1f", i, j, pingpong_us(S[i], S[j], 40, mode));
        }
        printf("\n");
    }
    // eight masked streams: queue k sits on pipe k mod 4 (scripts/stream_pingpong_probe.hip) -- what does a pair on ONE pipe cost
    // with words instead of events?
    {
        std::vector<hipStream_t> T;
        for(int i = 0; i < 8; i++) T.push_back(mk());
        for(int mode = 0; mode < 3; mode += 2) {
            printf("== eight more masked streams, %s\n      ", names[mode]);
            for(int j = 0; j < 8; j++) printf("   m%d", j);
            printf("\n");
            for(int i = 0; i < 8; i++) {
                printf("  m%d  ", i);
                for(int j = 0; j < 8; j++) {
                    if(i == j) { printf("    ."); continue; }
                    pingpong_us(T[i], T[j], 4, mode);
                    printf(" %4.0f", pingpong_us(T[i], T[j], 30, mode));
                }
                printf("\n");
            }
        }
        for(auto t : T) CHK(hipStreamDestroy(t));
    }
    printf("status %d (1 = a wait timed out)\n", *g_status);
    for(auto s : S) CHK(hipStreamDestroy(s));
    return 0;
}
