// stream_pingpong_probe.hip -- the cost of a cross-stream hand-over (event record on A, wait on B, kernel on B, and
// back) for every pair of a set of streams.  Developer probe behind DESIGN.md's note on W7, second part: do two
// streams whose hardware queues sit on the same pipe of the command processor hand over more slowly?
//   hipcc --offload-arch=gfx950 -O2 scripts/stream_pingpong_probe.hip -o scripts/stream_pingpong_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void k_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while(wall_clock64() - t0 < ticks) { }
}

static hipStream_t mk(char kind)
{
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if(kind == 'h')      CHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    else if(kind == 'l') CHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo));
    else if(kind == 'n') CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else {
        hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
        uint32_t mask[32] = {0};
        for(int c = 0; c < p.multiProcessorCount; c++) mask[c >> 5] |= 1u << (c & 31);
        CHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)((p.multiProcessorCount + 31) / 32), mask));
    }
    return s;
}

// `rounds` times: A: 10-us kernel, record; B: wait, 10-us kernel, record; A: wait.  Everything enqueued up front.
// With a third stream C given: C runs a chain of its own beside it (20-us kernels, C only) -- a bystander.
static double pingpong_us(hipStream_t a, hipStream_t b, int rounds, hipStream_t c = nullptr)
{
    static hipEvent_t ea = nullptr, eb = nullptr;
    if(!ea) { CHK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&eb, hipEventDisableTiming)); }
    CHK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for(int r = 0; r < rounds; r++) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 1000LL);
        CHK(hipEventRecord(ea, a));
        CHK(hipStreamWaitEvent(b, ea, 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, 1000LL);
        CHK(hipEventRecord(eb, b));
        CHK(hipStreamWaitEvent(a, eb, 0));
        if(c) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, c, 2000LL);
    }
    CHK(hipStreamSynchronize(a)); CHK(hipStreamSynchronize(b));
    if(c) CHK(hipStreamSynchronize(c));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
}

static void matrix(const char *title, const std::vector<hipStream_t> &S, const std::string &kinds)
{
    printf("%s  (us per round trip A -> B -> A of two 10-us kernels)\n      ", title);
    for(size_t j = 0; j < S.size(); j++) printf("  %c%-2zu", kinds[j], j);
    printf("\n");
    for(size_t i = 0; i < S.size(); i++) {
        printf("  %c%-2zu ", kinds[i], i);
        for(size_t j = 0; j < S.size(); j++) {
            if(j == i) { printf("    ."); continue; }
            pingpong_us(S[i], S[j], 5);
            printf(" %4.0f", pingpong_us(S[i], S[j], 40));
        }
        printf("\n");
    }
    fflush(stdout);
}

int main(int argc, char **argv)
{
    CHK(hipSetDevice(0));
    const int npool = argc > 1 ? atoi(argv[1]) : 0;      // pooled streams a host would hold (torch: 32 + 32)
    std::vector<hipStream_t> pool;
    for(int i = 0; i < npool; i++) { pool.push_back(mk('h')); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, pool.back(), 10LL); }
    for(int i = 0; i < npool; i++) { pool.push_back(mk('n')); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, pool.back(), 10LL); }
    CHK(hipDeviceSynchronize());
    {
        std::vector<hipStream_t> S; std::string kinds;
        for(int i = 0; i < 10; i++) { S.push_back(mk('m')); kinds += 'm'; }
        matrix("== 10 masked streams", S, kinds);
        // a bystander chain on a third stream
        printf("with a bystander chain on m2: m0<->m1 %.0f  m0<->m4 %.0f  m0<->m5 %.0f  m0<->m8 %.0f; bystander m6: m0<->m4 %.0f  m1<->m5 %.0f\n",
               pingpong_us(S[0], S[1], 40, S[2]), pingpong_us(S[0], S[4], 40, S[2]), pingpong_us(S[0], S[5], 40, S[2]), pingpong_us(S[0], S[8], 40, S[2]),
               pingpong_us(S[0], S[4], 40, S[6]), pingpong_us(S[1], S[5], 40, S[6]));
        if(npool) {
            std::vector<hipStream_t> T = {pool[0], pool[1], pool[2], pool[3], pool[4], pool[5], S[0], S[1], S[2], S[3]};
            matrix("== pooled high-priority streams of the host and masked ones", T, "hhhhhhmmmm");
        }
        for(auto s : S) CHK(hipStreamDestroy(s));
    }
    {
        std::vector<hipStream_t> S; std::string kinds;
        for(int i = 0; i < 8; i++) { S.push_back(mk('h')); kinds += 'h'; }
        matrix("== 8 more high-priority streams", S, kinds);
        for(auto s : S) CHK(hipStreamDestroy(s));
    }
    for(auto s : pool) CHK(hipStreamDestroy(s));
    return 0;
}
