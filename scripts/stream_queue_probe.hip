// stream_queue_probe.hip -- which HIP streams of one process run kernels CONCURRENTLY on this runtime, and which ones
// share a hardware queue (their kernels serialise)?  Developer probe behind DESIGN.md's note on W7 (the tick of one and
// the same world cost X or 2-3X depending on what the process had created before).
//   hipcc --offload-arch=gfx950 -O2 scripts/stream_queue_probe.hip -o scripts/stream_queue_probe.bin
// A pair of streams "overlaps" when two 200-us one-wave kernels, one on each, take ~200 us together and not ~400.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void k_spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    while(wall_clock64() - t0 < ticks) { }
    if(sink && threadIdx.x == 1000) *sink = 1;
}

static long long g_ticks;     // wall_clock64 ticks for ~200 us (100 MHz constant clock on gfx9: 20 000)

static double pair_us(hipStream_t a, hipStream_t b)
{
    CHK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, g_ticks, (int*)nullptr);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, g_ticks, (int*)nullptr);
    CHK(hipStreamSynchronize(a)); CHK(hipStreamSynchronize(b));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

static hipStream_t mk(const char *kind)
{
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if(kind[0] == 'h')      CHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    else if(kind[0] == 'l') CHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo));
    else if(kind[0] == 'n') CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else {                  // 'm': a CU mask with every compute unit in it
        hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
        uint32_t mask[32] = {0};
        for(int c = 0; c < p.multiProcessorCount; c++) mask[c >> 5] |= 1u << (c & 31);
        CHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)((p.multiProcessorCount + 31) / 32), mask));
    }
    return s;
}

static void matrix(const char *title, const std::vector<hipStream_t> &S, const std::string &kinds)
{
    printf("%s  (pairwise time of two 200-us kernels, us; ~200 = concurrent, ~400 = one hardware queue)\n      ", title);
    for(size_t j = 0; j < S.size(); j++) printf("  %c%-2zu", kinds[j], j);
    printf("\n");
    for(size_t i = 0; i < S.size(); i++) {
        printf("  %c%-2zu ", kinds[i], i);
        for(size_t j = 0; j < S.size(); j++) {
            if(j <= i) { printf("    ."); continue; }
            pair_us(S[i], S[j]);
            printf(" %4.0f", pair_us(S[i], S[j]));
        }
        printf("\n");
    }
    fflush(stdout);
}

int main(int argc, char **argv)
{
    CHK(hipSetDevice(0));
    g_ticks = 20000;
    {   // calibrate: one kernel alone
        hipStream_t s = mk("n");
        pair_us(s, s);
        const double two = pair_us(s, s);          // same stream: serial = 2 x one kernel
        g_ticks = (long long)(g_ticks * 400.0 / two);
        printf("calibration: two kernels on ONE stream %.0f us -> %lld ticks per 200 us; again: %.0f us\n", two, g_ticks, pair_us(s, s));
        CHK(hipStreamDestroy(s));
    }
    // 1. eight streams of each pooled kind
    for(const char *kind : {"h", "n", "l", "m"}) {
        std::vector<hipStream_t> S; std::string kinds;
        for(int i = 0; i < 8; i++) { S.push_back(mk(kind)); kinds += kind[0]; }
        matrix((std::string("== 8 streams of kind ") + kind).c_str(), S, kinds);
        for(auto s : S) CHK(hipStreamDestroy(s));
    }
    // 2. a process that already holds 32 high-priority + 32 normal streams (torch's pools), then the library's set
    std::vector<hipStream_t> pool;
    for(int i = 0; i < 32; i++) pool.push_back(mk("h"));
    for(int i = 0; i < 32; i++) pool.push_back(mk("n"));
    for(int rep = 0; rep < 8; rep++) {
        // caller's stream = pool stream `rep` (high priority); the library: aux0 high, aux1 low, field stream masked, ctx normal
        std::vector<hipStream_t> S = {pool[rep], mk("h"), mk("l"), mk("m"), mk("n")};
        char title[128]; snprintf(title, sizeof title, "== rep %d: caller = pool stream h%d | aux0 h | aux1 l | field m | ctx n", rep, rep);
        matrix(title, S, "hhlmn");
        for(size_t i = 1; i < S.size(); i++) CHK(hipStreamDestroy(S[i]));
    }
    // 3. the same with masked side streams
    for(int rep = 0; rep < 8; rep++) {
        std::vector<hipStream_t> S = {pool[rep], mk("m"), mk("m"), mk("m"), mk("n")};
        char title[128]; snprintf(title, sizeof title, "== rep %d (masked side streams): caller = pool stream h%d | aux0 m | aux1 m | field m | ctx n", rep, rep);
        matrix(title, S, "hmmmn");
        for(size_t i = 1; i < S.size(); i++) CHK(hipStreamDestroy(S[i]));
    }
    // 4. how many live masked streams before two of them stop overlapping (hardware queue slots)
    {
        std::vector<hipStream_t> M;
        for(int k = 0; k < 40; k++) {
            M.push_back(mk("m"));
            pair_us(M[0], M.back());          // (first use creates the queue)
            if(k >= 1 && (k % 4 == 3 || k > 20)) printf("live masked streams %2d: pair(first, last) %.0f us   pair(caller h0, last) %.0f us\n",
                                                       k + 1, pair_us(M[0], M.back()), pair_us(pool[0], M.back()));
        }
        for(auto s : M) CHK(hipStreamDestroy(s));
        fflush(stdout);
    }
    for(auto s : pool) CHK(hipStreamDestroy(s));
    return 0;
}
