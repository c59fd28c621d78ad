// tests/hostsim/fakehip/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// What the kernel translation units of libnavhip need from <hip/hip_runtime.h>, on top of the lockstep emulator of
// ../../wave_emu.h: a kernel launch runs the grid block by block, every block as fibers in lockstep (one wave of 64
// lanes or a workgroup of up to four); __shared__ variables are function statics (one block is alive at a time);
// device memory is host memory; streams and events are no-ops (everything is synchronous and in order).
#pragma once
#include <math.h>
#include "wave_emu.h"

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
typedef struct emu_stream_t *hipStream_t;
typedef struct emu_event_t *hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

#define __shared__ static
#define __constant__ static
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) x
#define amdgpu_waves_per_eu(...) unused      /* (an occupancy hint for the device compiler) */

namespace emu {
static thread_local dim3 blockIdx_emu, gridDim_emu, blockDim_emu;
template <typename F> static void lane_body(void *p) { (*(F*)p)(); }
template <typename F> static void launch(dim3 grid, dim3 block, F fn)
{
    gridDim_emu = grid; blockDim_emu = block;
    for(unsigned b = 0; b < grid.x; b++) {
        blockIdx_emu = dim3(b);
        const char *err = run((int)block.x, lane_body<F>, &fn);
        if(err) { fprintf(stderr, "emulated launch: block %u: %s\n", b, err); abort(); }
    }
}
}  // namespace emu
#define blockIdx emu::blockIdx_emu
#define gridDim  emu::gridDim_emu
#define blockDim emu::blockDim_emu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) (emu_scan_args(#kernel, __VA_ARGS__), emu::launch(grid, block, [&]() { kernel(__VA_ARGS__); }))

enum { hipErrorNotReady = 600, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2,
       hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
typedef void *hipDeviceptr_t;
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };
struct hipPointerAttribute_t { int type; };
static inline const char *hipGetErrorString(hipError_t) { return "emulated HIP runtime"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 64; return hipSuccess; }      /* (any rank of a multi-process test finds "its" device) */
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
/* "Device" allocations are remembered, so that a copy can be held against them: a copy whose direction flag
 * contradicts where its operands live (device memory as the source of a host-to-device copy, or as the destination of
 * a device-to-host one), or that runs past the end of a device allocation, fails here as it would on a GPU -- host
 * memory and device memory are the same thing on the emulator, and such a call would otherwise just work.  (Memory
 * the library did not allocate -- a test's numpy array standing in for a device buffer -- is not judged.) */
#include <map>
#include <mutex>
#include <stdio.h>
#include <stdint.h>
inline std::map<uintptr_t, size_t> &emu_dev_allocs() { static std::map<uintptr_t, size_t> m; return m; }
inline std::mutex &emu_dev_mutex() { static std::mutex m; return m; }
inline std::map<uintptr_t, size_t> &emu_pinned_allocs() { static std::map<uintptr_t, size_t> m; return m; }   /* hipHostMalloc */
/* 0: not in a device allocation, 1: inside one, -1: starts in one and runs past its end */
static inline int emu_dev_range(const void *p, size_t n) {
    std::lock_guard<std::mutex> g(emu_dev_mutex());
    auto &m = emu_dev_allocs();
    auto it = m.upper_bound((uintptr_t)p);
    if(it == m.begin()) return 0;
    --it;
    if((uintptr_t)p >= it->first + it->second) return 0;
    return (uintptr_t)p + n <= it->first + it->second ? 1 : -1;
}
/* EMU_STRICT_POINTERS=1 (the emulated runs of the host-buffer tests set it): a kernel handed a pointer to ordinary
 * host memory works on the emulator and faults on a GPU.  Every host operand of a copy (the source of a host-to-device
 * copy, the destination of a device-to-host one) is remembered as HOST memory; every 8-byte word of a kernel's
 * arguments -- plain pointers and the pointers inside argument structs alike -- that points into such a range, and
 * not into a device allocation or pinned memory, aborts the run with the kernel's name: the caller's buffer went to
 * the kernel instead of its staged copy.  (A word that is the address of some OTHER mapped memory is only reported
 * with EMU_STRICT_POINTERS=2: an int next to uninitialised padding can spell such an address.) */
#include <stdlib.h>
#include <sys/mman.h>
#include <unistd.h>
static inline int emu_strict_pointers() { static const int on = getenv("EMU_STRICT_POINTERS") ? atoi(getenv("EMU_STRICT_POINTERS")) : 0; return on; }
inline std::map<uintptr_t, size_t> &emu_host_ranges() { static std::map<uintptr_t, size_t> m; return m; }
static inline void emu_note_host(const void *p, size_t n) {
    if(!emu_strict_pointers() || n == 0 || emu_dev_range(p, 1) != 0) return;
    std::lock_guard<std::mutex> g(emu_dev_mutex());
    auto &m = emu_host_ranges();
    if(m.size() > (1u << 16)) m.clear();
    size_t &len = m[(uintptr_t)p];
    if(n > len) len = n;
}
static inline bool emu_in(std::map<uintptr_t, size_t> &m, uintptr_t v) {
    auto it = m.upper_bound(v);
    return it != m.begin() && (--it, v < it->first + it->second);
}
static inline void emu_scan_one(const char *kernel, int index, const void *arg, size_t bytes) {
    const unsigned char *b = (const unsigned char*)arg;
    for(size_t off = 0; off + 8 <= bytes; off += 8) {
        uintptr_t v;
        memcpy(&v, b + off, 8);
        if(v < 0x100000 || v >= 0x0000800000000000ull) continue;
        if(emu_dev_range((const void*)v, 1) != 0) continue;
        bool host;
        {
            std::lock_guard<std::mutex> g(emu_dev_mutex());
            if(emu_in(emu_pinned_allocs(), v)) continue;
            host = emu_in(emu_host_ranges(), v);
        }
        if(host) {
            fprintf(stderr, "emulated HIP runtime: %s: argument %d holds %p at byte %zu -- a host buffer, not its device copy\n",
                    kernel, index, (void*)v, off);
            abort();
        }
        if(emu_strict_pointers() >= 2) {
            unsigned char vec;
            const long page = sysconf(_SC_PAGESIZE);
            if(mincore((void*)(v & ~(uintptr_t)(page - 1)), 1, &vec) == 0)
                fprintf(stderr, "emulated HIP runtime: (note) %s: argument %d, byte %zu: %p is mapped memory outside every device allocation\n",
                        kernel, index, off, (void*)v);
        }
    }
}
template <typename... A> static inline void emu_scan_args(const char *kernel, const A &...args) {
    if(!emu_strict_pointers()) return;
    int index = 0;
    (void)index;
    ((emu_scan_one(kernel, index++, (const void*)&args, sizeof(args))), ...);
}
static inline hipError_t emu_check_copy(void *d, const void *s, size_t n, int kind, const char *what) {
    if(n == 0) return hipSuccess;
    const int rd = emu_dev_range(d, n), rs = emu_dev_range(s, n);
    const char *why = nullptr;
    if(rd < 0 || rs < 0) why = "runs past the end of a device allocation";
    else if(kind == 1 /* HostToDevice */ && rs == 1) why = "host-to-device copy whose source is device memory";
    else if(kind == 2 /* DeviceToHost */ && rd == 1) why = "device-to-host copy whose destination is device memory";
    if(!why) {
        if(kind == 1) emu_note_host(s, n);
        if(kind == 2) emu_note_host(d, n);
        return hipSuccess;
    }
    fprintf(stderr, "emulated HIP runtime: %s(%p, %p, %zu): %s\n", what, d, s, n, why);
    return 1;                                                               /* hipErrorInvalidValue */
}
/* Device allocations live in an address range of their own (0x6000'0000'0000 upwards, never reused, 64 MiB of
 * unmapped space in front of each): a base pointer the library has biased below the start of its buffer -- "row b of
 * the slab is the buffer's first row" -- still points into nothing, not into somebody's heap, and a kernel that runs
 * off the end of an allocation faults at the next page instead of scribbling over the host's memory. */
inline uintptr_t &emu_dev_bump() { static uintptr_t p = 0x600000000000ull; return p; }
static inline hipError_t hipMalloc(void **p, size_t n) {
    const size_t gap = (size_t)64 << 20, len = ((n ? n : 1) + 4095) & ~(size_t)4095;
    uintptr_t at;
    { std::lock_guard<std::mutex> g(emu_dev_mutex()); at = emu_dev_bump() + gap; emu_dev_bump() += len + gap; }
    void *m = mmap((void*)at, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0);
    if(m == MAP_FAILED || (uintptr_t)m != at) { if(m != MAP_FAILED) munmap(m, len); *p = nullptr; return 2; }
#ifdef EMU_MALLOC_FILL
    memset(m, EMU_MALLOC_FILL, len);
#endif
    *p = m;
    { std::lock_guard<std::mutex> g(emu_dev_mutex()); emu_dev_allocs()[(uintptr_t)m] = n ? n : 1; }
    return hipSuccess; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void *p) {
    if(!p) return hipSuccess;
    size_t n = 0;
    { std::lock_guard<std::mutex> g(emu_dev_mutex());
      auto it = emu_dev_allocs().find((uintptr_t)p);
      if(it == emu_dev_allocs().end()) { fprintf(stderr, "emulated HIP runtime: hipFree(%p): not a device allocation\n", p); return 1; }
      n = it->second; emu_dev_allocs().erase(it); }
    munmap(p, (n + 4095) & ~(size_t)4095);
    return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1);
    if(*p) { std::lock_guard<std::mutex> g(emu_dev_mutex()); emu_pinned_allocs()[(uintptr_t)*p] = n ? n : 1; }
    return *p ? hipSuccess : 2; }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f) { return hipHostMalloc((void**)p, n, f); }
static inline hipError_t hipHostFree(void *p) {
    if(p) { std::lock_guard<std::mutex> g(emu_dev_mutex()); emu_pinned_allocs().erase((uintptr_t)p); }
    free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int kind) {
    if(emu_check_copy(d, s, n, kind, "hipMemcpy")) return 1;
    memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int kind, hipStream_t) {
    if(emu_check_copy(d, s, n, kind, "hipMemcpyAsync")) return 1;
    memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t p, int v, size_t count, hipStream_t) { for(size_t i = 0; i < count; i++) ((int*)p)[i] = v; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeDevice; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }   /* (one address space) */
template <typename T> static inline hipError_t hipMemcpyToSymbol(T &sym, const void *src, size_t n) { memcpy((void*)&sym, src, n); return hipSuccess; }
template <typename T> static inline hipError_t hipMemcpyFromSymbol(void *dst, const T &sym, size_t n) { memcpy(dst, (const void*)&sym, n); return hipSuccess; }

// vector types: float2 / float4 come from the library's own host definitions (agent_types.h under NH_HOSTSIM)
#include "agent_types.h"
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
template <typename T, typename U> static inline T atomicSub(T *p, U v) { T o = *p; *p = (T)(o - (T)v); return o; }
template <typename T, typename U> static inline T atomicOr(T *p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <typename T, typename U> static inline T atomicAnd(T *p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <typename T, typename U> static inline T atomicExch(T *p, U v) { T o = *p; *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicCAS(T *p, U cmp, U v) { T o = *p; if(o == (T)cmp) *p = (T)v; return o; }

