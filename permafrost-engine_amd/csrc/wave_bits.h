// wave_bits.h -- a 64x64 bit tile held by one wave64: lane r owns row r as a 64-bit mask.
// Neighbour access: W/E by 64-bit shifts inside the lane, N/S by DPP wave shifts across lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct u64x { uint32_t lo, hi; };

__device__ __forceinline__ u64x mk(uint64_t v) { return u64x{(uint32_t)v, (uint32_t)(v >> 32)}; }
__device__ __forceinline__ uint64_t to64(u64x v) { return ((uint64_t)v.hi << 32) | v.lo; }
__device__ __forceinline__ u64x operator|(u64x a, u64x b) { return u64x{a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ u64x operator&(u64x a, u64x b) { return u64x{a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ u64x operator^(u64x a, u64x b) { return u64x{a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ u64x operator~(u64x a) { return u64x{~a.lo, ~a.hi}; }
__device__ __forceinline__ bool nz(u64x a) { return (a.lo | a.hi) != 0; }
// a & ~b
__device__ __forceinline__ u64x andn(u64x a, u64x b) { return u64x{a.lo & ~b.lo, a.hi & ~b.hi}; }

// bit c <- bit c-1 : the value of the WEST neighbour (column c-1) aligned on column c
__device__ __forceinline__ u64x from_w(u64x v)
{
    return u64x{v.lo << 1, __builtin_amdgcn_alignbit(v.hi, v.lo, 31)};
}
// bit c <- bit c+1 : the value of the EAST neighbour (column c+1)
__device__ __forceinline__ u64x from_e(u64x v)
{
    return u64x{__builtin_amdgcn_alignbit(v.hi, v.lo, 1), v.hi >> 1};
}
// lane r <- lane r-1 : the NORTH neighbour row (row r-1); lane 0 reads 0.  DPP wave_shr:1.
__device__ __forceinline__ u64x from_n(u64x v)
{
    return u64x{(uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.lo, 0x138, 0xf, 0xf, true),
                (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.hi, 0x138, 0xf, 0xf, true)};
}
// lane r <- lane r+1 : the SOUTH neighbour row (row r+1); lane 63 reads 0.  DPP wave_shl:1.
__device__ __forceinline__ u64x from_s(u64x v)
{
    return u64x{(uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.lo, 0x130, 0xf, 0xf, true),
                (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.hi, 0x130, 0xf, 0xf, true)};
}

// One BFS expansion: (from_w(f) | from_e(f) | from_n(f) | from_s(f)) & open, written so that the compiler folds the
// four DPP moves into the ORs that consume them (v_or_b32_dpp, v_lshl_or_b32) instead of feeding three-input ORs
// (v_or3_b32 is VOP3: no DPP operand on gfx9): 10 instead of 12 VALU instructions per 64x64 level.  The empty asm
// only keeps the two-input ORs from being merged back into v_or3_b32 before the DPP combine runs.
#ifdef NH_HOSTSIM
__device__ __forceinline__ uint32_t nh_keep32(uint32_t x) { return x; }
#else
__device__ __forceinline__ uint32_t nh_keep32(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
#endif
__device__ __forceinline__ u64x bfs_reach(u64x f, u64x open)
{
    const u64x w = from_w(f), e = from_e(f), n = from_n(f), s = from_s(f);
    const uint32_t a0 = nh_keep32(w.lo | n.lo), a1 = nh_keep32(w.hi | n.hi);
    const uint32_t b0 = nh_keep32(e.lo | s.lo), b1 = nh_keep32(e.hi | s.hi);
    return u64x{(a0 | b0) & open.lo, (a1 | b1) & open.hi};
}
