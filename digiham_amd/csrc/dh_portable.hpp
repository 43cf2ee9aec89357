// dh_portable.hpp -- how the kernel bodies are written.
//
// Every kernel in this library is "one channel (or one small group of codewords) per
// 64-lane wavefront", one wavefront per workgroup.  The bodies are written as a sequence of
// PHASES: inside a phase each lane works independently (DH_FOR_LANES), between phases the
// lanes exchange data through the workgroup's LDS block and meet at DH_BARRIER().  Wave-wide
// votes use DH_BALLOT_ACC.  Loop control between phases only uses wave-uniform values.
//
// When hipcc compiles this for gfx950, DH_FOR_LANES binds `lane` to threadIdx.x, DH_BARRIER is
// s_barrier (free for a single-wave workgroup apart from the LDS wait) and DH_BALLOT_ACC is
// v_cmp + s_mov of the EXEC-wide vote.  When a plain C++ compiler builds the *test harness*
// (tests/host_harness), the same phases run as `for (lane = 0..63)` loops, which lets the
// CPU-only test tier execute the exact wave algorithm (index maths, carries, block-parallel
// timing recovery) against the oracle without a GPU.  The shipped library contains only the
// gfx950 build; there is no CPU fallback in the product.
#pragma once

#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define DH_DEVICE_BUILD 1
#define DH_HD __host__ __device__ __forceinline__
#define DH_D __device__ __forceinline__
#else
#define DH_DEVICE_BUILD 0
#define DH_HD inline
#define DH_D inline
#endif

#define DH_WAVE 64

#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
// DH_FOR_LANES_FRESH: the lane id passes through an empty volatile asm, so nothing derived from it (per-lane LDS
// addresses, lane % 10, 64-bit row pointers) is loop-invariant to the compiler.  Left alone it hoists dozens of such
// values out of the slicer's per-run loop, across the FIR where every register is taken, spills them to scratch and
// reloads them with dependent memory round trips inside the latency-bound phases; recomputing them costs a few VALU
// instructions per phase instead.  Used for the phases inside that loop; everything else uses DH_FOR_LANES.
static __device__ __forceinline__ int dh_fresh_lane_id_() { int l = (int) threadIdx.x; asm volatile("" : "+v"(l)); return l; }
#define DH_FOR_LANES(lane) for (int lane = (int) threadIdx.x, dh_once_ = 1; dh_once_; dh_once_ = 0)
#define DH_FOR_LANES_FRESH(lane) for (int lane = dh_fresh_lane_id_(), dh_once_ = 1; dh_once_; dh_once_ = 0)
// The phases of a body exchange data through LDS only, so the barrier orders LDS traffic only: __syncthreads() also fences
// GLOBAL memory, which makes the compiler wait (s_waitcnt vmcnt(0)) for every global load in flight -- the next window's
// loads, requested a phase earlier precisely so that they can stay in flight through the phases that follow (round 3: the wait
// sat in front of the slicing phase and cost 1 ms of a 6.8 ms step).  Where a body hands data to another one through global
// memory the caller fences explicitly (k_chain: __syncthreads() between slicer and decoder, workgroup scope).
#define DH_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#define DH_BALLOT_ACC(mask, pred, lane) (mask) = __builtin_amdgcn_ballot_w64((bool) (pred))      /* (v_cmp straight into the scalar pair; __ballot((pred) ? 1 : 0) left a v_cndmask + v_cmp behind every vote) */
#define DH_IS_LANE0(lane) ((lane) == 0)
#else
#define DH_FOR_LANES(lane) for (int lane = 0; lane < DH_WAVE; ++lane)
#define DH_FOR_LANES_FRESH(lane) DH_FOR_LANES(lane)
#define DH_BARRIER() ((void) 0)
#define DH_BALLOT_ACC(mask, pred, lane) (mask) |= ((uint64_t) ((pred) ? 1 : 0) << (lane))
#define DH_IS_LANE0(lane) ((lane) == 0)
#endif

// A value every lane holds identically (loaded from LDS / memory at a wave-uniform address): moving it to a
// scalar register lets the arithmetic that follows run on the scalar ALU instead of costing VALU issue cycles.
DH_HD uint32_t dh_uniform(uint32_t x) {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t) __builtin_amdgcn_readfirstlane((int) x);
#else
    return x;
#endif
}

DH_HD int dh_popc32(uint32_t x) {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

DH_HD int dh_popc64(uint64_t x) {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

DH_HD uint32_t dh_brev32(uint32_t x) {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

DH_HD int dh_ffs64(uint64_t x) {   // index of lowest set bit, x != 0
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long) x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}

template <typename T> DH_HD T dh_min(T a, T b) { return a < b ? a : b; }
template <typename T> DH_HD T dh_max(T a, T b) { return a > b ? a : b; }

// branch weights: what the register allocator keeps out of the hot path (spill code goes where the weights are low)
#define DH_LIKELY(x) __builtin_expect(!!(x), 1)
#define DH_UNLIKELY(x) __builtin_expect(!!(x), 0)
