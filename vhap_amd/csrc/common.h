// Shared helpers for the gfx950 kernels (internal, not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vhap_hip.h"

// hipGetLastError() is sticky and process-wide: an earlier hipEventQuery()/hipStreamQuery() of the host
// framework legitimately leaves hipErrorNotReady behind.  Every entry point therefore clears the slot on
// entry (VHAP_ENTER) and only then attributes a non-success value to its own launches.
#define VHAP_ENTER() (void)hipGetLastError()
#define VHAP_LAUNCH_CHECK()                                                  \
    do {                                                                     \
        const hipError_t vhap_e_ = hipGetLastError();                        \
        if (vhap_e_ != hipSuccess && vhap_e_ != hipErrorNotReady) return VHAP_E_HIP; \
    } while (0)

// Profiling-only A/B switches: thread-local (a profiling script sets them on its own thread; nothing in the product path does),
// so entry points stay re-entrant.  misc.hip
extern thread_local int vhap_g_debug_flags;

// zero a small accumulator unless the caller declared -- in the `call_flags` argument of THIS call -- that it hands in pre-zeroed
// accumulators (step executors keep all of them in one arena cleared by a single launch)
#define VHAP_ZERO_ACC(ptr, bytes, st)                                   \
    do {                                                                \
        if (!(call_flags & VHAP_CALL_ACC_PREZEROED)) {                  \
            vhap_zero_async((ptr), (bytes), (st));                      \
            VHAP_LAUNCH_CHECK();                                        \
        }                                                               \
    } while (0)

static inline hipStream_t vhap_stream(vhap_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int vhap_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// XCD-aware logical block id: the dispatcher places hardware block b on XCD b % 8 (observed, used
// for L2 affinity only).  Give every XCD a CONTIGUOUS slice of the logical grid so that the
// blocks sharing one frame's geometry hit the same L2.
__device__ __forceinline__ unsigned vhap_xcd_remap(unsigned bid, unsigned nblocks) {
    constexpr unsigned NXCD = 8;
    if (nblocks % NXCD) return bid;
    return (bid % NXCD) * (nblocks / NXCD) + bid / NXCD;
}

__device__ __forceinline__ float vhap_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave-wide sum on the VALU: two quad permutes + two row mirrors (DPP modifiers, full rate) sum each row of 16 lanes, four
// v_readlane + three adds combine the rows; the result is wave-uniform.  __shfl_xor (ds_bpermute) goes through the LDS crossbar --
// one per CU, ~10x the cost when a kernel reduces dozens of values per wave (deferred.hip: 27 per wave).
template <int CTRL>
__device__ __forceinline__ float vhap_dpp(float v) {     // (every lane has a valid source for these controls: `old` = v costs no v_mov)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// N values at once, step by step: the N adds of a step are independent, so the DPP read-after-write hazard slots are filled with
// useful work instead of s_nop.  Afterwards EVERY lane holds the sum over its row of 16 lanes.
template <int N>
__device__ __forceinline__ void vhap_row_sums_dpp(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += vhap_dpp<0xB1>(v[i]);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += vhap_dpp<0x4E>(v[i]);      // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += vhap_dpp<0x141>(v[i]);     // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += vhap_dpp<0x140>(v[i]);     // row_mirror
}
__device__ __forceinline__ float vhap_wave_sum_dpp(float v) {
    v += vhap_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += vhap_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += vhap_dpp<0x141>(v);     // row_half_mirror: quads 0 <-> 1, 2 <-> 3
    v += vhap_dpp<0x140>(v);     // row_mirror: halves of the row
    const int i = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}

// the same reductions on 32-bit unsigned values (maximum / sum over the wave, result wave-uniform)
template <int CTRL>
__device__ __forceinline__ unsigned vhap_dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned vhap_wave_max_u32_dpp(unsigned v) {
    v = max(v, vhap_dpp_u<0xB1>(v));
    v = max(v, vhap_dpp_u<0x4E>(v));
    v = max(v, vhap_dpp_u<0x141>(v));
    v = max(v, vhap_dpp_u<0x140>(v));
    const int i = (int)v;
    return max(max((unsigned)__builtin_amdgcn_readlane(i, 0), (unsigned)__builtin_amdgcn_readlane(i, 16)),
               max((unsigned)__builtin_amdgcn_readlane(i, 32), (unsigned)__builtin_amdgcn_readlane(i, 48)));
}
__device__ __forceinline__ unsigned vhap_wave_sum_u32_dpp(unsigned v) {
    v += vhap_dpp_u<0xB1>(v);
    v += vhap_dpp_u<0x4E>(v);
    v += vhap_dpp_u<0x141>(v);
    v += vhap_dpp_u<0x140>(v);
    const int i = (int)v;
    return ((unsigned)__builtin_amdgcn_readlane(i, 0) + (unsigned)__builtin_amdgcn_readlane(i, 16)) +
           ((unsigned)__builtin_amdgcn_readlane(i, 32) + (unsigned)__builtin_amdgcn_readlane(i, 48));
}

// Zero-fill / copy as ordinary kernel launches.  hipMemsetAsync / hipMemcpyAsync become memset / memcpy NODES under
// stream capture, and on ROCm 7.2 those nodes were observed to run out of order with the neighbouring kernel nodes when
// the graph is replayed on the null stream (stale accumulators) -- kernel nodes keep their order.
// `bytes` must be a multiple of 4 and `p` 4-byte aligned.
static __global__ __launch_bounds__(256) void vhap_zero_words_kernel(uint32_t* __restrict__ p, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static inline void vhap_zero_async(void* p, size_t bytes, hipStream_t st) {
    const size_t nwords = bytes / 4;
    if (nwords == 0) return;
    const int blocks = (int)((nwords + 255) / 256 < 2048 ? (nwords + 255) / 256 : 2048);
    vhap_zero_words_kernel<<<blocks, 256, 0, st>>>(reinterpret_cast<uint32_t*>(p), nwords);
}
static __global__ __launch_bounds__(256) void vhap_copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static inline void vhap_copy_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
    const size_t nwords = bytes / 4;
    if (nwords == 0) return;
    const int blocks = (int)((nwords + 255) / 256 < 4096 ? (nwords + 255) / 256 : 4096);
    vhap_copy_words_kernel<<<blocks, 256, 0, st>>>(reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint32_t*>(dst), nwords);
}
