// Internal helpers shared by the libtfc translation units (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tfc.h"

int tfc_set_error(int code, const char* fmt, ...);

#define TFC_REQUIRE(cond, code, ...)                 \
    do {                                             \
        if (!(cond)) return tfc_set_error((code), __VA_ARGS__); \
    } while (0)

#define TFC_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess)                                                                  \
            return tfc_set_error(TFC_ELAUNCH, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                                 __FILE__, __LINE__);                                          \
    } while (0)

#define TFC_LAUNCH_CHECK(name)                                                                  \
    do {                                                                                        \
        hipError_t _e = hipGetLastError();                                                      \
        if (_e != hipSuccess)                                                                   \
            return tfc_set_error(TFC_ELAUNCH, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
    } while (0)

// Zero-fill of `bytes` (a multiple of 4) on `stream`, as a KERNEL (tfc_layout.hip).  Not hipMemsetAsync: a memset node of a
// captured graph that is one chain of nodes (a step captured on one stream) was seen to replay with another byte value
// from the second replay on (ROCm 7.2, gfx950; tools/selftest/graph_memset_probe.py: 0x10.., 0x40.. instead of 0) - in
// tfc_csr_build that turns the counts, and with them every edge index, into garbage.
int tfc_zero_async(void* p, size_t bytes, hipStream_t stream);

// opt-in per-kernel timing (tfc_prof.hip): events on the launch stream around a launch
int tfc_prof_begin(const char* name, hipStream_t stream);
void tfc_prof_end(int idx, hipStream_t stream);
struct TfcProfScope {
    int idx;
    hipStream_t stream;
    TfcProfScope(const char* name, hipStream_t s) : idx(tfc_prof_begin(name, s)), stream(s) {}
    ~TfcProfScope() { tfc_prof_end(idx, stream); }
};

// Blocks b and b+8 share an XCD (round-robin dispatch).  Map the hardware block id to a
// virtual id so that each XCD walks one contiguous slab of the virtual range: neighbouring
// virtual blocks (which gather the same rows) then share an L2.  Bijective for any n.
__device__ __forceinline__ unsigned tfc_xcd_remap(unsigned b, unsigned n) {
    const unsigned q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + i;
}

__device__ __forceinline__ int tfc_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Sum over the 64 lanes of a wave; the total is returned in every lane.
__device__ __forceinline__ float tfc_wave_sum(float v) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false);  // row_mirror
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1,3
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2,3
    v += __int_as_float(t);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---------------------------------------------------------------------------------------------
// Sums of several values over the 64 lanes at once ("reduce-scatter" butterfly): every step
// halves the number of live values, so V values cost about 2 V instructions instead of the
// 7 V of V separate tfc_wave_sum calls.  gfx950's v_permlane32_swap / v_permlane16_swap do the
// two cross-row steps on a PAIR of values in one instruction.
//
// tfc_wave_reduce16(v): on return every lane holds, in the returned float, the 64-lane total of
// v[tfc_reduce16_index(lane)] (all four lanes of a quad hold the same total).
// tfc_wave_reduce4(v):  the same for 4 values; the total of v[tfc_reduce4_index(lane)] is held by
// all 16 lanes of a row.
template <int CTRL>
__device__ __forceinline__ float tfc_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float tfc_swap_add32(float a, float b) {  // lanes < 32: sum of a; others: sum of b
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float tfc_swap_add16(float a, float b) {  // even rows: sum of a; odd rows: sum of b
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int tfc_reduce16_index(int lane) {
    return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}
__device__ __forceinline__ float tfc_wave_reduce16(const float (&v)[16]) {
    const int lane = threadIdx.x & 63;
    float s8[8], s4[4], s2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) s8[i] = tfc_swap_add32(v[i], v[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) s4[i] = tfc_swap_add16(s8[i], s8[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // lanes 0-7 of a row keep s4[i], lanes 8-15 keep s4[i + 2]
        const float a = s4[i] + tfc_dpp<0x128>(s4[i]);          // row_ror:8
        const float b = s4[i + 2] + tfc_dpp<0x128>(s4[i + 2]);
        s2[i] = (lane & 8) ? b : a;
    }
    // lanes with bit 2 clear keep s2[0]; partner = lane ^ 7 (row_half_mirror), same bit 3
    const float a = s2[0] + tfc_dpp<0x141>(s2[0]);
    const float b = s2[1] + tfc_dpp<0x141>(s2[1]);
    float r = (lane & 4) ? b : a;
    r += tfc_dpp<0x4E>(r);  // quad_perm [2,3,0,1]
    r += tfc_dpp<0xB1>(r);  // quad_perm [1,0,3,2]
    return r;
}
__device__ __forceinline__ int tfc_reduce4_index(int lane) { return ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1); }
__device__ __forceinline__ float tfc_wave_reduce4(const float (&v)[4]) {
    const float s0 = tfc_swap_add32(v[0], v[2]), s1 = tfc_swap_add32(v[1], v[3]);
    float r = tfc_swap_add16(s0, s1);
    r += tfc_dpp<0x128>(r);  // row_ror:8
    r += tfc_dpp<0x141>(r);  // row_half_mirror
    r += tfc_dpp<0x4E>(r);
    r += tfc_dpp<0xB1>(r);
    return r;
}
