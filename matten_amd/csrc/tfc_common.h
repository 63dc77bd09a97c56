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
