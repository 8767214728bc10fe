// Shared helpers for the gfx950 kernels of libposetraj_hip.so.  CDNA4 only: wave64, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/posetraj_hip.h"

typedef _Float16 f16;
typedef f16   f16x2 __attribute__((ext_vector_type(2)));
typedef f16   f16x4 __attribute__((ext_vector_type(4)));
typedef f16   f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define PT_WAVE 64

void pt_set_error(const char* fmt, ...);
const void* pt_zero_page();

#define PT_CHECK(cond, ...)                 \
    do {                                    \
        if (!(cond)) {                      \
            pt_set_error(__VA_ARGS__);      \
            return 1;                       \
        }                                   \
    } while (0)

#define PT_LAUNCH_CHECK(name)                                                       \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            pt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return 2;                                                               \
        }                                                                           \
    } while (0)

// ---- profiling hooks (api.hip)
void pt_prof_begin(int family, hipStream_t s, double flops);
void pt_prof_end(int family, hipStream_t s);

// ---- device helpers
__device__ __forceinline__ float pt_silu(float x) { return x / (1.0f + __expf(-x)); }

// erf-GELU with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 output's 5e-4):
//   erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z >= 0;  odd extension.
// ~16 VALU instructions (2 transcendental) instead of libm erff's ~40 with a branch - the GEGLU epilogue of the
// K = 320 feed-forward GEMMs was spending 2.5x the main loop's time in erff.
__device__ __forceinline__ float pt_gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = 1.0f - poly * t * e;
    const float erf_x = copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erf_x);
}

// async 16-byte global -> LDS copy (LDS-DMA).  `lds_wave_base` must be wave-uniform; lane i lands at base + 16*i.
__device__ __forceinline__ void pt_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 4-byte variant: lane i lands at base + 4*i.  Used as a cache-line touch (L2 prefetch with no register destination).
__device__ __forceinline__ void pt_glds4(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

__device__ __forceinline__ float pt_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// XCD-aware bijective remap of a linear workgroup id: workgroups that share an XCD (id % 8 equal under the
// round-robin dispatch) get a contiguous chunk of the tile space, so neighbouring tiles hit the same L2.
__device__ __forceinline__ int pt_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
