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
const void* pt_zero_page();      // the current device's zero page (nullptr until pt_set_zero_page ran on it)
int pt_device();                 // current HIP device, clamped to the per-device tables' range

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

// ---- profiling hooks (api.hip); families: 0 igemm, 1 spatial attention, 2 pt_gemm_f16 (training)
constexpr int PT_PROF_IGEMM = 0, PT_PROF_ATTN = 1, PT_PROF_GEMM = 2, PT_PROF_FAMILIES = 3;
void pt_prof_begin(int family, hipStream_t s, double flops);
void pt_prof_end(int family, hipStream_t s);

// ---- device helpers
// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE fp32 division would cost ~10 more instructions per value,
// which made the GroupNorm apply pass VALU-bound; outputs are rounded to fp16 (11 bits) anyway
__device__ __forceinline__ float pt_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// erf-GELU.  With z = |x| / sqrt(2):  gelu(x) = x * Phi(x) = relu(x) - (|x| / 2) * erfc(z), and
//   erfc(z) ~= 2^(-z (c1 + c2 z + c3 z^2 + c4 z^3 + c5 z^4))
// (weighted minimax fit on [0, 4.2], tools/fit_gelu.py; the exponent keeps growing beyond, so erfc -> 0 without a
// clamp): |erf error| <= 6.3e-7, |gelu error| <= 1.1e-6 evaluated in fp32 - far below the fp16 output's 5e-4.
// 9 plain VALU + 1 transcendental, no reciprocal and no sign fix-up (Abramowitz-Stegun 7.1.26 needed 13 + 2, libm
// erff ~40 with a branch): the GEGLU epilogue of the K = 320 feed-forward GEMMs costs as much as their main loop.
// pt_gelu_erf2 does two values with the polynomial in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32).
// relu(x) is written (x + |x|) / 2 - exact in fp32, so the same value as fmaxf(x, 0) - because fmaxf() on a value that comes out of
// an MFMA costs TWO instructions (hipcc first quiets a possible signalling NaN: `v_max_f32 x, x, x`; an fmed3 builtin is folded
// back to the same pair) and the kernels that run this are bound by instruction count (energy): 4 -> 2 instructions per
// pair of values in the packed form.
__device__ __forceinline__ float pt_gelu_erf(float x) {
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752f;
    float q = 0.00294415708f;
    q = q * z - 0.0295900398f;
    q = q * z + 0.148665627f;
    q = q * z + 0.918509366f;
    q = q * z + 1.62788901f;
    const float e = __builtin_amdgcn_exp2f(-(q * z));
    return (x + ax) * 0.5f - 0.5f * ax * e;
}
__device__ __forceinline__ f32x2 pt_gelu_erf2(f32x2 x) {
    f32x2 ax; ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
    const f32x2 z = ax * 0.70710678118654752f;
    f32x2 q = z * 0.00294415708f - 0.0295900398f;
    q = q * z + 0.148665627f;
    q = q * z + 0.918509366f;
    q = q * z + 1.62788901f;
    const f32x2 pz = q * z;
    f32x2 e; e[0] = __builtin_amdgcn_exp2f(-pz[0]); e[1] = __builtin_amdgcn_exp2f(-pz[1]);
    const f32x2 r = (x + ax) * 0.5f;
    return r - (ax * 0.5f) * e;
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
