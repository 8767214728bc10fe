// Micro-benchmark (MI355X): issue cost of the VALU instructions of the attention softmax, in cycles per wave instruction, with one
// and with two waves per SIMD, and beside a v_mfma_f32_32x32x16_f16 stream issued by the OTHER wave of the SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rate tools/micro/valu_rate.hip && tools/micro/valu_rate
// Each kernel runs REP x 64 independent instructions of one kind per iteration (no dependency chains shorter than 16) between two
// s_memtime stamps (core clock cycles; s_memrealtime is the 100-MHz reference, their ratio is the clock the wave ran at).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define R16(OP)  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

// KIND: 0 v_exp_f32, 1 v_add_f32, 2 v_max3_f32, 3 v_cvt_pk_f16_f32, 4 v_pk_add_f32, 5 v_pk_fma_f32, 6 v_lshl_add_u64, 7 v_mov_b32,
//       8 v_exp_f16, 9 v_rcp_f32, 10 v_fma_f32, 11 v_permlane32_swap
// MIX: waves with (wave & 4) != 0 (the second wave of each SIMD at 8 waves per workgroup) issue MFMAs instead
template <int KIND, bool MIX>
__global__ __launch_bounds__(512, 1) void k(float* __restrict__ sink, int iters, unsigned long long* __restrict__ stamps, float seed) {
    const int t = threadIdx.x, wave = t >> 6;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1) + t * 1e-3f;
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = (double)seed * (i + 1) + t;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (f16)(seed + i); b[i] = (f16)(seed - i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MIX && (wave & 4)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                if constexpr (KIND == 0) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 1) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 2) {
#define OP(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(seed), "v"(v[(i + 1) & 15]));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 3) {
#define OP(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 4) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 5) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 6) {
#define OP(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 7) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 8) {
#define OP(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 9) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                    R16(OP)
#undef OP
                } else if constexpr (KIND == 10) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(seed));
                    R16(OP)
#undef OP
                } else {
#define OP(i) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 8) & 15]));
                    R16(OP)
#undef OP
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) keep += v[i] + acc[i] + (float)d[i & 7];
    if (keep == 12345.678f) sink[t] = keep;
    if ((t & 63) == 0) {
        stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }
}

template <int KIND, bool MIX>
static void run(const char* name, int nthreads, int iters) {
    float* sink; unsigned long long* st;
    const int blocks = 256;
    hipMalloc(&sink, 4096); hipMalloc(&st, blocks * 8 * 2 * 8);
    hipMemset(st, 0, blocks * 8 * 2 * 8);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<KIND, MIX>), dim3(blocks), dim3(nthreads), 0, 0, sink, iters, st, 0.37f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8 * 2);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = nthreads / 64;
    std::vector<double> cyc, clk, mf;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < nw; ++w) {
            if (MIX && (w & 4)) { mf.push_back((double)h[(b * 8 + w) * 2] / (iters * 8.0)); continue; }
            const double tk = (double)h[(b * 8 + w) * 2], rt = (double)h[(b * 8 + w) * 2 + 1];
            const double ghz = tk / rt * 0.1;                // s_memtime ticks per 100-MHz tick -> GHz if s_memtime counts core clocks
            cyc.push_back(tk / (iters * 64.0)); clk.push_back(ghz);
        }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    printf("%-22s %d wave(s)/SIMD%s: %6.2f s_memtime ticks per instruction (median; min %.2f max %.2f); s_memtime / s_memrealtime = %.3f\n", name,
           nthreads / 256, MIX ? " (one VALU + one MFMA wave)" : "", cyc[cyc.size() / 2], cyc.front(), cyc.back(), clk[clk.size() / 2] * 10);
    if (MIX) {
        std::sort(mf.begin(), mf.end());
        printf("%-22s     the MFMA wave beside it: %6.2f cycles per v_mfma_f32_32x32x16_f16 (32 alone)\n", "", mf[mf.size() / 2]);
    }
    hipFree(sink); hipFree(st);
}


// One wave's own stream: NV independent VALU instructions (v_exp_f32 or v_max3_f32) behind every v_mfma_f32_32x32x16_f16 - does the
// wave's VALU work run under its own MFMA (32 cycles per group) or behind it (32 + NV x rate)?
template <int NV, int KIND>
__global__ __launch_bounds__(512, 1) void kmix(float* __restrict__ sink, int iters, unsigned long long* __restrict__ stamps, float seed) {
    const int t = threadIdx.x, wave = t >> 6;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1) + t * 1e-3f;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (f16)(seed + i); b[i] = (f16)(seed - i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j & 1]) : "v"(a), "v"(b));
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if constexpr (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(j * NV + n) & 15]));
                else asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[(j * NV + n) & 15]) : "v"(seed));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) keep += v[i] + acc[0][i] + acc[1][i];
    if (keep == 12345.678f) sink[t] = keep;
    if ((t & 63) == 0) stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
}
template <int NV, int KIND>
static void runmix(const char* name, int nthreads, int iters) {
    float* sink; unsigned long long* st;
    const int blocks = 256;
    hipMalloc(&sink, 4096); hipMalloc(&st, blocks * 8 * 2 * 8);
    hipMemset(st, 0, blocks * 8 * 2 * 8);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kmix<NV, KIND>), dim3(blocks), dim3(nthreads), 0, 0, sink, iters, st, 0.37f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8 * 2);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < nthreads / 64; ++w) cyc.push_back((double)h[(b * 8 + w) * 2] / (iters * 8.0));
    std::sort(cyc.begin(), cyc.end());
    printf("one stream: MFMA + %d x %-10s %d wave(s)/SIMD: %6.2f cycles per group per wave (median; min %.2f max %.2f)\n", NV, name, nthreads / 256,
           cyc[cyc.size() / 2], cyc.front(), cyc.back());
    hipFree(sink); hipFree(st);
}

#define BOTH(K, NAME) run<K, false>(NAME, 256, 2000); run<K, false>(NAME, 512, 2000); run<K, true>(NAME, 512, 2000);
int main() {
    runmix<0, 0>("v_exp_f32", 256, 2000); runmix<2, 0>("v_exp_f32", 256, 2000); runmix<2, 0>("v_exp_f32", 512, 2000); runmix<3, 0>("v_exp_f32", 256, 2000); runmix<3, 0>("v_exp_f32", 512, 2000);
    runmix<4, 1>("v_max3_f32", 256, 2000); runmix<4, 1>("v_max3_f32", 512, 2000); runmix<6, 1>("v_max3_f32", 256, 2000); runmix<6, 1>("v_max3_f32", 512, 2000);
    BOTH(0, "v_exp_f32") BOTH(8, "v_exp_f16") BOTH(9, "v_rcp_f32") BOTH(1, "v_add_f32") BOTH(10, "v_fma_f32") BOTH(2, "v_max3_f32")
    BOTH(3, "v_cvt_pk_f16_f32") BOTH(4, "v_pk_add_f32") BOTH(5, "v_pk_fma_f32") BOTH(6, "v_lshl_add_u64") BOTH(7, "v_mov_b32")
    BOTH(11, "s_nop 1 + permlane32_swap")
    return 0;
}
