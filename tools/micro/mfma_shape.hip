// Micro-benchmark (MI355X): v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16 at the per-wave tiles of the
// pipelined igemm kernels (64 x 160 = igemm10_kernel, 64 x 128 = igemm8_kernel), on the kernels' own LDS image
// (128-byte rows, the eight 16-byte chunks of row r XOR-swizzled by (r >> 1) & 7), every operand re-read from LDS by
// ds_read_b128 once per K tile (28 / 24 reads per 64-deep K tile per wave in BOTH shapes), 8 waves per workgroup (two
// per SIMD), one workgroup per CU, N(0,1) fp16 operands.  No global traffic inside the loop: what is compared is the
// matrix pipe + LDS reads, i.e. what the main loop of the igemm kernels would gain or lose from the other MFMA shape
// (MI355X_MICROARCH.md, DVFS give-back item 7: the chip may hold a different clock on the two shapes).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_shape tools/micro/mfma_shape.hip && tools/micro/mfma_shape
// Prints, per (tile, shape): wall time, TFLOP/s, cycles per K tile per wave (s_memtime), sustained clock (s_memtime /
// s_memrealtime x 100 MHz, median over workgroups).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS: X rows [0, 256) then W rows [256, 256 + 2 * TNW), 128 B each, two K-tile buffers (the loop alternates them so that
// the reads are not loop-invariant).
template <int TNW, int SHAPE>   // TNW = wave width in columns (160 / 128); SHAPE 0 = 16x16x32, 1 = 32x32x16
__global__ __launch_bounds__(512, 2) void k(const f16* __restrict__ src, float* __restrict__ sink, int iters,
                                            unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 256 + 2 * TNW, BUF = ROWS * 128;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 2 * BUF / 16; i += 512)              // fill both buffers with the random image
        ((f16x8*)smem)[i] = ((const f16x8*)src)[(i + blockIdx.x * 37) % (1 << 16)];
    __syncthreads();
    const int wr = wave >> 1, wc = wave & 1;
    float keep = 0.f;
    unsigned long long t0 = 0, r0 = 0;
    if constexpr (SHAPE == 0) {
        constexpr int TM = 4, TN = TNW / 16;
        const int frow = lane & 15, fq = lane >> 4, swz = frow >> 1;
        const int c0 = (fq ^ swz) * 16, c1 = ((fq + 4) ^ swz) * 16;
        const char* xrd = smem + (wr * 64 + frow) * 128;
        const char* wrd = smem + (256 + wc * TNW + frow) * 128;
        f32x4 acc[TN][TM];
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            const int bo = (it & 1) * BUF;
            f16x8 X[TM][2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < TM; ++i) X[i][h] = *(const f16x8*)(xrd + bo + i * 2048 + (h ? c1 : c0));
#pragma unroll
            for (int j = 0; j < TN / 2; ++j) {               // one phase of the igemm loops: 2 weight fragments, 16 MFMAs
                f16x8 W[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i) W[i][h] = *(const f16x8*)(wrd + bo + (2 * j + i) * 2048 + (h ? c1 : c0));
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[2 * j + n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[n][h], X[m][h], acc[2 * j + n][m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m) keep += acc[n][m][0] + acc[n][m][3];
    } else {
        constexpr int TM = 2, TN = TNW / 32;                  // 32 x 32 accumulator blocks: 2 x 5 (or 2 x 4)
        const int r32 = lane & 31, kh = lane >> 5, swz = (r32 >> 1) & 7;
        const char* xrd = smem + (wr * 64 + r32) * 128;
        const char* wrd = smem + (256 + wc * TNW + r32) * 128;
        f32x16 acc[TN][TM];
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[n][m][e] = 0.f;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            const int bo = (it & 1) * BUF;
            f16x8 X[TM][4];                                  // [fragment][k step of 16]: chunk 2 s + kh
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i) X[i][s] = *(const f16x8*)(xrd + bo + i * 4096 + (((2 * s + kh) ^ swz) * 16));
#pragma unroll
            for (int j = 0; j < TN; ++j) {                    // one weight fragment (32 columns), 8 MFMAs
                f16x8 W[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) W[s] = *(const f16x8*)(wrd + bo + j * 4096 + (((2 * s + kh) ^ swz) * 16));
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < TM; ++m)
                        acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[s], X[m][s], acc[j][m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m) keep += acc[n][m][0] + acc[n][m][15];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }
    sink[blockIdx.x * 512 + t] = keep;
}

template <int TNW, int SHAPE>
void run(const f16* src, float* sink, unsigned long long* stamps, int nwg, int iters, const char* what) {
    const size_t smem = (size_t)2 * (256 + 2 * TNW) * 128;
    hipFuncSetAttribute((const void*)k<TNW, SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> tf, cyc, clk;
    for (int rep = 0; rep < 7; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<TNW, SHAPE>), dim3(nwg), dim3(512), smem, 0, src, sink, iters, stamps);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nwg * 16);
        hipMemcpy(h.data(), stamps, nwg * 16 * 8, hipMemcpyDeviceToHost);
        std::vector<double> c, f;
        for (int i = 0; i < nwg * 8; ++i) { c.push_back((double)h[2 * i]); f.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }
        std::sort(c.begin(), c.end()); std::sort(f.begin(), f.end());
        if (rep < 2) continue;                               // warm-up: the clock settles under load
        tf.push_back(2.0 * 256 * (2 * TNW) * 64 * (double)iters * nwg / ms / 1e9);
        cyc.push_back(c[c.size() / 2] / iters); clk.push_back(f[f.size() / 2]);
    }
    std::sort(tf.begin(), tf.end()); std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    printf("%-34s %4d workgroups: median %7.1f TFLOP/s (min %7.1f max %7.1f)  %7.1f cycles / K tile / wave  clock %.3f GHz\n", what, nwg,
           tf[tf.size() / 2], tf.front(), tf.back(), cyc[cyc.size() / 2], clk[clk.size() / 2]);
}

int main() {
    f16* src; float* sink; unsigned long long* stamps;
    const int NSRC = (1 << 16) * 8 + 64;
    std::vector<f16> h(NSRC);
    srand(1);
    for (int i = 0; i < NSRC; ++i) {                          // N(0, 1) by Box-Muller
        const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        h[i] = (f16)(float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
    }
    hipMalloc(&src, NSRC * 2); hipMalloc(&sink, 512 * 512 * 4); hipMalloc(&stamps, 512 * 16 * 8);
    hipMemcpy(src, h.data(), NSRC * 2, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int round = 0; round < 2; ++round)                   // interleaved rounds in one process (rule 24)
        for (int nwg : {256}) {
            run<160, 0>(src, sink, stamps, nwg, iters, "64x160 per wave, 16x16x32_f16");
            run<160, 1>(src, sink, stamps, nwg, iters, "64x160 per wave, 32x32x16_f16");
            run<128, 0>(src, sink, stamps, nwg, iters, "64x128 per wave, 16x16x32_f16");
            run<128, 1>(src, sink, stamps, nwg, iters, "64x128 per wave, 32x32x16_f16");
        }
    // the same on all-zero operands (ranks the shapes by cycles only: no DVFS difference)
    hipMemset(src, 0, NSRC * 2);
    run<160, 0>(src, sink, stamps, 256, iters, "64x160, 16x16x32_f16, ZERO operands");
    run<160, 1>(src, sink, stamps, 256, iters, "64x160, 32x32x16_f16, ZERO operands");
    return 0;
}
