// Micro-benchmark (MI355X): store throughput per CU as a function of how many CUs store at once, and of the segment
// shape.  One 512-thread workgroup per CU slot; every lane issues 16-byte stores.
//   mode 0: fully coalesced (a wave covers 1 KiB contiguous)
//   mode 1: rows of 320 B, 3 rows per wave-instruction, row pitch 1920 B  (the igemm tail at N = 960)
//   mode 2: rows of 160 B, 6 rows per instruction, row pitch 2560 B       (the GEGLU tail)
//   mode 3: 16 rows x 64 B per instruction, pitch 1920 B                  (direct stores from the accumulator layout, paired)
//   mode 4: 16 rows x 32 B per instruction (8-byte stores), pitch 1920 B  (direct stores from the accumulator layout)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k(char* out, size_t per_wg, int mode, int iters, unsigned long long* cyc) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    char* base = out + (size_t)blockIdx.x * per_wg;
    const f4 v = {1.f, 2.f, 3.f, (float)t};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        size_t off;
        if (mode == 0) off = ((size_t)i * 8 + wave) * 1024 + lane * 16;
        else if (mode == 1) { const int r = lane / 20, c = lane % 20; off = ((size_t)(i * 8 + wave) * 3 + r) * 1920 + c * 16; if (lane >= 60) off = ((size_t)(i * 8 + wave) * 3) * 1920 + 320; }
        else if (mode == 2) { const int r = lane / 10, c = lane % 10; off = ((size_t)(i * 8 + wave) * 6 + r) * 2560 + c * 16; if (lane >= 60) off = ((size_t)(i * 8 + wave) * 6) * 2560 + 160; }
        else if (mode == 3) { const int r = lane & 15, c = lane >> 4; off = ((size_t)(i * 8 + wave) * 16 + r) * 1920 + c * 16; }
        else { const int r = lane & 15, c = lane >> 4; off = ((size_t)(i * 8 + wave) * 16 + r) * 1920 + c * 8; }
        off %= per_wg - 64;
        off &= ~(size_t)15;
        if (mode == 4) *(f2*)(base + off) = (f2){v[0], v[1]};
        else *(f4*)(base + off) = v;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
    const size_t per_wg = 8u << 20;
    char* out; unsigned long long* cyc;
    hipMalloc(&out, per_wg * 256); hipMalloc(&cyc, 256 * 8);
    unsigned long long h[256];
    const int iters = 2048;
    for (int mode = 0; mode < 5; ++mode)
        for (int nwg : {8, 32, 64, 128, 256}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 0, 0, out, per_wg, mode, 64, cyc);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 0, 0, out, per_wg, mode, iters, cyc);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, cyc, nwg * 8, hipMemcpyDeviceToHost);
            double c = 0; for (int i = 0; i < nwg; ++i) c += h[i]; c /= nwg;
            const double bytes = (double)iters * 512 * (mode == 4 ? 8 : ((mode == 2 || mode == 1) ? 16.0 * 60 / 64 : 16));
            printf("mode %d  %3d workgroups: %8.1f us  %7.2f TB/s chip  %6.1f B/clk/CU (s_memtime ticks)  %6.1f GB/s per CU\n", mode, nwg,
                   ms * 1e3, bytes * nwg / ms / 1e9, bytes / c, bytes / ms / 1e6);
        }
    return 0;
}
