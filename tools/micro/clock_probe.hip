// In-kernel shader clock beside any other kernel (MI355X_MICROARCH.md, DVFS give-back item 6): one 64-lane workgroup samples
// s_memtime (shader cycles) and s_memrealtime (100 MHz) once per `period` real-time ticks, `n` times, into out[2 i], out[2 i + 1].
// Launched on a stream of its own BEFORE the kernels under test, it stays resident on one SIMD slot for the whole measurement;
// clock_i = (memtime_i - memtime_{i-1}) / (realtime_i - realtime_{i-1}) x 100 MHz.  Its values reach no other kernel.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/clock_probe.hip -o tools/micro/libclock_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int n, unsigned long long period,
                                                         const volatile int* stop) {
    if (threadIdx.x != 0) return;
    unsigned long long next = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        unsigned long long rt;
        while ((rt = __builtin_amdgcn_s_memrealtime()) < next) __builtin_amdgcn_s_sleep(64);
        const unsigned long long ct = __builtin_amdgcn_s_memtime();
        out[2 * i] = ct;
        out[2 * i + 1] = rt;
        next = rt + period;
        if (stop && *stop) {                                  // host-set flag (pinned memory): end early, mark the tail unused
            for (int j = i + 1; j < n; ++j) out[2 * j + 1] = 0;
            break;
        }
    }
}

extern "C" int clock_probe_launch(void* out, int n, unsigned long long period_ticks, const void* stop, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, n, period_ticks,
                       (const volatile int*)stop);
    return (int)hipGetLastError();
}
