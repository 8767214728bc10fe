// VAE-side element-wise kernels (HBM-bound): the tail of AutoencoderKLTemporalDecoder.decode (time_conv_out + the layout of
// decode_latents), tensor2vid's post-processing, the posterior of the encoder.
#include "pt_common.h"

namespace {

// time_conv_out: Conv3d(3 -> 3, kernel (3,1,1), padding (1,0,0)) over the F frames of ONE decode call, fused with the
// channels-last -> [frame][channel][pixel] transposition decode_latents needs.
//   x   : fp32 channels-last [F, HW, ldx] (conv_out's output; the first 3 columns are used)
//   out : fp32, frame f of the call lands at out + ((i0 + f) * 3 + co) * HW  (the caller's [B*F_total, 3, H, W] buffer)
// One thread per (frame, pixel); weights live in constant registers (27 + 3 floats).
struct TconvW { float w[3][3][3]; float b[3]; };      // [co][ci][kt]

__global__ __launch_bounds__(256) void vae_time_conv_out_kernel(const float* __restrict__ x, int ldx, TconvW W, int F,
                                                                int64_t HW, float* __restrict__ out) {
    const int f = blockIdx.y;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (int64_t)gridDim.x * 256) {
        float acc[3] = {W.b[0], W.b[1], W.b[2]};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int ff = f + kt - 1;
            if (ff < 0 || ff >= F) continue;
            const float* px = x + ((int64_t)ff * HW + p) * ldx;
            const float v0 = px[0], v1 = px[1], v2 = px[2];
#pragma unroll
            for (int co = 0; co < 3; ++co) acc[co] += W.w[co][0][kt] * v0 + W.w[co][1][kt] * v1 + W.w[co][2][kt] * v2;
        }
#pragma unroll
        for (int co = 0; co < 3; ++co) out[((int64_t)f * 3 + co) * HW + p] = acc[co];
    }
}

// tensor2vid + VaeImageProcessor.postprocess for one clip: src fp32 [F, 3, HW] in [-1, 1] (any range: clamped)
//   mode 0 ("pt") : dst fp32 [F, 3, HW]   = clamp(x / 2 + 0.5, 0, 1)
//   mode 1 ("np") : dst fp32 [F, HW, 3]   = the same, channels last
//   mode 2 ("pil"): dst uint8 [F, HW, 3]  = round-half-even(255 * that)     (numpy: (x * 255).round().astype(uint8))
__global__ __launch_bounds__(256) void frames_postprocess_kernel(const float* __restrict__ src, int64_t HW, int mode,
                                                                 void* __restrict__ dst) {
    const int f = blockIdx.y;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float u = src[((int64_t)f * 3 + c) * HW + p] * 0.5f + 0.5f;
            u = fminf(fmaxf(u, 0.f), 1.f);
            if (mode == 0) ((float*)dst)[((int64_t)f * 3 + c) * HW + p] = u;
            else if (mode == 1) ((float*)dst)[((int64_t)f * HW + p) * 3 + c] = u;
            else ((uint8_t*)dst)[((int64_t)f * HW + p) * 3 + c] = (uint8_t)__builtin_rintf(u * 255.0f);
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const float* __restrict__ src, int C, int64_t HW, int ld,
                                                               float* __restrict__ dst) {
    const int n = blockIdx.y;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (int64_t)gridDim.x * 256)
        for (int c = 0; c < C; ++c) dst[((int64_t)n * C + c) * HW + p] = src[((int64_t)n * HW + p) * ld + c];
}

// DiagonalGaussianDistribution.sample: mean + exp(0.5 * clamp(logvar, -30, 20)) * noise; params [N, 2C, HW] (mean | logvar)
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* __restrict__ params, const float* __restrict__ noise,
                                                              int C, int64_t HW, float* __restrict__ out, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / (C * HW), r = i - n * (C * HW);
        const float mean = params[n * 2 * C * HW + r];
        float lv = params[n * 2 * C * HW + C * HW + r];
        lv = fminf(fmaxf(lv, -30.f), 20.f);
        out[i] = mean + expf(0.5f * lv) * noise[i];
    }
}

unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int pt_vae_time_conv_out(const float* x, int32_t ldx, const float* w_host, const float* b_host, int32_t F,
                                    int64_t HW, float* out, void* stream) {
    PT_CHECK(x && w_host && b_host && out, "pt_vae_time_conv_out: null pointer");
    PT_CHECK(ldx >= 3 && F >= 1 && F < 65536 && HW > 0, "pt_vae_time_conv_out: bad sizes (ldx=%d F=%d)", ldx, F);
    TconvW W;
    memcpy(W.w, w_host, sizeof(W.w));
    memcpy(W.b, b_host, sizeof(W.b));
    hipLaunchKernelGGL(vae_time_conv_out_kernel, dim3(grid_for(HW), (unsigned)F), dim3(256), 0, (hipStream_t)stream, x, ldx, W, F,
                       HW, out);
    PT_LAUNCH_CHECK("pt_vae_time_conv_out");
    return 0;
}

extern "C" int pt_frames_postprocess(const float* src, int32_t F, int64_t HW, int32_t mode, void* dst, void* stream) {
    PT_CHECK(src && dst, "pt_frames_postprocess: null pointer");
    PT_CHECK(F >= 1 && F < 65536 && HW > 0 && mode >= 0 && mode <= 2, "pt_frames_postprocess: bad arguments (F=%d mode=%d)", F, mode);
    hipLaunchKernelGGL(frames_postprocess_kernel, dim3(grid_for(HW), (unsigned)F), dim3(256), 0, (hipStream_t)stream, src, HW, mode,
                       dst);
    PT_LAUNCH_CHECK("pt_frames_postprocess");
    return 0;
}

extern "C" int pt_nhwc_to_nchw_f32(const float* src, int32_t N, int32_t C, int64_t HW, int32_t ld, float* dst, void* stream) {
    PT_CHECK(src && dst, "pt_nhwc_to_nchw_f32: null pointer");
    PT_CHECK(N >= 1 && N < 65536 && C >= 1 && ld >= C && HW > 0, "pt_nhwc_to_nchw_f32: bad sizes");
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(grid_for(HW), (unsigned)N), dim3(256), 0, (hipStream_t)stream, src, C, HW, ld,
                       dst);
    PT_LAUNCH_CHECK("pt_nhwc_to_nchw_f32");
    return 0;
}

extern "C" int pt_gaussian_sample(const float* params, const float* noise, int32_t N, int32_t C, int64_t HW, float* out,
                                  void* stream) {
    PT_CHECK(params && noise && out, "pt_gaussian_sample: null pointer");
    PT_CHECK(N >= 1 && C >= 1 && HW > 0, "pt_gaussian_sample: bad sizes");
    const int64_t total = (int64_t)N * C * HW;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, params, noise, C, HW, out,
                       total);
    PT_LAUNCH_CHECK("pt_gaussian_sample");
    return 0;
}
