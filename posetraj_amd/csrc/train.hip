// Element-wise pieces of the ControlNet training objective (scripts/train_svd_traj_VIPSeg_14.py:1282-1407): the
// network input of a training step and the sigma-weighted EDM loss.  fp32 arithmetic like the reference's fp32 statements.
#include "pt_common.h"

namespace {

// latents / noise fp32 [B, F, 4, h, w]; sigma[b], cond_scale[b] (= image_mask / scaling_factor) fp32 on the device.
//   noisy        = latents + noise * sigma[b]                                   (fp32 [B, F, 4, h, w], kept for the loss)
//   out[..., 0:4] = noisy / sqrt(sigma^2 + 1)                                   (fp16 channels-last [B, F, h, w, 8])
//   out[..., 4:8] = (latents[b, 0] + noise[b, 0] * aug) * cond_scale[b]         (the noise-augmented first frame, every frame)
__global__ __launch_bounds__(256) void edm_train_input_kernel(const float* __restrict__ lat, const float* __restrict__ noise,
                                                              const float* __restrict__ sigma, const float* __restrict__ cond_scale,
                                                              float aug, int F, int64_t HW, float* __restrict__ noisy,
                                                              f16* __restrict__ out, int64_t total_pix) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_pix; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % HW, bf = i / HW;
        const int64_t b = bf / F;
        const float s = sigma[b], k = 1.0f / sqrtf(s * s + 1.0f), cs = cond_scale[b];
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t e = (bf * 4 + c) * HW + p, e0 = ((b * F) * 4 + c) * HW + p;
            const float n = lat[e] + noise[e] * s;
            noisy[e] = n;
            o[c] = (f16)(n * k);
            o[4 + c] = (f16)((lat[e0] + noise[e0] * aug) * cs);
        }
        *(f16x8*)(out + i * 8) = o;
    }
}

// per sample b: mean over its n elements of w (pred * c_out + c_skip * noisy - target)^2 with c_out = -s / sqrt(s^2 + 1),
// c_skip = 1 / (s^2 + 1), w = (1 + s^2) / s^2.  pred: channels-last (fp16 or fp32) [B, F, HW, ldp], first 4 channels;
// noisy / target fp32 [B, F, 4, HW].  One block per sample, fixed-order tree reduction (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void edm_loss_kernel(const T* __restrict__ pred, int ldp, const float* __restrict__ noisy,
                                                       const float* __restrict__ target, const float* __restrict__ sigma, int F,
                                                       int64_t HW, float* __restrict__ loss) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    const float s = sigma[b], c_out = -s / sqrtf(s * s + 1.0f), c_skip = 1.0f / (s * s + 1.0f), w = (1.0f + s * s) / (s * s);
    const int64_t n = (int64_t)F * 4 * HW;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const int64_t p = i % HW, fc = i / HW;
        const int64_t f = fc / 4, c = fc % 4;
        const float pr = (float)pred[(((int64_t)b * F + f) * HW + p) * ldp + c];
        const int64_t e = (int64_t)b * n + i;
        const float d = pr * c_out + c_skip * noisy[e] - target[e];
        acc += (double)(w * d * d);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[b] = (float)(red[0] / (double)n);
}

}  // namespace

extern "C" int pt_edm_train_input(const float* latents, const float* noise, const float* sigma, const float* cond_scale, float aug,
                                  int32_t B, int32_t F, int64_t HW, float* noisy, void* out, void* stream) {
    PT_CHECK(latents && noise && sigma && cond_scale && noisy && out, "pt_edm_train_input: null pointer");
    PT_CHECK(B > 0 && F > 0 && HW > 0, "pt_edm_train_input: bad sizes");
    const int64_t total = (int64_t)B * F * HW;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(edm_train_input_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, latents, noise, sigma,
                       cond_scale, aug, F, HW, noisy, (f16*)out, total);
    PT_LAUNCH_CHECK("pt_edm_train_input");
    return 0;
}

extern "C" int pt_edm_loss(const void* pred, int32_t pred_is_f32, int32_t ldp, const float* noisy, const float* target,
                           const float* sigma, int32_t B, int32_t F, int64_t HW, float* loss, void* stream) {
    PT_CHECK(pred && noisy && target && sigma && loss, "pt_edm_loss: null pointer");
    PT_CHECK(B > 0 && B < 65536 && F > 0 && HW > 0 && ldp >= 4, "pt_edm_loss: bad sizes");
    if (pred_is_f32)
        hipLaunchKernelGGL(edm_loss_kernel<float>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (const float*)pred, ldp, noisy,
                           target, sigma, F, HW, loss);
    else
        hipLaunchKernelGGL(edm_loss_kernel<f16>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (const f16*)pred, ldp, noisy,
                           target, sigma, F, HW, loss);
    PT_LAUNCH_CHECK("pt_edm_loss");
    return 0;
}
