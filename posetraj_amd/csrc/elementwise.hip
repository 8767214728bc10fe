// Element-wise / layout kernels of the denoise loop (HBM-bound, 16-byte accesses where the layout allows).
#include "pt_common.h"

namespace {

__global__ __launch_bounds__(256) void axpy_kernel(const f16* __restrict__ a, const f16* __restrict__ r, float m,
                                                   f16* __restrict__ out, int64_t n8, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f16x8 x = *(const f16x8*)(a + i * 8), y = *(const f16x8*)(r + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)x[j] + m * (float)y[j]);
        *(f16x8*)(out + i * 8) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = n8 * 8 + threadIdx.x;
        out[i] = (f16)((float)a[i] + m * (float)r[i]);
    }
}

__global__ __launch_bounds__(256) void silu_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = (f16)pt_silu((float)x[i]);
}

__global__ __launch_bounds__(256) void timestep_embedding_kernel(const float* __restrict__ t, int n, int dim,
                                                                 f16* __restrict__ out) {
    const int half = dim >> 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * half) return;
    const int row = i / half, k = i - row * half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);     // ln(10000)
    const float ang = t[row] * freq;
    out[(int64_t)row * dim + k] = (f16)cosf(ang);                              // flip_sin_to_cos: cos first
    out[(int64_t)row * dim + half + k] = (f16)sinf(ang);
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T* __restrict__ src, int C, int HW, int Cpad,
                                                           f16* __restrict__ dst, int64_t total) {
    // one thread per (n, pixel, cpad); reads are strided by HW across c, coalesced across pixels via the tile below
    __shared__ float tile[32][33];
    // grid: x = pixel tiles (32), y = channel tiles (32), z = n
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                    // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        tile[k][tx] = (c < C && p < HW) ? (float)src[((int64_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        if (p < HW && c < Cpad) dst[((int64_t)n * HW + p) * Cpad + c] = (f16)tile[tx][k];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const f16* __restrict__ src, int C, int HW, int ld,
                                                           T* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        tile[k][tx] = (p < HW && c < C) ? (float)src[((int64_t)n * HW + p) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        if (c < C && p < HW) dst[((int64_t)n * C + c) * HW + p] = (T)tile[tx][k];
    }
}

__global__ __launch_bounds__(256) void concat_camera_kernel(const f16* __restrict__ feat, int C, const f16* __restrict__ cam,
                                                            int64_t pix_per_img, int Cpad, f16* __restrict__ dst,
                                                            int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / Cpad;
        const int c = (int)(i - p * Cpad);
        f16 v = (f16)0.f;
        if (c < C) v = feat[p * C + c];
        else if (c < C + 12) v = cam[(p / pix_per_img) * 12 + (c - C)];
        dst[i] = v;
    }
}

__global__ __launch_bounds__(256) void scale_concat_kernel(const float* __restrict__ lat, const f16* __restrict__ img,
                                                           float inv, int Bc, int F, int HW, f16* __restrict__ out,
                                                           int64_t total_pix) {
    // one thread per output pixel (c2, f, p): 8 channels = one 16-byte store
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_pix; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t nf = i / HW;
        const int f = (int)(nf % F);
        const int c2 = (int)(nf / F);
        const int clip = c2 % Bc;                       // torch.cat([latents] * 2): halves are [all clips][all clips]
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            o[c] = (f16)(lat[(((int64_t)clip * F + f) * 4 + c) * HW + p] * inv);
            o[4 + c] = img[((int64_t)c2 * 4 + c) * HW + p];
        }
        *(f16x8*)(out + i * 8) = o;
    }
}

// One Euler step in the reference's own fp32 operation order (scheduling_euler_discrete_karras_fix.py:504-517), with
// fused multiply-add contraction disabled: at sigma = 700 the update cancels a 700-scale sample down to O(1), so a
// differently rounded intermediate shows up at 1e-5 relative in the result.
__device__ __forceinline__ float euler_update(float mo, float x, float sigma, float c_out, float dt, int ptype) {
#pragma clang fp contract(off)
    float x0;
    if (ptype == 0) x0 = __fadd_rn(__fmul_rn(mo, c_out), __fdiv_rn(x, __fadd_rn(__fmul_rn(sigma, sigma), 1.0f)));   // v_prediction
    else if (ptype == 1) x0 = __fsub_rn(x, __fmul_rn(sigma, mo));                                                   // epsilon
    else x0 = mo;                                                                                                   // sample
    const float deriv = __fdiv_rn(__fsub_rn(x, x0), sigma);
    return __fadd_rn(x, __fmul_rn(deriv, dt));
}

template <typename T>
__global__ __launch_bounds__(256) void cfg_euler_kernel(const T* __restrict__ pred, int ldn, const float* __restrict__ guidance,
                                                        float sigma, float sigma_next, int ptype, int Bc, int F, int HW,
                                                        float* __restrict__ lat, int64_t total_pix) {
    const float c_out = -sigma / sqrtf(sigma * sigma + 1.0f);
    const float dt = sigma_next - sigma;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_pix; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t nf = i / HW;
        const int f = (int)(nf % F);
        const int clip = (int)(nf / F);
        const T* u = pred + (((int64_t)clip * F + f) * HW + p) * ldn;                   // uncond half
        const T* c = pred + (((int64_t)(Bc + clip) * F + f) * HW + p) * ldn;            // cond half
        const float g = guidance[(int64_t)clip * F + f];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const float pu = (float)u[ch], pc = (float)c[ch];
            // fp16 predictions: the guided prediction is rounded to fp16 like the reference's fp16 model output;
            // fp32 predictions (the pipeline's own U-Net call): guidance and the Euler update stay in fp32
            const float mo = sizeof(T) == 2 ? (float)(f16)(pu + g * (pc - pu)) : pu + g * (pc - pu);
            float* xp = lat + (((int64_t)clip * F + f) * 4 + ch) * HW + p;
            const float x = *xp;
            *xp = euler_update(mo, x, sigma, c_out, dt, ptype);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(const T* __restrict__ x, float k, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = (T)((float)x[i] * k);
}

template <typename T>
__global__ __launch_bounds__(256) void euler_flat_kernel(const T* __restrict__ mo, const float* __restrict__ x, float sigma,
                                                         float sigma_next, int ptype, float* __restrict__ out, int64_t n) {
    const float c_out = -sigma / sqrtf(sigma * sigma + 1.0f);
    const float dt = sigma_next - sigma;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = euler_update((float)mo[i], x[i], sigma, c_out, dt, ptype);
}

// EulerDiscreteScheduler.add_noise: y = x + noise * sigma[sample], in the tensor's own dtype like the reference
// (fp16: the product and the sum each round to fp16)
template <typename T>
__global__ __launch_bounds__(256) void add_noise_kernel(const T* __restrict__ x, const T* __restrict__ noise,
                                                        const float* __restrict__ sigma, int64_t per_sample, T* __restrict__ y,
                                                        int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const T s = (T)sigma[i / per_sample];
        float prod = (float)(T)((float)noise[i] * (float)s);
        asm volatile("" : "+v"(prod));                                    // keeps the product a rounded value of its own:
        y[i] = (T)((float)x[i] + prod);                                  // two roundings like torch, never one FMA
    }
}

unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int pt_axpy_f16(const void* a, const void* r, float m, void* out, int64_t n, void* stream) {
    PT_CHECK(a && r && out && n > 0, "pt_axpy_f16: bad arguments");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, (const f16*)a,
                       (const f16*)r, m, (f16*)out, n / 8, n);
    PT_LAUNCH_CHECK("pt_axpy_f16");
    return 0;
}

extern "C" int pt_silu_f16(const void* x, void* y, int64_t n, void* stream) {
    PT_CHECK(x && y && n > 0, "pt_silu_f16: bad arguments");
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, n);
    PT_LAUNCH_CHECK("pt_silu_f16");
    return 0;
}

extern "C" int pt_timestep_embedding(const float* t, int32_t n, int32_t dim, void* out, void* stream) {
    PT_CHECK(t && out && n > 0 && dim > 0 && dim % 2 == 0, "pt_timestep_embedding: bad arguments");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n * (dim / 2) + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       t, n, dim, (f16*)out);
    PT_LAUNCH_CHECK("pt_timestep_embedding");
    return 0;
}

extern "C" int pt_nchw_to_nhwc_f16(const void* src, int32_t src_is_f32, int32_t N, int32_t C, int32_t H, int32_t W,
                                   int32_t Cpad, void* dst, void* stream) {
    PT_CHECK(src && dst && N > 0 && C > 0 && Cpad >= C, "pt_nchw_to_nhwc_f16: bad arguments");
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (Cpad + 31) / 32, N);
    PT_CHECK(grid.y <= 65535 && grid.z <= 65535, "pt_nchw_to_nhwc_f16: grid too large");
    if (src_is_f32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, C, HW, Cpad, (f16*)dst, 0);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)src, C, HW, Cpad, (f16*)dst, 0);
    PT_LAUNCH_CHECK("pt_nchw_to_nhwc_f16");
    return 0;
}

extern "C" int pt_nhwc_to_nchw(const void* src, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ld, void* dst,
                               int32_t dst_is_f32, void* stream) {
    PT_CHECK(src && dst && N > 0 && C > 0 && ld >= C, "pt_nhwc_to_nchw: bad arguments");
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
    PT_CHECK(grid.y <= 65535 && grid.z <= 65535, "pt_nhwc_to_nchw: grid too large");
    if (dst_is_f32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)src, C, HW, ld, (float*)dst);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)src, C, HW, ld, (f16*)dst);
    PT_LAUNCH_CHECK("pt_nhwc_to_nchw");
    return 0;
}

extern "C" int pt_concat_camera(const void* feat, int32_t C, const void* cam, int32_t n_img, int64_t pix_per_img,
                                int32_t Cpad, void* dst, void* stream) {
    PT_CHECK(feat && cam && dst && Cpad >= C + 12, "pt_concat_camera: bad arguments");
    const int64_t total = (int64_t)n_img * pix_per_img * Cpad;
    hipLaunchKernelGGL(concat_camera_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f16*)feat, C,
                       (const f16*)cam, pix_per_img, Cpad, (f16*)dst, total);
    PT_LAUNCH_CHECK("pt_concat_camera");
    return 0;
}

extern "C" int pt_scale_concat_input(const float* latents, const void* image_latents, float sigma, int32_t Bc, int32_t F,
                                     int32_t h, int32_t w, void* out, void* stream) {
    PT_CHECK(latents && image_latents && out && Bc > 0 && F > 0, "pt_scale_concat_input: bad arguments");
    const int64_t total = (int64_t)2 * Bc * F * h * w;
    const float inv = 1.0f / sqrtf(sigma * sigma + 1.0f);
    hipLaunchKernelGGL(scale_concat_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, latents,
                       (const f16*)image_latents, inv, Bc, F, h * w, (f16*)out, total);
    PT_LAUNCH_CHECK("pt_scale_concat_input");
    return 0;
}

extern "C" int pt_cfg_euler_step(const void* noise_pred, int32_t np_is_f32, int32_t ldn, const float* guidance, float sigma,
                                 float sigma_next, int32_t prediction_type, int32_t Bc, int32_t F, int32_t h, int32_t w,
                                 float* latents, void* stream) {
    PT_CHECK(noise_pred && guidance && latents && Bc > 0 && F > 0 && ldn >= 4, "pt_cfg_euler_step: bad arguments");
    PT_CHECK(prediction_type >= 0 && prediction_type <= 2, "pt_cfg_euler_step: prediction_type %d", prediction_type);
    PT_CHECK(sigma > 0.f, "pt_cfg_euler_step: sigma must be > 0");
    const int64_t total = (int64_t)Bc * F * h * w;
    if (np_is_f32)
        hipLaunchKernelGGL(cfg_euler_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)noise_pred, ldn, guidance, sigma, sigma_next, prediction_type, Bc, F, h * w, latents, total);
    else
        hipLaunchKernelGGL(cfg_euler_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const f16*)noise_pred, ldn, guidance, sigma, sigma_next, prediction_type, Bc, F, h * w, latents, total);
    PT_LAUNCH_CHECK("pt_cfg_euler_step");
    return 0;
}

extern "C" int pt_scale(const void* x, int32_t is_f32, float k, void* y, int64_t n, void* stream) {
    PT_CHECK(x && y && n > 0, "pt_scale: bad arguments");
    if (is_f32) hipLaunchKernelGGL(scale_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, k, (float*)y, n);
    else        hipLaunchKernelGGL(scale_kernel<f16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, k, (f16*)y, n);
    PT_LAUNCH_CHECK("pt_scale");
    return 0;
}

extern "C" int pt_add_noise(const void* x, const void* noise, int32_t is_f32, const float* sigma_per_sample, int64_t per_sample,
                            void* y, int64_t n, void* stream) {
    PT_CHECK(x && noise && sigma_per_sample && y && n > 0 && per_sample > 0, "pt_add_noise: bad arguments");
    if (is_f32) hipLaunchKernelGGL(add_noise_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)noise, sigma_per_sample, per_sample, (float*)y, n);
    else        hipLaunchKernelGGL(add_noise_kernel<f16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (const f16*)noise, sigma_per_sample, per_sample, (f16*)y, n);
    PT_LAUNCH_CHECK("pt_add_noise");
    return 0;
}

extern "C" int pt_euler_step(const void* model_output, int32_t mo_is_f32, const float* sample, float sigma, float sigma_next,
                             int32_t prediction_type, float* prev_sample, int64_t n, void* stream) {
    PT_CHECK(model_output && sample && prev_sample && n > 0, "pt_euler_step: bad arguments");
    PT_CHECK(prediction_type >= 0 && prediction_type <= 2, "pt_euler_step: prediction_type %d", prediction_type);
    PT_CHECK(sigma > 0.f, "pt_euler_step: sigma must be > 0");
    if (mo_is_f32) hipLaunchKernelGGL(euler_flat_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)model_output, sample, sigma, sigma_next, prediction_type, prev_sample, n);
    else           hipLaunchKernelGGL(euler_flat_kernel<f16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const f16*)model_output, sample, sigma, sigma_next, prediction_type, prev_sample, n);
    PT_LAUNCH_CHECK("pt_euler_step");
    return 0;
}


// ------------------------------------------------------------------------------------------ pre-loop image resize
// _resize_with_antialiasing (pipeline/...controlnet.py:604-712): once per clip on one 3-channel image - HBM-bound, tiny.
namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {   // F.pad(mode = "reflect"): -1 -> 1, n -> n - 2
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// cross-correlation along x (axis = 1) or y (axis = 0) with reflect padding; pad_front = (k - 1) / 2 (_compute_padding)
__global__ __launch_bounds__(256) void blur1d_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes, int H,
                                                     int W, const float* __restrict__ taps, int k, int axis) {
    const long long total = (long long)planes * H * W;
    const int front = (k - 1) / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float* base = src + (i - x - (long long)y * W);
        float acc = 0.f;
        for (int t = 0; t < k; ++t) {
            const float v = axis ? base[(long long)y * W + reflect_idx(x + t - front, W)]
                                 : base[(long long)reflect_idx(y + t - front, H) * W + x];
            acc += taps[t] * v;
        }
        dst[i] = acc;
    }
}

__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {   // PyTorch's A = -0.75 convolution kernel
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x1 = t, x2 = 1.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// F.interpolate(mode = "bicubic", align_corners = True): src = dst * (in - 1) / (out - 1); x taps first, then y (ATen order)
__global__ __launch_bounds__(256) void bicubic_ac_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes,
                                                         int H, int W, int oh, int ow, float sy, float sx) {
    // sy, sx = (in - 1) / (out - 1) come from the HOST: the device's default fp32 division is not correctly rounded, and one
    // ulp in the scale is 1e-5 in the source coordinate of the last rows (3.5e-5 in the output against the reference)
    const long long total = (long long)planes * oh * ow;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
        const float* base = src + (i / ((long long)oh * ow)) * (long long)H * W;
        // source coordinate and fraction with the reference's roundings: product rounded, THEN floor and subtract.  hipcc
        // contracts `sy * oy - iy` into one fma (also through __fmul_rn / __fsub_rn), which keeps the product's rounding error
        // in the fraction: 1e-5 in t where the coordinate lands near an integer, 3.5e-5 in the output of the 160 -> 224 case
        float wy[4], wx[4];
        int iy, ix;
        {
#pragma clang fp contract(off)
            const float fy = sy * (float)oy, fx = sx * (float)ox;
            iy = (int)floorf(fy); ix = (int)floorf(fx);
            const float ty = fy - (float)iy, tx = fx - (float)ix;
            cubic_coeffs(ty, wy);
            cubic_coeffs(tx, wx);
        }
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), H - 1);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) row += wx[b] * base[(long long)yy * W + min(max(ix - 1 + b, 0), W - 1)];
            acc += wy[a] * row;
        }
        dst[i] = acc;
    }
}

}  // namespace

extern "C" int pt_resize_antialias_f32(const float* src, int32_t planes, int32_t H, int32_t W, int32_t oh, int32_t ow,
                                       const float* taps_x, int32_t kx, const float* taps_y, int32_t ky, float* tmp, float* dst,
                                       void* stream) {
    PT_CHECK(src && taps_x && taps_y && tmp && dst, "pt_resize_antialias_f32: null pointer");
    PT_CHECK(planes > 0 && H > 1 && W > 1 && oh > 0 && ow > 0, "pt_resize_antialias_f32: bad sizes %d x %d x %d -> %d x %d", planes, H, W, oh, ow);
    PT_CHECK(kx >= 1 && ky >= 1 && kx < 2 * W && ky < 2 * H, "pt_resize_antialias_f32: kernel sizes %d x %d exceed what reflect padding allows", ky, kx);
    hipStream_t s = (hipStream_t)stream;
    const long long n_in = (long long)planes * H * W, n_out = (long long)planes * oh * ow;
    const unsigned b_in = (unsigned)((n_in + 255) / 256 < 4096 ? (n_in + 255) / 256 : 4096);
    const unsigned b_out = (unsigned)((n_out + 255) / 256 < 4096 ? (n_out + 255) / 256 : 4096);
    hipLaunchKernelGGL(blur1d_kernel, dim3(b_in), dim3(256), 0, s, src, tmp, planes, H, W, taps_x, kx, 1);
    hipLaunchKernelGGL(blur1d_kernel, dim3(b_in), dim3(256), 0, s, (const float*)tmp, tmp + n_in, planes, H, W, taps_y, ky, 0);
    const float sy = oh > 1 ? (float)(H - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(W - 1) / (float)(ow - 1) : 0.f;
    hipLaunchKernelGGL(bicubic_ac_kernel, dim3(b_out), dim3(256), 0, s, (const float*)(tmp + n_in), dst, planes, H, W, oh, ow, sy, sx);
    PT_LAUNCH_CHECK("pt_resize_antialias_f32");
    return 0;
}
