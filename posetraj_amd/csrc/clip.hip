// Element-wise pieces of the CLIP vision tower (the pipeline's image_encoder): patch extraction for the patch-embedding
// convolution (kernel = stride = patch, so it is a GEMM over flattened patches) and the MLP activation.
#include "pt_common.h"

namespace {

// image [B, C, H, W] (fp32 or fp16) -> fp16 rows [B * gh * gw, ld]: row (b, py, px) = the patch's C * P * P values in
// (c, ky, kx) order - the order of Conv2d's weight [Cout, C, P, P] flattened - zero padded to ld.
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ img, int C, int H, int W, int P, int gh, int gw,
                                                       int ld, f16* __restrict__ out, int64_t total) {
    const int K = C * P * P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / ld;
        const int k = (int)(i - row * ld);
        float v = 0.f;
        if (k < K) {
            const int c = k / (P * P), r = k - c * P * P, ky = r / P, kx = r - ky * P;
            const int px = (int)(row % gw), py = (int)((row / gw) % gh);
            const int64_t b = row / ((int64_t)gw * gh);
            v = (float)img[((b * C + c) * H + py * P + ky) * W + px * P + kx];
        }
        out[i] = (f16)v;
    }
}

// kind 1: erf GELU ("gelu"); kind 2: x * sigmoid(1.702 x) ("quick_gelu")
__global__ __launch_bounds__(256) void act_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t n, int kind) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = (float)x[i];
        y[i] = (f16)(kind == 1 ? pt_gelu_erf(v) : v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * (-1.702f * 1.4426950408889634f))));
    }
}

}  // namespace

extern "C" int pt_patchify_f16(const void* img, int32_t img_is_f32, int32_t B, int32_t C, int32_t H, int32_t W, int32_t P,
                               int32_t ld, void* out, void* stream) {
    PT_CHECK(img && out, "pt_patchify_f16: null pointer");
    PT_CHECK(B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0 && ld >= C * P * P, "pt_patchify_f16: bad geometry (%d x %d, patch %d, ld %d)", H, W, P, ld);
    const int gh = H / P, gw = W / P;
    const int64_t total = (int64_t)B * gh * gw * ld;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (img_is_f32)
        hipLaunchKernelGGL(patchify_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)img, C, H, W, P, gh, gw, ld, (f16*)out, total);
    else
        hipLaunchKernelGGL(patchify_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)img, C, H, W, P, gh, gw, ld, (f16*)out, total);
    PT_LAUNCH_CHECK("pt_patchify_f16");
    return 0;
}

extern "C" int pt_act_f16(const void* x, void* y, int64_t n, int32_t kind, void* stream) {
    PT_CHECK(x && y, "pt_act_f16: null pointer");
    PT_CHECK(n > 0 && (kind == 1 || kind == 2), "pt_act_f16: kind %d (1 = gelu, 2 = quick_gelu)", kind);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, n, kind);
    PT_LAUNCH_CHECK("pt_act_f16");
    return 0;
}
