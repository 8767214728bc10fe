// fp32 path of AutoencoderKLTemporalDecoder.encode for gfx950 (MI355X): what `force_upcast` asks for.
//
//   The reference runs an fp16 VAE whose config says force_upcast in fp32 around encode()
//   (/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:453-462: `self.vae.to(dtype=torch.float32)` ...
//   `_encode_vae_image` ... back to fp16): weights widened exactly, every product, sum, normalisation and softmax in fp32.
//   The fp16 kernels of this library round every layer's operands to 11 bits and land 1e-3 from that; these three kernels keep
//   fp32 operands end to end.  Once per clip, ~2.5 TFLOP at 576 x 1024: built for exactness, not for the roofline.
//
//   pt_conv2d_f32      implicit-GEMM convolution / linear layer / plain A . B^T product on v_mfma_f32_16x16x4_f32 (f32 operands:
//                      bitwise an fmaf chain, MI355X_MICROARCH.md "Matrix cores"), channels-last, both operands K-contiguous;
//                      64 pixels x 64 output channels per 256-thread workgroup, K steps of 16 through LDS; bias, residual and a
//                      scale in the epilogue.  Also serves Q K^T and P V of the mid block's single-head attention (the "weights"
//                      are then an activation matrix).
//   pt_groupnorm_f32   GroupNorm (+ SiLU) over channels-last fp32, statistics in fp64, one workgroup per (sample, group) for the
//                      sums, then a flat apply pass.
//   pt_softmax_rows_f32  row softmax of fp32 scores with a scale, in place.
#include "pt_common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct ConvF32 {
    const float* x; const float* w; const float* bias; const float* res; float* out;
    int Nimg, Hin, Win, Hout, Wout, Ci, Co, KH, KW, stride, pad_h, pad_w;
    int ldx, ldw, ldo, ldr;          // elements between pixels of x / rows of w / pixels of out / pixels of res
    long long M; int K;              // M = Nimg * Hout * Wout, K = KH * KW * Ci
    float scale;
};

constexpr int CT = 64, CK = 16, CLD = CK + 4;        // tile edge, K step, padded LDS row (floats): conflict-free column reads

__global__ __launch_bounds__(256) void conv2d_f32_kernel(const ConvF32 p) {
    __shared__ float As[CT][CLD];                        // weights  [co][k]
    __shared__ float Bs[CT][CLD];                        // pixels   [px][k]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long tiles_n = (p.Co + CT - 1) / CT;
    const long long tm = blockIdx.x / tiles_n;
    const int tn = (int)(blockIdx.x - tm * tiles_n);
    const long long m0 = tm * CT;
    const int n0 = tn * CT;
    // staging: thread t moves 4 consecutive k of row t >> 2 of each tile
    const int srow = t >> 2, sk = (t & 3) * 4;
    const long long m = m0 + srow;
    int img = 0, oy = 0, ox = 0;
    const bool mvalid = m < p.M;
    if (mvalid) {
        const long long hw = (long long)p.Hout * p.Wout;
        img = (int)(m / hw);
        const int rem = (int)(m - (long long)img * hw);
        oy = rem / p.Wout; ox = rem - oy * p.Wout;
    }
    const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
    const int wrow = n0 + srow;
    const bool aligned = (p.Ci % CK == 0) && (p.ldx % 4 == 0) && (p.ldw % 4 == 0);   // a K step lies inside one tap; 16-byte loads
    const int wr = wave >> 1, wc = wave & 1;             // wave tile: 32 co x 32 px
    const int frow = lane & 15, fk = lane >> 4;
    f32x4v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < p.K; k0 += CK) {
        float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (aligned) {
            const int tap = k0 / p.Ci, ci = k0 - tap * p.Ci + sk;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const int iy = iy0 + ky, ix = ix0 + kx;
            if (mvalid && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) {
                const f32x4v v = *(const f32x4v*)(p.x + ((long long)(img * p.Hin + iy) * p.Win + ix) * p.ldx + ci);
                b4[0] = v[0]; b4[1] = v[1]; b4[2] = v[2]; b4[3] = v[3];
            }
            if (wrow < p.Co) {
                const f32x4v v = *(const f32x4v*)(p.w + (long long)wrow * p.ldw + k0 + sk);
                a4[0] = v[0]; a4[1] = v[1]; a4[2] = v[2]; a4[3] = v[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + sk + j;
                if (k < p.K) {
                    const int tap = k / p.Ci, ci = k - tap * p.Ci;
                    const int ky = tap / p.KW, kx = tap - ky * p.KW;
                    const int iy = iy0 + ky, ix = ix0 + kx;
                    if (mvalid && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win)
                        b4[j] = p.x[((long long)(img * p.Hin + iy) * p.Win + ix) * p.ldx + ci];
                    if (wrow < p.Co) a4[j] = p.w[(long long)wrow * p.ldw + k];
                }
            }
        }
        __syncthreads();                                     // the previous step's fragment reads are done
        *(f32x4v*)&As[srow][sk] = (f32x4v){a4[0], a4[1], a4[2], a4[3]};
        *(f32x4v*)&Bs[srow][sk] = (f32x4v){b4[0], b4[1], b4[2], b4[3]};
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < CK / 4; ++ks) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = As[wr * 32 + i * 16 + frow][ks * 4 + fk];
                bf[i] = Bs[wc * 32 + i * 16 + frow][ks * 4 + fk];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[co][px]: lane holds channels n0 + wr*32 + i*16 + 4 fk .. + 3 of pixel m0 + wc*32 + j*16 + frow
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long long mo = m0 + wc * 32 + j * 16 + frow;
        if (mo >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = n0 + wr * 32 + i * 16 + 4 * fk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (co + e < p.Co) {
                    float v = acc[i][j][e];
                    if (p.bias) v += p.bias[co + e];
                    v *= p.scale;
                    if (p.res) v += p.res[mo * p.ldr + co + e];
                    p.out[mo * p.ldo + co + e] = v;
                }
            }
        }
    }
}

// ---- GroupNorm: sums of one (sample, group) by one workgroup (fp64), then y = silu?((x - mean) rstd gamma + beta)
__global__ __launch_bounds__(256) void groupnorm_f32_stats_kernel(const float* __restrict__ x, long long rows, int C, int groups,
                                                                  double* __restrict__ stats) {
    const int n = blockIdx.y, g = blockIdx.x, cpg = C / groups;
    const float* base = x + (long long)n * rows * C + g * cpg;
    double s = 0.0, q = 0.0;
    const long long total = rows * cpg;
    for (long long i = threadIdx.x; i < total; i += 256) {
        const long long r = i / cpg;
        const int c = (int)(i - r * cpg);
        const double v = base[r * C + c];
        s += v; q += v * v;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = sh[0][0] / (double)total;
        double var = sh[1][0] / (double)total - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * ((long long)n * groups + g)] = mean;
        stats[2 * ((long long)n * groups + g) + 1] = var;
    }
}

__global__ __launch_bounds__(256) void groupnorm_f32_apply_kernel(const float* __restrict__ x, long long rows, int C, int groups, float eps,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const double* __restrict__ stats, int silu, long long total,
                                                                  float* __restrict__ y) {
    const int cpg = C / groups;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long n = i / ((long long)rows * C);
        const double* st = stats + 2 * (n * groups + c / cpg);
        const float mean = (float)st[0], rstd = (float)(1.0 / sqrt(st[1] + (double)eps));
        float v = (x[i] - mean) * rstd * gamma[c] + beta[c];
        if (silu) v = v / (1.0f + expf(-v));
        y[i] = v;
    }
}

__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(float* __restrict__ s, long long rows, int n, long long ld, float scale) {
    const long long r = blockIdx.x;
    float* row = s + r * ld;
    __shared__ float sh[256];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, row[i] * scale);
    sh[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]); __syncthreads(); }
    mx = sh[0];
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float e = expf(row[i] * scale - mx); row[i] = e; sum += e; }
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    const float inv = 1.0f / sh[0];
    for (int i = threadIdx.x; i < n; i += 256) row[i] *= inv;
}

}  // namespace

extern "C" int pt_conv2d_f32(const pt_conv_f32_params* q, void* stream) {
    PT_CHECK(q->x && q->w && q->out, "pt_conv2d_f32: null pointer");
    PT_CHECK(q->Nimg > 0 && q->Hout > 0 && q->Wout > 0 && q->Ci > 0 && q->Co > 0 && q->KH > 0 && q->KW > 0, "pt_conv2d_f32: empty problem");
    PT_CHECK(q->stride >= 1 && q->ldx >= q->Ci && q->ldw >= q->KH * q->KW * q->Ci && q->ldo >= q->Co && (!q->res || q->ldr >= q->Co),
             "pt_conv2d_f32: pitches");
    PT_CHECK((((uintptr_t)q->x | (uintptr_t)q->w) & 15) == 0, "pt_conv2d_f32: x and w must be 16-byte aligned");
    ConvF32 p;
    p.x = (const float*)q->x; p.w = (const float*)q->w; p.bias = (const float*)q->bias; p.res = (const float*)q->res; p.out = (float*)q->out;
    p.Nimg = q->Nimg; p.Hin = q->Hin; p.Win = q->Win; p.Hout = q->Hout; p.Wout = q->Wout; p.Ci = q->Ci; p.Co = q->Co;
    p.KH = q->KH; p.KW = q->KW; p.stride = q->stride; p.pad_h = q->pad_h; p.pad_w = q->pad_w;
    p.ldx = q->ldx; p.ldw = q->ldw; p.ldo = q->ldo; p.ldr = q->ldr;
    p.M = (long long)q->Nimg * q->Hout * q->Wout; p.K = q->KH * q->KW * q->Ci;
    p.scale = q->scale;
    const long long tiles = ((p.M + CT - 1) / CT) * ((p.Co + CT - 1) / CT);
    PT_CHECK(tiles < (1ll << 31), "pt_conv2d_f32: grid too large");
    hipLaunchKernelGGL(conv2d_f32_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    PT_LAUNCH_CHECK("pt_conv2d_f32");
    return 0;
}

extern "C" int pt_groupnorm_f32(const float* x, int64_t rows_per_sample, int32_t n_samples, int32_t C, int32_t groups, float eps,
                                const float* gamma, const float* beta, int32_t silu, double* stats_scratch, float* y, void* stream) {
    PT_CHECK(x && y && gamma && beta && stats_scratch, "pt_groupnorm_f32: null pointer");
    PT_CHECK(groups > 0 && C % groups == 0 && rows_per_sample > 0 && n_samples > 0, "pt_groupnorm_f32: C=%d groups=%d", C, groups);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(groupnorm_f32_stats_kernel, dim3(groups, n_samples), dim3(256), 0, s, x, (long long)rows_per_sample, C, groups, stats_scratch);
    const long long total = (long long)rows_per_sample * n_samples * C;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(groupnorm_f32_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, (long long)rows_per_sample, C, groups, eps, gamma, beta,
                       stats_scratch, silu, total, y);
    PT_LAUNCH_CHECK("pt_groupnorm_f32");
    return 0;
}

extern "C" int pt_softmax_rows_f32(float* scores, int64_t rows, int32_t n, int64_t ld, float scale, void* stream) {
    PT_CHECK(scores && rows > 0 && n > 0 && ld >= n, "pt_softmax_rows_f32: bad arguments");
    PT_CHECK(rows < (1ll << 31), "pt_softmax_rows_f32: too many rows");
    hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, scores, (long long)rows, n, (long long)ld, scale);
    PT_LAUNCH_CHECK("pt_softmax_rows_f32");
    return 0;
}
