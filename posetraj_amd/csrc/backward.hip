// Backward kernels of the ControlNet training step (scripts/train_svd_traj_VIPSeg_14.py:1414-1425:
// `accelerator.backward(loss)`, `optimizer.step()`): everything of the reverse pass that is not a matrix product
// (those are pt_igemm_f16 with a transposed pack - data gradients - and pt_gemm_f16 - weight gradients and attention).
// HBM-bound passes over channels-last fp16 activations with fp32 statistics; parameter gradients are ACCUMULATED into fp32
// buffers with atomics (the caller zeroes them once per optimizer step, so gradient accumulation over micro-batches is free).
#include "pt_common.h"

namespace {

constexpr int TX = 32, TY = 8;           // thread (ty, tx): chunk column tx (8 channels) of a 256-channel strip, rows ty, ty + 8, ...

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dsilu(float z) { const float s = sigmoidf_(z); return s * (1.0f + z * (1.0f - s)); }

// fold the (ty) partials of a strip's 256 channels: red[ty][ch][2] -> out through f(channel_in_strip, a, b)
template <typename F>
__device__ __forceinline__ void fold_strip(float (*red)[256 * 2], const float* s, const float* q, F f) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[ty][(tx * 8 + j) * 2] = s[j]; red[ty][(tx * 8 + j) * 2 + 1] = q[j]; }
    __syncthreads();
    const int ch = threadIdx.x;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int y = 0; y < TY; ++y) { a += red[y][ch * 2]; b += red[y][ch * 2 + 1]; }
    f(ch, a, b);
}

// ------------------------------------------------------------------------------------------ GroupNorm backward
// y = silu?(xh * gamma + beta), xh = (x - mean_g) * rstd_g over the (rows_per_sample x C/groups) elements of a group.
//   g  = dy * silu'(z)                                   dgamma[c] += sum g xh      dbeta[c] += sum g
//   dx = rstd (g gamma - (s1 + xh s2) / n),  s1 = sum_group g gamma,  s2 = sum_group g gamma xh
// stat[sample][group][4] = (sum x, sum x^2, s1, s2) in fp32.  Three passes: MODE 0 (x statistics), MODE 1 (s1, s2 and the
// parameter gradients), apply.  grid (slabs, strips, samples).
template <int MODE>
__global__ __launch_bounds__(256) void gnb_reduce_kernel(const f16* __restrict__ x0, const f16* __restrict__ x1, int C0, int C1, int groups,
                                                         int64_t rows_per_sample, int rows_per_slab, float eps,
                                                         const f16* __restrict__ gamma, const f16* __restrict__ beta, int silu,
                                                         const f16* __restrict__ dy, float* __restrict__ stat,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[TY][256 * 2];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int slab = blockIdx.x, strip = blockIdx.y, sample = blockIdx.z;
    const int Ct = C0 + C1, cg = Ct / groups;
    const int c = strip * 256 + tx * 8;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (c < Ct) {
        const f16* src; int ld, co;
        if (c < C0) { src = x0; ld = C0; co = c; } else { src = x1; ld = C1; co = c - C0; }
        const int64_t r0 = (int64_t)slab * rows_per_slab;
        int64_t r1 = r0 + rows_per_slab; if (r1 > rows_per_sample) r1 = rows_per_sample;
        const int64_t row_base = (int64_t)sample * rows_per_sample;
        float mean[8], rstd[8], ga[8], be[8];
        if (MODE == 1) {
            const double cnt = (double)rows_per_sample * cg;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int g = (c + j) / cg;
                const float* st = stat + ((int64_t)sample * groups + g) * 4;
                const double m = (double)st[0] / cnt;
                double var = (double)st[1] / cnt - m * m;
                if (var < 0.0) var = 0.0;
                mean[j] = (float)m; rstd[j] = (float)(1.0 / sqrt(var + (double)eps));
                ga[j] = (float)gamma[c + j]; be[j] = (float)beta[c + j];
            }
        }
        for (int64_t r = r0 + ty; r < r1; r += TY) {
            const f16x8 v = *(const f16x8*)(src + (row_base + r) * ld + co);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s[j] += f; q[j] += f * f; }
            } else {
                const f16x8 d = *(const f16x8*)(dy + (row_base + r) * Ct + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)v[j] - mean[j]) * rstd[j];
                    float g = (float)d[j];
                    if (silu) g *= dsilu(xh * ga[j] + be[j]);
                    s[j] += g; q[j] += g * xh;
                }
            }
        }
    }
    // per-channel totals of the strip -> LDS; then one thread per (group touched by the strip, statistic) folds its channels:
    // two atomics per group and block instead of two per channel (the group sums were 10 x more contended than the rest)
    __shared__ float chan[256 * 2];
    fold_strip(red, s, q, [&](int ch, float a, float b) {
        const int cc = strip * 256 + ch;
        float wa = a, wb = b;
        if (MODE == 1 && cc < Ct) {
            const float gm = (float)gamma[cc];
            wa = gm * a; wb = gm * b;
            if (dgamma) { atomicAdd(dgamma + cc, b); atomicAdd(dbeta + cc, a); }
        }
        chan[ch * 2] = cc < Ct ? wa : 0.f;
        chan[ch * 2 + 1] = cc < Ct ? wb : 0.f;
    });
    __syncthreads();
    const int c_lo = strip * 256, c_hi = (c_lo + 256 < Ct) ? c_lo + 256 : Ct;
    const int g_lo = c_lo / cg, g_hi = (c_hi - 1) / cg;
    const int ng = g_hi - g_lo + 1;
    for (int i = threadIdx.x; i < 2 * ng; i += 256) {
        const int g = g_lo + (i >> 1), which = i & 1;
        int a0 = g * cg, a1 = a0 + cg;
        if (a0 < c_lo) a0 = c_lo;
        if (a1 > c_hi) a1 = c_hi;
        float acc = 0.f;
        for (int ch = a0; ch < a1; ++ch) acc += chan[(ch - c_lo) * 2 + which];
        atomicAdd(stat + ((int64_t)sample * groups + g) * 4 + (MODE == 0 ? 0 : 2) + which, acc);
    }
}

__global__ __launch_bounds__(256) void gnb_apply_kernel(const f16* __restrict__ x0, const f16* __restrict__ x1, int C0, int C1, int groups,
                                                        int64_t rows_per_sample, int rows_per_slab, float eps,
                                                        const f16* __restrict__ gamma, const f16* __restrict__ beta, int silu,
                                                        const f16* __restrict__ dy, const float* __restrict__ stat,
                                                        f16* __restrict__ dx0, f16* __restrict__ dx1) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int slab = blockIdx.x, strip = blockIdx.y, sample = blockIdx.z;
    const int Ct = C0 + C1, cg = Ct / groups;
    const int c = strip * 256 + tx * 8;
    if (c >= Ct) return;
    const f16* src; f16* dst; int ld, co;
    if (c < C0) { src = x0; dst = dx0; ld = C0; co = c; } else { src = x1; dst = dx1; ld = C1; co = c - C0; }
    float mean[8], rstd[8], ga[8], be[8], k1[8], k2[8];
    const double cnt = (double)rows_per_sample * cg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* st = stat + ((int64_t)sample * groups + (c + j) / cg) * 4;
        const double m = (double)st[0] / cnt;
        double var = (double)st[1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        mean[j] = (float)m; rstd[j] = (float)(1.0 / sqrt(var + (double)eps));
        ga[j] = (float)gamma[c + j]; be[j] = (float)beta[c + j];
        k1[j] = (float)((double)st[2] / cnt); k2[j] = (float)((double)st[3] / cnt);
    }
    const int64_t r0 = (int64_t)slab * rows_per_slab;
    int64_t r1 = r0 + rows_per_slab; if (r1 > rows_per_sample) r1 = rows_per_sample;
    const int64_t row_base = (int64_t)sample * rows_per_sample;
    for (int64_t r = r0 + ty; r < r1; r += TY) {
        const f16x8 v = *(const f16x8*)(src + (row_base + r) * ld + co);
        const f16x8 d = *(const f16x8*)(dy + (row_base + r) * Ct + c);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = ((float)v[j] - mean[j]) * rstd[j];
            float g = (float)d[j];
            if (silu) g *= dsilu(xh * ga[j] + be[j]);
            o[j] = (f16)(rstd[j] * (g * ga[j] - k1[j] - xh * k2[j]));
        }
        *(f16x8*)(dst + (row_base + r) * ld + co) = o;
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm backward
// one wave per row; a block of 4 waves owns 64 rows.  A lane keeps the dgamma / dbeta contributions of ITS channels (chunk
// lane, lane + 64, ...: C <= 1536) in registers over the wave's 16 rows, the four waves meet in LDS, one global atomic per
// channel and block.
constexpr int LN_MAXCH = 3;                          // chunks of 8 channels per lane (C <= 1536)
__global__ __launch_bounds__(256) void lnb_kernel(const f16* __restrict__ x, int64_t M, int C, const f16* __restrict__ gamma, float eps,
                                                  const f16* __restrict__ dy, f16* __restrict__ dx, float* __restrict__ dgamma,
                                                  float* __restrict__ dbeta, int rows_per_block) {
    extern __shared__ float lds[];                 // [2][C] when dgamma
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int CH = C >> 3;
    float pg[LN_MAXCH][8], pb[LN_MAXCH][8];
#pragma unroll
    for (int u = 0; u < LN_MAXCH; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) { pg[u][j] = 0.f; pb[u][j] = 0.f; }
    if (dgamma) {
        for (int i = threadIdx.x; i < 2 * C; i += 256) lds[i] = 0.f;
        __syncthreads();
    }
    const int64_t rb = (int64_t)blockIdx.x * rows_per_block;
    for (int rr = wave; rr < rows_per_block; rr += 4) {
        const int64_t r = rb + rr;
        if (r >= M) break;
        const f16* xr = x + r * C;
        const f16* dr = dy + r * C;
        f16x8 xv[LN_MAXCH], dv[LN_MAXCH];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int u = 0; u < LN_MAXCH; ++u) {
            const int ch = lane + 64 * u;
            if (ch < CH) {
                xv[u] = *(const f16x8*)(xr + ch * 8);
                dv[u] = *(const f16x8*)(dr + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)xv[u][j]; s += f; q += f * f; }
            }
        }
        s = pt_wave_sum(s); q = pt_wave_sum(q);
        const float mean = s / C;
        float var = q / C - mean * mean; if (var < 0.f) var = 0.f;
        const float rstd = 1.0f / sqrtf(var + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int u = 0; u < LN_MAXCH; ++u) {
            const int ch = lane + 64 * u;
            if (ch < CH) {
                const f16x8 gm = *(const f16x8*)(gamma + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[u][j] - mean) * rstd, g = (float)dv[u][j] * (float)gm[j];
                    a += g; b += g * xh;
                }
            }
        }
        a = pt_wave_sum(a) / C; b = pt_wave_sum(b) / C;
#pragma unroll
        for (int u = 0; u < LN_MAXCH; ++u) {
            const int ch = lane + 64 * u;
            if (ch < CH) {
                const f16x8 gm = *(const f16x8*)(gamma + ch * 8);
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[u][j] - mean) * rstd, dyj = (float)dv[u][j];
                    o[j] = (f16)(rstd * (dyj * (float)gm[j] - a - xh * b));
                    pg[u][j] += dyj * xh; pb[u][j] += dyj;
                }
                *(f16x8*)(dx + r * C + ch * 8) = o;
            }
        }
    }
    if (dgamma) {
#pragma unroll
        for (int u = 0; u < LN_MAXCH; ++u) {
            const int ch = lane + 64 * u;
            if (ch < CH)
#pragma unroll
                for (int j = 0; j < 8; ++j) { atomicAdd(&lds[ch * 8 + j], pg[u][j]); atomicAdd(&lds[C + ch * 8 + j], pb[u][j]); }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < C; i += 256) { atomicAdd(dgamma + i, lds[i]); atomicAdd(dbeta + i, lds[C + i]); }
    }
}

// Narrow rows: LPR lanes per row (3 chunks of 8 channels each), 64 / LPR rows of a wave in flight together - the one-wave-per-row
// kernel above is latency-bound there (two dependent wave reductions per 640-byte row: 61 us for [40320, 320] against 8 us for
// the forward).  With parameter gradients it also writes (mean, rstd) per row for lnb_param_kernel below - LDS atomics from the
// 4-8 rows a wave has in flight serialise (166 us), register partials push the kernel to 2 waves per SIMD (61 us); two lean
// passes take 21 + 20.
template <int LPR, bool PARAMS>
__global__ __launch_bounds__(256) void lnb_narrow_kernel(const f16* __restrict__ x, int64_t M, int C, const f16* __restrict__ gamma, float eps,
                                                         const f16* __restrict__ dy, f16* __restrict__ dx, float* __restrict__ rowstat,
                                                         int rows_per_block) {
    constexpr int NCH = 3, RPW = 64 / LPR;               // chunks of 8 channels per lane; rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % LPR, grp = lane / LPR;
    const int CH = C >> 3;
    const int64_t rb = (int64_t)blockIdx.x * rows_per_block;
    const float invC = 1.0f / C;
    for (int rr = wave * RPW + grp; rr < rows_per_block; rr += 4 * RPW) {
        const int64_t r = rb + rr;
        const bool ok = r < M;                            // lanes of a missing row still take part in the shuffles
        f16x8 xv[NCH], dv[NCH];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int ch = li + LPR * u;
            if (ok && ch < CH) {
                xv[u] = *(const f16x8*)(x + r * C + ch * 8);
                dv[u] = *(const f16x8*)(dy + r * C + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)xv[u][j]; s += f; q += f * f; }
            }
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        const float mean = s * invC;
        float var = q * invC - mean * mean; if (var < 0.f) var = 0.f;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (PARAMS && ok && li == 0) *(f32x2*)(rowstat + 2 * r) = f32x2{mean, rstd};
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int ch = li + LPR * u;
            if (ok && ch < CH) {
                const f16x8 gm = *(const f16x8*)(gamma + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[u][j] - mean) * rstd, g = (float)dv[u][j] * (float)gm[j];
                    a += g; b += g * xh;
                }
            }
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        a *= invC; b *= invC;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int ch = li + LPR * u;
            if (ok && ch < CH) {
                const f16x8 gm = *(const f16x8*)(gamma + ch * 8);
                f16x8 o8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[u][j] - mean) * rstd, dyj = (float)dv[u][j];
                    o8[j] = (f16)(rstd * (dyj * (float)gm[j] - a - xh * b));
                }
                *(f16x8*)(dx + r * C + ch * 8) = o8;
            }
        }
    }
}

// dgamma[c] += sum_rows dy xh, dbeta[c] += sum_rows dy with (mean, rstd) per row from the pass above; grid (slabs, strips)
__global__ __launch_bounds__(256) void lnb_param_kernel(const f16* __restrict__ x, const f16* __restrict__ dy, const float* __restrict__ rowstat,
                                                        int64_t M, int rows_per_slab, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[TY][256 * 2];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int slab = blockIdx.x, strip = blockIdx.y;
    const int c = strip * 256 + tx * 8;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (c < C) {
        const int64_t r0 = (int64_t)slab * rows_per_slab;
        int64_t r1 = r0 + rows_per_slab; if (r1 > M) r1 = M;
        for (int64_t r = r0 + ty; r < r1; r += TY) {
            const f16x8 v = *(const f16x8*)(x + r * C + c), d = *(const f16x8*)(dy + r * C + c);
            const f32x2 st = *(const f32x2*)(rowstat + 2 * r);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dj = (float)d[j]; s[j] += dj; q[j] += dj * ((float)v[j] - st[0]) * st[1]; }
        }
    }
    fold_strip(red, s, q, [&](int ch, float a, float b) {
        const int cc = strip * 256 + ch;
        if (cc < C) { atomicAdd(dbeta + cc, a); atomicAdd(dgamma + cc, b); }
    });
}

inline int slab_rows(int64_t rows_per_sample, int64_t other_blocks);

template <int LPR>
void launch_lnb_narrow(const f16* x, int64_t M, int C, const f16* gamma, float eps, const f16* dy, f16* dx, float* dgamma, float* dbeta,
                       float* rowstat, hipStream_t s) {
    const int rpb = 64;
    const unsigned blocks = (unsigned)((M + rpb - 1) / rpb);
    if (dgamma) {
        hipLaunchKernelGGL((lnb_narrow_kernel<LPR, true>), dim3(blocks), dim3(256), 0, s, x, M, C, gamma, eps, dy, dx, rowstat, rpb);
        const int strips = (C + 255) / 256;
        const int rps = slab_rows(M, strips);
        hipLaunchKernelGGL(lnb_param_kernel, dim3((unsigned)((M + rps - 1) / rps), strips), dim3(256), 0, s, x, dy, (const float*)rowstat, M, rps, C, dgamma,
                           dbeta);
    } else {
        hipLaunchKernelGGL((lnb_narrow_kernel<LPR, false>), dim3(blocks), dim3(256), 0, s, x, M, C, gamma, eps, dy, dx, rowstat, rpb);
    }
}

// ------------------------------------------------------------------------------------------ segmented column sums
// out[seg, c] += sum over the rows of segment seg of dy[row, c]: bias gradients (one segment), the gradient of a per-frame /
// per-clip row vector broadcast over its rows (time-embedding rows, collapsed cross-attention, frame position embedding).
__global__ __launch_bounds__(256) void colsum_kernel(const f16* __restrict__ dy, int64_t rows_per_seg, int rows_per_slab, int C, int ld,
                                                     float* __restrict__ out) {
    __shared__ float red[TY][256 * 2];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int slab = blockIdx.x, strip = blockIdx.y, seg = blockIdx.z;
    const int c = strip * 256 + tx * 8;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (c < C) {
        const int64_t r0 = (int64_t)slab * rows_per_slab;
        int64_t r1 = r0 + rows_per_slab; if (r1 > rows_per_seg) r1 = rows_per_seg;
        const f16* base = dy + (int64_t)seg * rows_per_seg * ld + c;
        int64_t r = r0 + ty;
        if (c + 8 <= C) {
            for (; r + 3 * TY < r1; r += 4 * TY) {           // four rows' loads in flight: one per iteration is a latency chain
                f16x8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const f16x8*)(base + (r + u * TY) * ld);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[j] += (float)v[u][j];
            }
            for (; r < r1; r += TY) {
                const f16x8 v = *(const f16x8*)(base + r * ld);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
            }
        } else {
            for (; r < r1; r += TY) {
                const f16* src = base + r * ld;
                for (int j = 0; j < 8; ++j) if (c + j < C) s[j] += (float)src[j];
            }
        }
    }
    fold_strip(red, s, q, [&](int ch, float a, float) {
        const int cc = strip * 256 + ch;
        if (cc < C) atomicAdd(out + (int64_t)seg * C + cc, a);
    });
}

// ------------------------------------------------------------------------------------------ softmax rows (attention backward)
// one wave per row of n fp32 scores (already scaled): P = softmax(S) as fp16;  dS = P (dP - sum_j P_j dP_j) as fp16.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t rows, int n, int64_t ld, f16* __restrict__ P,
                                                           int64_t ldp) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* s = S + r * ld;
    float m = -3.0e38f;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, s[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float z = 0.f;
    for (int j = lane; j < n; j += 64) z += __expf(s[j] - m);
    z = pt_wave_sum(z);
    const float inv = 1.0f / z;
    for (int j = lane; j < n; j += 64) P[r * ldp + j] = (f16)(__expf(s[j] - m) * inv);
}

__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const f16* __restrict__ P, int64_t ldp, const float* __restrict__ dP, int64_t ld,
                                                               int64_t rows, int n, f16* __restrict__ dS, int64_t lds_) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float dot = 0.f;
    for (int j = lane; j < n; j += 64) dot += (float)P[r * ldp + j] * dP[r * ld + j];
    dot = pt_wave_sum(dot);
    for (int j = lane; j < n; j += 64) dS[r * lds_ + j] = (f16)((float)P[r * ldp + j] * (dP[r * ld + j] - dot));
}

// ------------------------------------------------------------------------------------------ element-wise
__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// GEGLU (diffusers GEGLU.forward: hidden, gate = proj.chunk(2, -1); hidden * gelu(gate)): h [M, 2 I] -> y [M, I]
__global__ __launch_bounds__(256) void geglu_kernel(const f16* __restrict__ h, int64_t M, int I, f16* __restrict__ y) {
    const int64_t n8 = M * (I >> 3);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (I >> 3);
        const int c = (int)(i % (I >> 3)) * 8;
        const f16x8 v = *(const f16x8*)(h + r * 2 * I + c), g = *(const f16x8*)(h + r * 2 * I + I + c);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)v[j] * pt_gelu_erf((float)g[j]));
        *(f16x8*)(y + r * I + c) = o;
    }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const f16* __restrict__ h, const f16* __restrict__ dy, int64_t M, int I,
                                                        f16* __restrict__ dh) {
    const int64_t n8 = M * (I >> 3);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (I >> 3);
        const int c = (int)(i % (I >> 3)) * 8;
        const f16x8 v = *(const f16x8*)(h + r * 2 * I + c), g = *(const f16x8*)(h + r * 2 * I + I + c), d = *(const f16x8*)(dy + r * I + c);
        f16x8 dv, dg;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gj = (float)g[j], dj = (float)d[j];
            dv[j] = (f16)(dj * gelu_erf_exact(gj));
            dg[j] = (f16)(dj * (float)v[j] * dgelu_erf(gj));
        }
        *(f16x8*)(dh + r * 2 * I + c) = dv;
        *(f16x8*)(dh + r * 2 * I + I + c) = dg;
    }
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(const f16* __restrict__ x, const f16* __restrict__ dy, int64_t n, f16* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = (f16)((float)dy[i] * dsilu((float)x[i]));
}

// out = alpha a + (1 - alpha) b      (AlphaBlender with the clip-wide scalar alpha = sigmoid(mix_factor))
__global__ __launch_bounds__(256) void lerp_kernel(const f16* __restrict__ a, const f16* __restrict__ b, float alpha, int64_t n8,
                                                   f16* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f16x8 x = *(const f16x8*)(a + i * 8), y = *(const f16x8*)(b + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)(alpha * (float)x[j] + (1.0f - alpha) * (float)y[j]);
        *(f16x8*)(out + i * 8) = o;
    }
}

// out += scale * sum_i dy_i (a_i - b_i)     (gradient of the blend weight; fp32, one atomic per block)
__global__ __launch_bounds__(256) void dot_diff_kernel(const f16* __restrict__ dy, const f16* __restrict__ a, const f16* __restrict__ b,
                                                       int64_t n, float scale, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        acc += (float)dy[i] * ((float)a[i] - (b ? (float)b[i] : 0.f));
    acc = pt_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, scale * (red[0] + red[1] + red[2] + red[3]));
}

// The same three with the blend weight read from DEVICE memory (round 6: the training step as a hipGraph - a captured launch cannot take a
// host float that changes every step; alpha = sigmoid(mix_factor) is produced by sigmoid_gather_kernel inside the same graph).
__global__ __launch_bounds__(256) void lerp_dev_kernel(const f16* __restrict__ a, const f16* __restrict__ b, const float* __restrict__ alpha_p,
                                                       int64_t n8, f16* __restrict__ out) {
    const float alpha = *alpha_p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f16x8 x = *(const f16x8*)(a + i * 8), y = *(const f16x8*)(b + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)(alpha * (float)x[j] + (1.0f - alpha) * (float)y[j]);
        *(f16x8*)(out + i * 8) = o;
    }
}

__global__ __launch_bounds__(256) void dot_diff_dev_kernel(const f16* __restrict__ dy, const f16* __restrict__ a, const f16* __restrict__ b,
                                                           int64_t n, const float* __restrict__ alpha_p, float* __restrict__ out) {
    __shared__ float red[4];
    const float alpha = *alpha_p, scale = alpha * (1.0f - alpha);          // d alpha / d mix_factor
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        acc += (float)dy[i] * ((float)a[i] - (b ? (float)b[i] : 0.f));
    acc = pt_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, scale * (red[0] + red[1] + red[2] + red[3]));
}

// y = k x (one_minus = 0) or (1 - k) x (one_minus = 1), k from device memory
__global__ __launch_bounds__(256) void scale_dev_kernel(const f16* __restrict__ x, const float* __restrict__ k_p, int one_minus, int64_t n8,
                                                        f16* __restrict__ y) {
    const float k = one_minus ? 1.0f - *k_p : *k_p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f16x8 v = *(const f16x8*)(x + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)v[j] * k);
        *(f16x8*)(y + i * 8) = o;
    }
}

// out[i] = sigmoid(flat[idx[i]]): every AlphaBlender weight of the trainable network in one launch
__global__ __launch_bounds__(256) void sigmoid_gather_kernel(const float* __restrict__ flat, const int64_t* __restrict__ idx, int n,
                                                             float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 1.0f / (1.0f + expf(-flat[idx[i]]));
}

// y[r, :] = x[r, :] + vec[r / rows_per_vec, :]
__global__ __launch_bounds__(256) void add_rowvec_kernel(const f16* __restrict__ x, const f16* __restrict__ vec, int64_t rows, int C,
                                                         int64_t rows_per_vec, f16* __restrict__ y) {
    const int CH = C >> 3;
    const int64_t n8 = rows * CH;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / CH;
        const int c = (int)(i % CH) * 8;
        const f16x8 a = *(const f16x8*)(x + r * C + c), v = *(const f16x8*)(vec + (r / rows_per_vec) * C + c);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)a[j] + (float)v[j]);
        *(f16x8*)(y + r * C + c) = o;
    }
}

// backward of nearest 2x upsampling: dx[n, y, x, :] = sum of the 2 x 2 block of du [n, 2H, 2W, C]
__global__ __launch_bounds__(256) void sumpool2x_kernel(const f16* __restrict__ du, int N, int H, int W, int C, f16* __restrict__ dx) {
    const int CH = C >> 3;
    const int64_t n8 = (int64_t)N * H * W * CH;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % CH) * 8;
        int64_t p = i / CH;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H);
        const int64_t n = p / H;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
            for (int dx_ = 0; dx_ < 2; ++dx_) {
                const f16x8 v = *(const f16x8*)(du + (((n * 2 * H + 2 * yy + dy_) * 2 * W) + 2 * xx + dx_) * C + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)acc[j];
        *(f16x8*)(dx + i * 8) = o;
    }
}

// data gradient of a stride-2 convolution = stride-1 convolution (flipped taps) over the zero-interleaved output gradient:
// z[n, 2 y, 2 x, :] = dy[n, y, x, :], zero elsewhere; z is [N, H, W, C]
__global__ __launch_bounds__(256) void zero_insert2x_kernel(const f16* __restrict__ dy, int N, int OH, int OW, int H, int W, int C,
                                                            f16* __restrict__ z) {
    const int CH = C >> 3;
    const int64_t n8 = (int64_t)N * H * W * CH;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % CH) * 8;
        int64_t p = i / CH;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H);
        const int64_t n = p / H;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)0.f;
        if (!(yy & 1) && !(xx & 1) && (yy >> 1) < OH && (xx >> 1) < OW) o = *(const f16x8*)(dy + ((n * OH + (yy >> 1)) * OW + (xx >> 1)) * C + c);
        *(f16x8*)(z + i * 8) = o;
    }
}

// d loss / d pred of the EDM objective (pt_edm_loss) times `scale` (loss scale x the term's weight):
//   2 w c_out (pred c_out + c_skip noisy - target) / (F 4 HW) / B  -> fp16 channels-last [B, F, HW, 8] (channels 4..7 zero)
template <typename T>
__global__ __launch_bounds__(256) void edm_loss_bwd_kernel(const T* __restrict__ pred, int ldp, const float* __restrict__ noisy,
                                                           const float* __restrict__ target, const float* __restrict__ sigma, int B, int F,
                                                           int64_t HW, float scale, f16* __restrict__ dpred) {
    const int64_t total = (int64_t)B * F * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % HW, bf = i / HW, b = bf / F;
        const float s = sigma[b], c_out = -s / sqrtf(s * s + 1.0f), c_skip = 1.0f / (s * s + 1.0f), w = (1.0f + s * s) / (s * s);
        const float k = 2.0f * w * c_out * scale / ((float)F * 4.0f * (float)HW * (float)B);
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t e = (bf * 4 + c) * HW + p;
            const float d = (float)pred[i * ldp + c] * c_out + c_skip * noisy[e] - target[e];
            o[c] = (f16)(k * d);
            o[4 + c] = (f16)0.f;
        }
        *(f16x8*)(dpred + i * 8) = o;
    }
}

// AdamW (torch.optim.AdamW, the optimizer of scripts/train_svd_traj_VIPSeg_14.py:1051,1070-1076), fp32, in place:
//   g' = g * inv_scale ; p *= 1 - lr wd ; m = b1 m + (1 - b1) g' ; v = b2 v + (1 - b2) g'^2 ;
//   p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float inv_scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * inv_scale;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
    }
}

// The same update, four values per thread, and the two passes that follow an optimizer step in the trainer folded in (round 6): the fp16
// mirror of the parameters (ParamStore.flat16: norm weights / biases are read from it) is written here instead of by a cast over the
// whole buffer, and the gradient is zeroed here instead of by a fill - 5.4 GB of the step's 28 GB of optimizer traffic.
__global__ __launch_bounds__(256) void adamw_fused_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          int64_t n4, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                          float bc2_sqrt, float inv_scale, f16* __restrict__ mirror, int zero_g) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 g4 = *(const f32x4*)(g + 4 * i), m4 = *(const f32x4*)(m + 4 * i), v4 = *(const f32x4*)(v + 4 * i);
        f32x4 p4 = *(const f32x4*)(p + 4 * i), mo, vo;
        f16x4 h4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gi = g4[j] * inv_scale;
            float pi = p4[j] * (1.0f - lr * wd);
            const float mi = b1 * m4[j] + (1.0f - b1) * gi;
            const float vi = b2 * v4[j] + (1.0f - b2) * gi * gi;
            mo[j] = mi; vo[j] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pi -= (lr / bc1) * (mi / denom);
            p4[j] = pi;
            h4[j] = (f16)pi;
        }
        *(f32x4*)(p + 4 * i) = p4; *(f32x4*)(m + 4 * i) = mo; *(f32x4*)(v + 4 * i) = vo;
        if (mirror) *(f16x4*)(mirror + 4 * i) = h4;
        if (zero_g) *(f32x4*)(g + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// out[0] += sum g^2 (fp32 in, fp64 block sums); a non-finite gradient anywhere makes the result non-finite (the GradScaler check)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += (double)g[i] * (double)g[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, red[0]);
}


// ------------------------------------------------------------------------------------------ weight packs from the fp32 master
// w fp32 [T][Co][Ci] (the ParamStore's tap-major layout; T = kh kw taps, 1 for a linear layer) -> the fp16 image
// pt_igemm_f16 streams:
//   forward pack   dst[co][t Cpad + ci]
//   transposed     dst[ci][(T - 1 - t) Cpad + co]      (the data gradient's weight: channels swapped, taps flipped)
// A block moves a 32 (co) x 32 (ci) tile of every tap: 128-byte runs in, 64-byte runs out (through LDS when transposing).
// Padding of dst is never written (the buffers are zero-filled once).
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int Co, int Ci, int T, int transposed,
                                                          f16* __restrict__ dst, int Kpad, int Cpad, const float* __restrict__ bias,
                                                          f16* __restrict__ dst_bias) {
    __shared__ f16 tile[32][33];
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    for (int t = 0; t < T; ++t) {
        const float* src = w + (int64_t)t * Co * Ci;
        if (!transposed) {
            for (int r = r0; r < 32; r += 8)
                if (co0 + r < Co && ci0 + c < Ci) dst[(int64_t)(co0 + r) * Kpad + t * Cpad + ci0 + c] = (f16)src[(int64_t)(co0 + r) * Ci + ci0 + c];
        } else {
            __syncthreads();
            for (int r = r0; r < 32; r += 8) tile[r][c] = (co0 + r < Co && ci0 + c < Ci) ? (f16)src[(int64_t)(co0 + r) * Ci + ci0 + c] : (f16)0.f;
            __syncthreads();
            for (int q = r0; q < 32; q += 8)          // q: ci within the tile, c: co within the tile
                if (ci0 + q < Ci && co0 + c < Co) dst[(int64_t)(ci0 + q) * Kpad + (T - 1 - t) * Cpad + co0 + c] = tile[c][q];
        }
    }
    if (!transposed && blockIdx.x == 0 && bias && dst_bias && threadIdx.x < 32 && co0 + (int)threadIdx.x < Co)
        dst_bias[co0 + threadIdx.x] = (f16)bias[co0 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------ few-row linear layers
// out[m, n] = sum_k x[m, k] W[n, k] + bias[n] (+ res[m, n]) for M <= 16 rows: the time-embedding MLPs, time_emb_proj of every
// residual block, the collapsed cross-attentions and the frame position embedding (M = 1 ... frames).  One wave per output
// column: a GEMM tile would idle 15/16 of the matrix core and, split-K, cost two launches of ~30 us for 3 MB of weights.
__global__ __launch_bounds__(256) void gemv_kernel(const f16* __restrict__ x, int ldx, int M, const f16* __restrict__ W, int Kpad, int K, int N,
                                                   const f16* __restrict__ bias, const f16* __restrict__ res, int ldr, f16* __restrict__ out,
                                                   int ldo) {
    const int lane = threadIdx.x & 63;
    for (int n = blockIdx.x * 4 + (threadIdx.x >> 6); n < N; n += gridDim.x * 4) {
        float acc[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m] = 0.f;
        const f16* wr = W + (int64_t)n * Kpad;
        for (int k0 = lane * 8; k0 < K; k0 += 512) {
            // all 17 loads of a step are issued before the first use: with `if (m < M)` around each row's load the compiler
            // waited for every one of them in turn (34 s_waitcnt vmcnt(0) per step of the loop, tools/isa_wait_scan.py; VERDICT
            // r04 #4).  Rows >= M re-read row M - 1 (L1 hits) and their sums are never stored; the arithmetic per row is unchanged.
            const f16x8 wv = *(const f16x8*)(wr + k0);
            f16x8 xv[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) xv[m] = *(const f16x8*)(x + (int64_t)(m < M ? m : M - 1) * ldx + k0);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[m] += (float)wv[j] * (float)xv[m][j];
            }
        }
        const float b = bias ? (float)bias[n] : 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m < M) {
                float v = pt_wave_sum(acc[m]);
                if (lane == 0) {
                    v += b;
                    if (res) v += (float)res[(int64_t)m * ldr + n];
                    out[(int64_t)m * ldo + n] = (f16)v;
                }
            }
        }
    }
}

inline unsigned ew_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

inline int slab_rows(int64_t rows_per_sample, int64_t other_blocks) {
    // aim at ~4096 blocks in total, at least 64 rows per slab
    int64_t slabs = 4096 / (other_blocks > 0 ? other_blocks : 1);
    if (slabs < 1) slabs = 1;
    int64_t rps = (rows_per_sample + slabs - 1) / slabs;
    if (rps < 64) rps = 64;
    rps = (rps + TY - 1) / TY * TY;
    return (int)rps;
}

}  // namespace

extern "C" int pt_groupnorm_bwd(const void* x0, const void* x1, int32_t C0, int32_t C1, int32_t groups, int64_t rows_per_sample,
                                int32_t n_samples, float eps, const void* gamma, const void* beta, int32_t silu, const void* dy,
                                void* dx0, void* dx1, float* dgamma, float* dbeta, float* stat, void* stream) {
    const int Ct = C0 + C1;
    PT_CHECK(x0 && gamma && beta && dy && dx0 && stat, "pt_groupnorm_bwd: null pointer");
    PT_CHECK((((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)dy | (uintptr_t)dx0 | (uintptr_t)dx1) & 15) == 0, "pt_groupnorm_bwd: 16-byte aligned rows required");
    PT_CHECK(C0 > 0 && C0 % 8 == 0 && C1 >= 0 && C1 % 8 == 0 && (C1 == 0 || (x1 && dx1)), "pt_groupnorm_bwd: channel counts %d + %d", C0, C1);
    PT_CHECK(groups > 0 && Ct % groups == 0, "pt_groupnorm_bwd: %d channels / %d groups", Ct, groups);
    PT_CHECK(rows_per_sample > 0 && n_samples > 0 && n_samples < 65536, "pt_groupnorm_bwd: bad sizes");
    PT_CHECK((dgamma == nullptr) == (dbeta == nullptr), "pt_groupnorm_bwd: dgamma and dbeta come together");
    hipStream_t s = (hipStream_t)stream;
    const int strips = (Ct + 255) / 256;
    const int rps = slab_rows(rows_per_sample, (int64_t)strips * n_samples);
    const int slabs = (int)((rows_per_sample + rps - 1) / rps);
    if (hipMemsetAsync(stat, 0, sizeof(float) * 4 * (size_t)n_samples * groups, s) != hipSuccess) { pt_set_error("pt_groupnorm_bwd: memset failed"); return 2; }
    const dim3 grid(slabs, strips, n_samples);
    hipLaunchKernelGGL(gnb_reduce_kernel<0>, grid, dim3(256), 0, s, (const f16*)x0, (const f16*)x1, C0, C1, groups, rows_per_sample, rps, eps,
                       (const f16*)gamma, (const f16*)beta, silu, (const f16*)dy, stat, dgamma, dbeta);
    hipLaunchKernelGGL(gnb_reduce_kernel<1>, grid, dim3(256), 0, s, (const f16*)x0, (const f16*)x1, C0, C1, groups, rows_per_sample, rps, eps,
                       (const f16*)gamma, (const f16*)beta, silu, (const f16*)dy, stat, dgamma, dbeta);
    hipLaunchKernelGGL(gnb_apply_kernel, grid, dim3(256), 0, s, (const f16*)x0, (const f16*)x1, C0, C1, groups, rows_per_sample, rps, eps,
                       (const f16*)gamma, (const f16*)beta, silu, (const f16*)dy, (const float*)stat, (f16*)dx0, (f16*)dx1);
    PT_LAUNCH_CHECK("pt_groupnorm_bwd");
    return 0;
}

extern "C" int pt_layernorm_bwd(const void* x, int64_t M, int32_t Cc, const void* gamma, float eps, const void* dy, void* dx,
                                float* dgamma, float* dbeta, float* rowstat, void* stream) {
    PT_CHECK(x && gamma && dy && dx, "pt_layernorm_bwd: null pointer");
    PT_CHECK((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma) & 15) == 0, "pt_layernorm_bwd: 16-byte aligned rows required");
    PT_CHECK(M > 0 && Cc > 0 && Cc % 8 == 0 && Cc <= 512 * LN_MAXCH, "pt_layernorm_bwd: bad sizes (M %lld, C %d)", (long long)M, Cc);
    PT_CHECK((dgamma == nullptr) == (dbeta == nullptr), "pt_layernorm_bwd: dgamma and dbeta come together");
    PT_CHECK(!dgamma || rowstat, "pt_layernorm_bwd: parameter gradients need the 2 M floats of rowstat scratch");
    const int CH = Cc / 8;
    hipStream_t s = (hipStream_t)stream;
    if (CH <= 24) launch_lnb_narrow<8>((const f16*)x, M, Cc, (const f16*)gamma, eps, (const f16*)dy, (f16*)dx, dgamma, dbeta, rowstat, s);
    else if (CH <= 48) launch_lnb_narrow<16>((const f16*)x, M, Cc, (const f16*)gamma, eps, (const f16*)dy, (f16*)dx, dgamma, dbeta, rowstat, s);
    else if (CH <= 96) launch_lnb_narrow<32>((const f16*)x, M, Cc, (const f16*)gamma, eps, (const f16*)dy, (f16*)dx, dgamma, dbeta, rowstat, s);
    else {
        // the wide rows of this network are its low-resolution levels (630 / 2 520 rows of 1 280 channels): 64 rows per
        // block left 10 - 40 blocks for 256 CUs (83 us per launch); 8 ... 64 rows per block, aiming at ~512 blocks
        int rpb = (int)((M + 511) / 512);
        rpb = rpb < 8 ? 8 : (rpb > 64 ? 64 : (rpb + 3) / 4 * 4);
        const int64_t blocks = (M + rpb - 1) / rpb;
        hipLaunchKernelGGL(lnb_kernel, dim3((unsigned)blocks), dim3(256), dgamma ? sizeof(float) * 2 * Cc : 0, s, (const f16*)x, M, Cc,
                           (const f16*)gamma, eps, (const f16*)dy, (f16*)dx, dgamma, dbeta, rpb);
    }
    PT_LAUNCH_CHECK("pt_layernorm_bwd");
    return 0;
}

extern "C" int pt_colsum_f16(const void* dy, int64_t rows_per_seg, int32_t nseg, int32_t Cc, int32_t ld, float* out, void* stream) {
    PT_CHECK(dy && out, "pt_colsum_f16: null pointer");
    PT_CHECK(rows_per_seg > 0 && nseg > 0 && nseg < 65536 && Cc > 0 && ld >= Cc && ld % 8 == 0, "pt_colsum_f16: bad sizes");
    const int strips = (Cc + 255) / 256;
    const int rps = slab_rows(rows_per_seg, (int64_t)strips * nseg);
    const int slabs = (int)((rows_per_seg + rps - 1) / rps);
    hipLaunchKernelGGL(colsum_kernel, dim3(slabs, strips, nseg), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, rows_per_seg, rps, Cc, ld, out);
    PT_LAUNCH_CHECK("pt_colsum_f16");
    return 0;
}

extern "C" int pt_softmax_rows(const float* S, int64_t rows, int32_t n, int64_t ld, void* P, int64_t ldp, void* stream) {
    PT_CHECK(S && P && rows > 0 && n > 0 && ld >= n && ldp >= n, "pt_softmax_rows: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, rows, n, ld, (f16*)P, ldp);
    PT_LAUNCH_CHECK("pt_softmax_rows");
    return 0;
}

extern "C" int pt_softmax_bwd_rows(const void* P, int64_t ldp, const float* dP, int64_t ld, int64_t rows, int32_t n, void* dS, int64_t lds,
                                   void* stream) {
    PT_CHECK(P && dP && dS && rows > 0 && n > 0 && ld >= n && ldp >= n && lds >= n, "pt_softmax_bwd_rows: bad arguments");
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const f16*)P, ldp, dP, ld, rows,
                       n, (f16*)dS, lds);
    PT_LAUNCH_CHECK("pt_softmax_bwd_rows");
    return 0;
}

extern "C" int pt_geglu_f16(const void* h, int64_t M, int32_t I, void* y, void* stream) {
    PT_CHECK(h && y && M > 0 && I > 0 && I % 8 == 0, "pt_geglu_f16: bad arguments");
    hipLaunchKernelGGL(geglu_kernel, dim3(ew_blocks(M * (I / 8))), dim3(256), 0, (hipStream_t)stream, (const f16*)h, M, I, (f16*)y);
    PT_LAUNCH_CHECK("pt_geglu_f16");
    return 0;
}

extern "C" int pt_geglu_bwd(const void* h, const void* dy, int64_t M, int32_t I, void* dh, void* stream) {
    PT_CHECK(h && dy && dh && M > 0 && I > 0 && I % 8 == 0, "pt_geglu_bwd: bad arguments");
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_blocks(M * (I / 8))), dim3(256), 0, (hipStream_t)stream, (const f16*)h, (const f16*)dy, M, I, (f16*)dh);
    PT_LAUNCH_CHECK("pt_geglu_bwd");
    return 0;
}

extern "C" int pt_silu_bwd(const void* x, const void* dy, int64_t n, void* dx, void* stream) {
    PT_CHECK(x && dy && dx && n > 0, "pt_silu_bwd: bad arguments");
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (const f16*)dy, n, (f16*)dx);
    PT_LAUNCH_CHECK("pt_silu_bwd");
    return 0;
}

extern "C" int pt_lerp_f16(const void* a, const void* b, float alpha, int64_t n, void* out, void* stream) {
    PT_CHECK(a && b && out && n > 0 && n % 8 == 0, "pt_lerp_f16: bad arguments");
    hipLaunchKernelGGL(lerp_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)a, (const f16*)b, alpha, n / 8, (f16*)out);
    PT_LAUNCH_CHECK("pt_lerp_f16");
    return 0;
}

extern "C" int pt_dot_diff(const void* dy, const void* a, const void* b, int64_t n, float scale, float* out, void* stream) {
    PT_CHECK(dy && a && out && n > 0, "pt_dot_diff: bad arguments");
    unsigned blocks = ew_blocks(n);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(dot_diff_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, (const f16*)a, (const f16*)b, n, scale, out);
    PT_LAUNCH_CHECK("pt_dot_diff");
    return 0;
}

extern "C" int pt_lerp_f16_dev(const void* a, const void* b, const float* alpha_dev, int64_t n, void* out, void* stream) {
    PT_CHECK(a && b && out && alpha_dev && n > 0 && n % 8 == 0, "pt_lerp_f16_dev: bad arguments");
    hipLaunchKernelGGL(lerp_dev_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)a, (const f16*)b, alpha_dev, n / 8, (f16*)out);
    PT_LAUNCH_CHECK("pt_lerp_f16_dev");
    return 0;
}

extern "C" int pt_dot_diff_dev(const void* dy, const void* a, const void* b, int64_t n, const float* alpha_dev, float* out, void* stream) {
    PT_CHECK(dy && a && out && alpha_dev && n > 0, "pt_dot_diff_dev: bad arguments");
    unsigned blocks = ew_blocks(n);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(dot_diff_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, (const f16*)a, (const f16*)b, n, alpha_dev, out);
    PT_LAUNCH_CHECK("pt_dot_diff_dev");
    return 0;
}

extern "C" int pt_scale_f16_dev(const void* x, const float* k_dev, int32_t one_minus, int64_t n, void* y, void* stream) {
    PT_CHECK(x && y && k_dev && n > 0 && n % 8 == 0, "pt_scale_f16_dev: bad arguments");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, k_dev, one_minus, n / 8, (f16*)y);
    PT_LAUNCH_CHECK("pt_scale_f16_dev");
    return 0;
}

extern "C" int pt_sigmoid_gather_f32(const float* flat, const int64_t* idx, int32_t n, float* out, void* stream) {
    PT_CHECK(flat && idx && out && n > 0, "pt_sigmoid_gather_f32: bad arguments");
    hipLaunchKernelGGL(sigmoid_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat, idx, n, out);
    PT_LAUNCH_CHECK("pt_sigmoid_gather_f32");
    return 0;
}

extern "C" int pt_add_rowvec_f16(const void* x, const void* vec, int64_t rows, int32_t Cc, int64_t rows_per_vec, void* y, void* stream) {
    PT_CHECK(x && vec && y && rows > 0 && Cc > 0 && Cc % 8 == 0 && rows_per_vec > 0, "pt_add_rowvec_f16: bad arguments");
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(ew_blocks(rows * (Cc / 8))), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (const f16*)vec, rows, Cc,
                       rows_per_vec, (f16*)y);
    PT_LAUNCH_CHECK("pt_add_rowvec_f16");
    return 0;
}

extern "C" int pt_sumpool2x_f16(const void* du, int32_t N, int32_t H, int32_t W, int32_t Cc, void* dx, void* stream) {
    PT_CHECK(du && dx && N > 0 && H > 0 && W > 0 && Cc > 0 && Cc % 8 == 0, "pt_sumpool2x_f16: bad arguments");
    hipLaunchKernelGGL(sumpool2x_kernel, dim3(ew_blocks((int64_t)N * H * W * (Cc / 8))), dim3(256), 0, (hipStream_t)stream, (const f16*)du, N, H, W, Cc,
                       (f16*)dx);
    PT_LAUNCH_CHECK("pt_sumpool2x_f16");
    return 0;
}

extern "C" int pt_zero_insert2x_f16(const void* dy, int32_t N, int32_t OH, int32_t OW, int32_t H, int32_t W, int32_t Cc, void* z, void* stream) {
    PT_CHECK(dy && z && N > 0 && OH > 0 && OW > 0 && H > 0 && W > 0 && Cc > 0 && Cc % 8 == 0, "pt_zero_insert2x_f16: bad arguments");
    hipLaunchKernelGGL(zero_insert2x_kernel, dim3(ew_blocks((int64_t)N * H * W * (Cc / 8))), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, N, OH, OW,
                       H, W, Cc, (f16*)z);
    PT_LAUNCH_CHECK("pt_zero_insert2x_f16");
    return 0;
}

extern "C" int pt_edm_loss_bwd(const void* pred, int32_t pred_is_f32, int32_t ldp, const float* noisy, const float* target, const float* sigma,
                               int32_t B, int32_t F, int64_t HW, float scale, void* dpred, void* stream) {
    PT_CHECK(pred && noisy && target && sigma && dpred, "pt_edm_loss_bwd: null pointer");
    PT_CHECK(B > 0 && F > 0 && HW > 0 && ldp >= 4, "pt_edm_loss_bwd: bad sizes");
    const unsigned blocks = ew_blocks((int64_t)B * F * HW);
    if (pred_is_f32)
        hipLaunchKernelGGL(edm_loss_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)pred, ldp, noisy, target, sigma, B, F,
                           HW, scale, (f16*)dpred);
    else
        hipLaunchKernelGGL(edm_loss_bwd_kernel<f16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)pred, ldp, noisy, target, sigma, B, F, HW,
                           scale, (f16*)dpred);
    PT_LAUNCH_CHECK("pt_edm_loss_bwd");
    return 0;
}

extern "C" int pt_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t step, float inv_scale, void* stream) {
    PT_CHECK(p && g && m && v && n > 0 && step >= 1, "pt_adamw_f32: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       (float)bc1, (float)sqrt(bc2), inv_scale);
    PT_LAUNCH_CHECK("pt_adamw_f32");
    return 0;
}

extern "C" int pt_adamw_fused_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  int32_t step, float inv_scale, void* half_mirror, int32_t zero_grad, void* stream) {
    PT_CHECK(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "pt_adamw_fused_f32: bad arguments (n must be a multiple of 4)");
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    PT_CHECK(al16(p) && al16(g) && al16(m) && al16(v) && (!half_mirror || ((uintptr_t)half_mirror & 7) == 0), "pt_adamw_fused_f32: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_fused_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, lr, beta1, beta2, eps, weight_decay,
                       (float)bc1, (float)sqrt(bc2), inv_scale, (f16*)half_mirror, zero_grad);
    PT_LAUNCH_CHECK("pt_adamw_fused_f32");
    return 0;
}

extern "C" int pt_sumsq_f32(const float* g, int64_t n, double* out, void* stream) {
    PT_CHECK(g && out && n > 0, "pt_sumsq_f32: bad arguments");
    unsigned blocks = ew_blocks(n);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, out);
    PT_LAUNCH_CHECK("pt_sumsq_f32");
    return 0;
}

extern "C" int pt_pack_weight_f32(const float* w, int32_t Co, int32_t Ci, int32_t T, int32_t transposed, const float* bias, void* dst,
                                  int32_t Kpad, int32_t Cpad, void* dst_bias, void* stream) {
    PT_CHECK(w && dst, "pt_pack_weight_f32: null pointer");
    PT_CHECK(Co > 0 && Ci > 0 && T > 0 && T <= 9 && Kpad > 0 && Cpad > 0, "pt_pack_weight_f32: bad sizes (Co %d Ci %d T %d)", Co, Ci, T);
    PT_CHECK((transposed ? Co : Ci) <= Cpad && (T - 1) * Cpad + (transposed ? Co : Ci) <= Kpad, "pt_pack_weight_f32: the taps do not fit the row (Cpad %d, Kpad %d)", Cpad, Kpad);
    const dim3 grid((Ci + 31) / 32, (Co + 31) / 32);
    PT_CHECK(grid.y < 65536, "pt_pack_weight_f32: too many output channels");
    hipLaunchKernelGGL(pack_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, Co, Ci, T, transposed, (f16*)dst,
                       Kpad, Cpad, bias, (f16*)dst_bias);
    PT_LAUNCH_CHECK("pt_pack_weight_f32");
    return 0;
}

extern "C" int pt_gemv_f16(const void* x, int32_t ldx, int32_t M, const void* W, int32_t Kpad, int32_t K, int32_t N, const void* bias,
                           const void* res, int32_t ldr, void* out, int32_t ldo, void* stream) {
    PT_CHECK(x && W && out, "pt_gemv_f16: null pointer");
    PT_CHECK(M >= 1 && M <= 16 && K > 0 && K % 8 == 0 && K <= Kpad && N > 0 && ldx % 8 == 0 && Kpad % 8 == 0, "pt_gemv_f16: bad sizes (M %d K %d N %d)", M, K, N);
    PT_CHECK((((uintptr_t)x | (uintptr_t)W) & 15) == 0, "pt_gemv_f16: x and W must be 16-byte aligned");
    int blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemv_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, M, (const f16*)W, Kpad, K, N, (const f16*)bias,
                       (const f16*)res, ldr, (f16*)out, ldo);
    PT_LAUNCH_CHECK("pt_gemv_f16");
    return 0;
}
