// GroupNorm (statistics + affine/SiLU apply) and LayerNorm for channels-last fp16 activations.  HBM-bound
// kernels: 16-byte accesses per lane, fp32 statistics.
#include "pt_common.h"

namespace {

// ------------------------------------------------------------------------------------------ GroupNorm stats
// grid (slabs, strips, samples).  A strip is 256 consecutive channels (32 chunks of 8), a slab a run of rows.
// Thread (ty, tx): chunk column tx of the strip, rows ty, ty+8, ...  -> 16 fp32 accumulators in registers, folded to
// per-channel and then per-group sums inside the block; each block writes its <= 34 group sums (the groups its strip
// touches) to part[sample][slab][strip][GN_SLOTS][2]; the apply pass folds them (gn_fold_sample): bit-reproducible, no atomics.
constexpr int GN_TX = 32, GN_TY = 8, GN_SLOTS = 36;

__global__ __launch_bounds__(256) void gn_partial_kernel(const f16* __restrict__ x0, const f16* __restrict__ x1,
                                                         int C0, int C1, int groups, int64_t rows_per_sample,
                                                         int rows_per_slab, float* __restrict__ part) {
    __shared__ float red[GN_TY][GN_TX * 8 * 2];
    __shared__ float chan[256 * 2];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int slab = blockIdx.x, strip = blockIdx.y, sample = blockIdx.z;
    const int Ctot = C0 + C1;
    const int c = strip * 256 + tx * 8;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (c < Ctot) {
        const f16* src; int ld, co;
        if (c < C0) { src = x0; ld = C0; co = c; } else { src = x1; ld = C1; co = c - C0; }
        const int64_t r0 = (int64_t)slab * rows_per_slab;
        int64_t r1 = r0 + rows_per_slab; if (r1 > rows_per_sample) r1 = rows_per_sample;
        const f16* base = src + ((int64_t)sample * rows_per_sample) * ld + co;
        int64_t r = r0 + ty;
        for (; r + 3 * GN_TY < r1; r += 4 * GN_TY) {         // four loads in flight per thread, accumulated in row order
            f16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const f16x8*)(base + (r + u * GN_TY) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)v[u][j]; s[j] += f; q[j] += f * f; }
        }
        for (; r < r1; r += GN_TY) {
            const f16x8 v = *(const f16x8*)(base + r * ld);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s[j] += f; q[j] += f * f; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[ty][(tx * 8 + j) * 2] = s[j]; red[ty][(tx * 8 + j) * 2 + 1] = q[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 256 * 2; i += 256) {
        float a = 0.f;
#pragma unroll
        for (int y = 0; y < GN_TY; ++y) a += red[y][i];
        chan[i] = a;
    }
    __syncthreads();
    // groups touched by this strip: channels [c_lo, c_hi)
    const int cg = Ctot / groups;
    const int c_lo = strip * 256, c_hi = (c_lo + 256 < Ctot) ? c_lo + 256 : Ctot;
    const int g_lo = c_lo / cg, g_hi = (c_hi - 1) / cg;
    const int ng = g_hi - g_lo + 1;
    if ((int)threadIdx.x < 2 * ng) {
        const int g = g_lo + (threadIdx.x >> 1), which = threadIdx.x & 1;
        int a0 = g * cg, a1 = a0 + cg;
        if (a0 < c_lo) a0 = c_lo;
        if (a1 > c_hi) a1 = c_hi;
        float acc = 0.f;
        for (int ch = a0; ch < a1; ++ch) acc += chan[(ch - c_lo) * 2 + which];
        part[((((int64_t)sample * gridDim.x + slab) * gridDim.y + strip) * GN_SLOTS + (g - g_lo)) * 2 + which] = acc;
    }
}

// per sample: fold the per-block group partials in a fixed order -> mean / rstd of every group in LDS (grp[2 g], [2 g + 1]).
// 8 threads per group (256 threads = 32 groups at once), each summing every 8th slab, then a fixed-order fold over the 8:
// all the loads of a thread are independent and in flight together.  Every block of gn_apply_kernel runs this for its
// sample (same order, same bits in every block) instead of a finalize launch of its own (28 blocks, latency-bound:
// 11-15 us per GroupNorm, 1.7 ms per denoise iteration), and instead of the in-launch fold by the last-arriving block
// of round 2's first form (agent-scope ticket + release per block: 45-62 us per GroupNorm against 19 us for the bare
// partial pass, profiles/r02/rocprofv3_kernel_stats_L_2iters_r02a.csv / _r02b.csv).
__device__ __forceinline__ void gn_fold_sample(const float* __restrict__ part, int sample, int nslabs, int nstrips,
                                               int Ctot, int groups, int64_t rows_per_sample, float eps, float* grp) {
    const int cg = Ctot / groups;
    const int sub = threadIdx.x & 7;
    for (int g = threadIdx.x >> 3; g < groups; g += blockDim.x >> 3) {
        float s = 0.f, q = 0.f;
        const int ch0 = g * cg, ch1 = ch0 + cg - 1;
        for (int strip = ch0 / 256; strip <= ch1 / 256; ++strip) {       // the 1-2 strips this group lives in
            const int g_lo = (strip * 256) / cg;
            const float* base = part + ((int64_t)sample * nslabs * nstrips + strip) * GN_SLOTS * 2 + (g - g_lo) * 2;
            const int64_t step = (int64_t)nstrips * GN_SLOTS * 2;
            int sl = sub;
            for (; sl + 24 < nslabs; sl += 32) {              // four independent 8-byte loads in flight, summed in slab order
                const f32x2 e0 = *(const f32x2*)(base + sl * step), e1 = *(const f32x2*)(base + (sl + 8) * step);
                const f32x2 e2 = *(const f32x2*)(base + (sl + 16) * step), e3 = *(const f32x2*)(base + (sl + 24) * step);
                s += e0[0]; q += e0[1]; s += e1[0]; q += e1[1]; s += e2[0]; q += e2[1]; s += e3[0]; q += e3[1];
            }
            for (; sl < nslabs; sl += 8) {
                const float* e = base + sl * step;
                s += e[0]; q += e[1];
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        if (sub == 0) {
            const double cnt = (double)rows_per_sample * cg;
            const double mean = (double)s / cnt;
            double var = (double)q / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            grp[2 * g] = (float)mean;
            grp[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
}

// ------------------------------------------------------------------------------------------ GroupNorm apply
// grid (row blocks, samples): a block owns rows [r0, r1) of ONE sample, so the (a, b) pairs of the sample sit in LDS and
// the chunk index advances without any division (the first version divided two 64-bit indices per 16-byte chunk and was
// bound by that arithmetic, not by HBM: 2.8-4.4 TB/s).  Thread t starts at chunk t of the block's [rows x C/8] range and
// steps by 256 chunks: (row, column) advance by the constant (256 / CH, 256 % CH) with one carry.
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x0, const f16* __restrict__ x1, int C0,
                                                       int C1, int rows_per_sample, int rows_per_block,
                                                       const float* __restrict__ part, int nslabs, int nstrips, int groups,
                                                       float eps, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                       int silu, f16* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float s_ab[];          // [Ctot][2]: a = rstd*gamma, b = beta - mean*a
    __shared__ float grp[2 * 64];
    const int Ctot = C0 + C1, CH = Ctot >> 3;
    const int sample = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, rows_per_sample);
    gn_fold_sample(part, sample, nslabs, nstrips, Ctot, groups, rows_per_sample, eps, grp);
    __syncthreads();
    {
        const int cg = Ctot / groups;
        for (int c = threadIdx.x; c < Ctot; c += 256) {
            const int g = c / cg;
            const float a = grp[2 * g + 1] * (float)gamma[c];
            s_ab[2 * c] = a;
            s_ab[2 * c + 1] = (float)beta[c] - grp[2 * g] * a;
        }
    }
    __syncthreads();
    const int64_t base_row = (int64_t)sample * rows_per_sample;
    const int dr = 256 / CH, dc = 256 - dr * CH;
    int row = r0 + (int)threadIdx.x / CH, c = (int)threadIdx.x % CH;
    auto advance = [&](int& rw, int& cc) {
        rw += dr; cc += dc;
        if (cc >= CH) { cc -= CH; ++rw; }
    };
    auto load = [&](int rw, int cc) -> f16x8 {
        const int ch = cc * 8;
        const int64_t grow = base_row + rw;
        return ch < C0 ? *(const f16x8*)(x0 + grow * C0 + ch) : *(const f16x8*)(x1 + grow * C1 + (ch - C0));
    };
    auto finish = [&](int rw, int cc, const f16x8& v) {
        const int ch = cc * 8;
        const f32x4* abp = (const f32x4*)(s_ab + ch * 2);
        f16x8 o;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f32x4 k = abp[h];                       // a0 b0 a1 b1
            float u0 = (float)v[2 * h] * k[0] + k[1], u1 = (float)v[2 * h + 1] * k[2] + k[3];
            if (silu) { u0 = pt_silu(u0); u1 = pt_silu(u1); }
            o[2 * h] = (f16)u0; o[2 * h + 1] = (f16)u1;
        }
        *(f16x8*)(y + (base_row + rw) * Ctot + ch) = o;
    };
    // four chunks per trip, every load issued before the first use: one 16-byte load in flight per thread left the pass
    // latency-bound wherever few waves share a CU (a level-3 tensor took 25 us for 20 MB: 48 dependent round trips per thread)
    for (;;) {
        int rw[4], cc[4];
        rw[0] = row; cc[0] = c;
#pragma unroll
        for (int u = 1; u < 4; ++u) { rw[u] = rw[u - 1]; cc[u] = cc[u - 1]; advance(rw[u], cc[u]); }
        if (rw[3] >= r1) break;
        f16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load(rw[u], cc[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(rw[u], cc[u], v[u]);
        row = rw[3]; c = cc[3];
        advance(row, c);
    }
    for (; row < r1; advance(row, c)) finish(row, c, load(row, c));
}

// ------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; C <= 8 * 64 * 4 = 2048.  Two passes over registers (mean, then centred variance).
template <int NCH>   // chunks of 8 per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x, int64_t M, int C,
                                                        const f16* __restrict__ vec, int ldv, int vec_mode, int vG,
                                                        const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                        float eps, f16* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int CH = C >> 3;
    float v[NCH][8];
    float sum = 0.f;
    const f16* vrow = vec_mode ? vec + (int64_t)(row / vG) * ldv : nullptr;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < CH) {
            const f16x8 a = *(const f16x8*)(x + row * C + ch * 8);
            if (vrow) {
                const f16x8 b = *(const f16x8*)(vrow + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[k][j] = (float)(f16)((float)a[j] + (float)b[j]);   // fp16 add like the reference
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[k][j] = (float)a[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[k][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[k][j] = 0.f;
        }
    }
    const float mean = pt_wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
        if (lane + 64 * k < CH) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[k][j] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(pt_wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < CH) {
            const f16x8 g = *(const f16x8*)(gamma + ch * 8), b = *(const f16x8*)(beta + ch * 8);
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)((v[k][j] - mean) * rstd * (float)g[j] + (float)b[j]);
            *(f16x8*)(y + row * C + ch * 8) = o;
        }
    }
}

// The SVD widths (C = 320 / 640 / 1280: 40 / 80 / 160 chunks, which fill a wave's 64 lanes badly one row at a time):
// LPR lanes per row, 64 / LPR rows per wave, each lane owns the CPL chunks l, l + LPR, ..., so every load instruction
// covers whole 128-byte lines and all 64 lanes have CPL loads in flight.  C = LPR * CPL * 8.
template <int LPR, int CPL>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const f16* __restrict__ x, int64_t M, int C,
                                                               const f16* __restrict__ vec, int ldv, int vec_mode, int vG,
                                                               const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                               float eps, f16* __restrict__ y) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane & (LPR - 1);
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool ok = row < M;
    const int64_t r = ok ? row : M - 1;
    float v[CPL][8];
    float sum = 0.f;
    const f16* vrow = vec_mode ? vec + (int64_t)(r / vG) * ldv : nullptr;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int ch = l + LPR * k;
        const f16x8 a = *(const f16x8*)(x + r * C + ch * 8);
        if (vrow) {
            const f16x8 b = *(const f16x8*)(vrow + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[k][j] = (float)(f16)((float)a[j] + (float)b[j]);   // fp16 add like the reference
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[k][j] = (float)a[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[k][j];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[k][j] - mean; sq += d * d; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq / (float)C + eps);
    if (!ok) return;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int ch = l + LPR * k;
        const f16x8 g = *(const f16x8*)(gamma + ch * 8), b = *(const f16x8*)(beta + ch * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((v[k][j] - mean) * rstd * (float)g[j] + (float)b[j]);
        *(f16x8*)(y + row * C + ch * 8) = o;
    }
}

int slab_rows(int64_t rows_per_sample, int nstrips, int n_samples) {
    // aim for ~2048 blocks, slabs of at least 64 rows and at most 1024 slabs per sample
    int64_t target_slabs = 2048 / ((int64_t)nstrips * n_samples);
    if (target_slabs < 1) target_slabs = 1;
    if (target_slabs > 1024) target_slabs = 1024;
    int64_t rows = (rows_per_sample + target_slabs - 1) / target_slabs;
    if (rows < 64) rows = 64;
    rows = (rows + 7) / 8 * 8;
    return (int)rows;
}

}  // namespace

extern "C" int64_t pt_groupnorm_scratch_floats(int64_t rows_total, int32_t C, int32_t n_samples) {
    const int64_t rps = rows_total / (n_samples > 0 ? n_samples : 1);
    const int nstrips = (C + 255) / 256;
    const int rows = slab_rows(rps, nstrips, n_samples);
    const int64_t nslabs = (rps + rows - 1) / rows;
    return (int64_t)n_samples * nslabs * nstrips * GN_SLOTS * 2;
}

extern "C" int pt_groupnorm_stats(const void* x0, const void* x1, int32_t C0, int32_t C1, int32_t groups,
                                  int64_t rows_per_sample, int32_t n_samples, float* partials, void* stream) {
    const int Ctot = C0 + C1;
    PT_CHECK(x0 && partials, "pt_groupnorm_stats: null pointer");
    PT_CHECK(C0 % 8 == 0 && C1 % 8 == 0 && groups > 0 && groups <= 32 && Ctot % groups == 0 && Ctot / groups >= 2,
             "pt_groupnorm_stats: C0=%d C1=%d groups=%d", C0, C1, groups);
    PT_CHECK((C1 == 0) == (x1 == nullptr), "pt_groupnorm_stats: x1/C1 mismatch");
    PT_CHECK(rows_per_sample > 0 && n_samples > 0, "pt_groupnorm_stats: empty input");
    const int nstrips = (Ctot + 255) / 256;
    const int rows = slab_rows(rows_per_sample, nstrips, n_samples);
    const int nslabs = (int)((rows_per_sample + rows - 1) / rows);
    PT_CHECK(n_samples <= 65535 && nstrips <= 65535, "pt_groupnorm_stats: grid too large");
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nslabs, nstrips, n_samples), dim3(256), 0, (hipStream_t)stream, (const f16*)x0,
                       (const f16*)x1, C0, C1, groups, rows_per_sample, rows, partials);
    PT_LAUNCH_CHECK("pt_groupnorm_stats");
    return 0;
}

extern "C" int pt_groupnorm_apply(const void* x0, const void* x1, int32_t C0, int32_t C1, int32_t groups,
                                  int64_t rows_per_sample, int32_t n_samples, float eps, const void* gamma,
                                  const void* beta, const float* partials, int32_t silu, void* y, void* stream) {
    PT_CHECK(x0 && gamma && beta && partials && y, "pt_groupnorm_apply: null pointer");
    PT_CHECK(C0 % 8 == 0 && C1 % 8 == 0, "pt_groupnorm_apply: channels must be multiples of 8");
    const int Ctot = C0 + C1;
    PT_CHECK(Ctot >= 8 && Ctot <= 4096, "pt_groupnorm_apply: %d channels unsupported (8 .. 4096)", Ctot);
    PT_CHECK(groups > 0 && groups <= 32 && Ctot % groups == 0 && Ctot / groups >= 2, "pt_groupnorm_apply: C=%d groups=%d", Ctot, groups);
    PT_CHECK(rows_per_sample > 0 && rows_per_sample < (1ll << 31) && n_samples > 0 && n_samples <= 65535, "pt_groupnorm_apply: bad sizes");
    // the geometry pt_groupnorm_stats used for `partials`
    const int nstrips = (Ctot + 255) / 256;
    const int srows = slab_rows(rows_per_sample, nstrips, n_samples);
    const int nslabs = (int)((rows_per_sample + srows - 1) / srows);
    // ~4096 blocks over the launch, at least 16 rows (>= 32 chunks per thread at C = 320 .. 1280 keeps the LDS fill cheap)
    int64_t per_sample = 4096 / n_samples;
    if (per_sample < 1) per_sample = 1;
    int64_t rows_per_block = (rows_per_sample + per_sample - 1) / per_sample;
    // small tensors (level 2 / 3: a few MB) would otherwise run on a fraction of the CUs: 8 chunks per thread are enough there
    const int64_t total_chunks = rows_per_sample * n_samples * (Ctot >> 3);
    const int min_chunks = total_chunks < (int64_t)256 * 32 * 1024 ? 8 : 32;
    const int64_t min_rows = (256 * min_chunks + (Ctot >> 3) - 1) / (Ctot >> 3);
    if (rows_per_block < min_rows) rows_per_block = min_rows;
    const unsigned bx = (unsigned)((rows_per_sample + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(bx, (unsigned)n_samples), dim3(256), (size_t)Ctot * 8, (hipStream_t)stream,
                       (const f16*)x0, (const f16*)x1, C0, C1, (int)rows_per_sample, (int)rows_per_block, partials, nslabs,
                       nstrips, groups, eps, (const f16*)gamma, (const f16*)beta, silu, (f16*)y);
    PT_LAUNCH_CHECK("pt_groupnorm_apply");
    return 0;
}

extern "C" int pt_layernorm_f16(const void* x, int64_t M, int32_t C, const void* vec, int32_t ldv, int32_t vec_mode,
                                int32_t vG, const void* gamma, const void* beta, float eps, void* y, void* stream) {
    PT_CHECK(x && gamma && beta && y, "pt_layernorm_f16: null pointer");
    PT_CHECK(C % 8 == 0 && C <= 2048, "pt_layernorm_f16: C=%d must be a multiple of 8 and <= 2048", C);
    PT_CHECK(vec_mode == 0 || (vec_mode == 1 && vec && vG > 0 && ldv % 8 == 0), "pt_layernorm_f16: bad vec arguments");
    hipStream_t s = (hipStream_t)stream;
#define LN_NARROW(LPR, CPL)                                                                                            \
    do {                                                                                                               \
        const int64_t rpb = 4 * (64 / LPR);                                                                            \
        hipLaunchKernelGGL((layernorm_narrow_kernel<LPR, CPL>), dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, s, \
                           (const f16*)x, M, C, (const f16*)vec, ldv, vec_mode, vG, (const f16*)gamma, (const f16*)beta, \
                           eps, (f16*)y);                                                                              \
        PT_LAUNCH_CHECK("pt_layernorm_f16");                                                                           \
        return 0;                                                                                                      \
    } while (0)
    if (C == 320) LN_NARROW(8, 5);                           // the SVD widths
    if (C == 640) LN_NARROW(16, 5);
    if (C == 1280) LN_NARROW(32, 5);
#undef LN_NARROW
    const unsigned blocks = (unsigned)((M + 3) / 4);
    const int nch = ((C >> 3) + 63) / 64;
#define LN_LAUNCH(NCH)                                                                                              \
    hipLaunchKernelGGL(layernorm_kernel<NCH>, dim3(blocks), dim3(256), 0, s, (const f16*)x, M, C, (const f16*)vec, \
                       ldv, vec_mode, vG, (const f16*)gamma, (const f16*)beta, eps, (f16*)y)
    if (nch == 1) LN_LAUNCH(1); else if (nch == 2) LN_LAUNCH(2); else if (nch == 3) LN_LAUNCH(3); else LN_LAUNCH(4);
#undef LN_LAUNCH
    PT_LAUNCH_CHECK("pt_layernorm_f16");
    return 0;
}
