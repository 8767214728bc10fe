// pt_gemm_f16 : C[b] (+)= alpha * op(A[b]) * op(B[b]) - the activation x activation products of the TRAINING step
// (scripts/train_svd_traj_VIPSeg_14.py:1414 `accelerator.backward(loss)`), which the inference path never needs because one
// operand of every product there is a pre-packed weight (pt_igemm_f16):
//   * weight gradients   dW[co, (ky, kx), ci] = sum_p dY[p, co] * X[p + (ky, kx), ci]   (K = every output pixel of the batch;
//     B is gathered on the fly with the convolution's stride / padding, one batch entry per tap; split-K with fp32 atomics
//     straight into the gradient tensor in the framework's [Co, Ci, KH, KW] layout);
//   * attention backward S = Q K^T, dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q over (frame, head) batches addressed
//     in place inside the fused QKV projection (three-level batch strides), spatial (S = h w) and temporal (S = frames).
// Both operands are addressed by element strides (one of the two strides of each must be 1), so all four transpose
// combinations are one kernel.  A tile is staged global -> registers -> LDS in its MEMORY orientation with 16-byte stores
// (prefetching the next K step's global loads under the current step's MFMAs): an operand whose unit stride runs along k
// sits as [row][k], one whose unit stride runs along the row (dY^T, X, P^T, K as the B of dS K ...) as [k][row], and its
// v_mfma_f32_16x16x32_f16 fragments come out of ds_read_b64_tr_b16 (a 4 x 16 block of halfs delivered column-major per
// 16-lane group) - no transposing scatter.  Inside a K step of 32 lane group g then holds k = 4g..4g+3 and 16+4g..16+4g+3;
// the [row][k] operand reads the same two quads (two 8-byte reads), so both fragments agree on the k order.
// Four tile shapes (waves 2 x 2, TM x TN MFMA tiles per wave).
#include "pt_common.h"

namespace {

constexpr int GK = 64, GPITCH = GK + 8;          // K step; [row][k] rows: 64 + 8 halfs

typedef _Float16 g_f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 g_hw_f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ g_f16x4 g_lds_tr16(const f16* p) {
    const g_hw_f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) g_hw_f16x4*)p);
    return __builtin_bit_cast(g_f16x4, v);
}

struct Operand {
    const f16* p;
    int64_t s_r, s_k;        // element strides along the tile's row (m or n) and along k
};

// conv gather for the B operand of a weight gradient: row k = output pixel -> the input pixel the tap (ky, kx) reads
struct Gather {
    int on, H, W, OH, OW, stride, pad_h, pad_w, ky, kx;
    int64_t ld;
};

__device__ __forceinline__ const f16* gather_row(const Operand& o, const Gather& g, int64_t k) {
    if (!g.on) return o.p + k * o.s_k;
    const unsigned kk = (unsigned)k;                 // output pixels of a batch: < 2^32 (checked by the host)
    const int ox = (int)(kk % (unsigned)g.OW);
    const unsigned t = kk / (unsigned)g.OW;
    const int oy = (int)(t % (unsigned)g.OH);
    const int64_t img = t / (unsigned)g.OH;
    const int iy = oy * g.stride + g.ky - g.pad_h, ix = ox * g.stride + g.kx - g.pad_w;
    if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) return nullptr;
    return o.p + ((img * g.H + iy) * g.W + ix) * g.ld;
}

// one group of 8 halfs of a [ROWS x GK] tile.  kcontig: 8 consecutive k of one row; else 8 consecutive rows at one k.
template <int ROWS>
__device__ __forceinline__ f16x8 load_group(const Operand& o, const Gather& g, bool kcontig, int grp, int r0, int R, int64_t k0,
                                            int64_t k_end) {
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (f16)0.f;
    if (kcontig) {
        const int r = r0 + grp / (GK / 8);
        const int64_t k = k0 + (grp % (GK / 8)) * 8;
        if (r < R && k < k_end) {
            const f16* src = o.p + (int64_t)r * o.s_r + k;
            if (k + 8 <= k_end && (((uintptr_t)src) & 15) == 0) v = *(const f16x8*)src;
            else
                for (int j = 0; j < 8; ++j) if (k + j < k_end) v[j] = src[j];
        }
    } else {
        constexpr int RG = ROWS / 8;
        const int64_t k = k0 + grp / RG;
        const int r = r0 + (grp % RG) * 8;
        if (k < k_end && r < R) {
            const f16* row = gather_row(o, g, k);
            if (row) {
                const f16* src = row + r;
                if (r + 8 <= R && (((uintptr_t)src) & 15) == 0) v = *(const f16x8*)src;
                else
                    for (int j = 0; j < 8; ++j) if (r + j < R) v[j] = src[j];
            }
        }
    }
    return v;
}

template <int ROWS>
__device__ __forceinline__ void store_group(f16* lds, bool kcontig, int grp, f16x8 v) {
    if (kcontig) {
        *(f16x8*)(lds + (grp / (GK / 8)) * GPITCH + (grp % (GK / 8)) * 8) = v;
    } else {                                            // [k][row], pitch ROWS + 8
        constexpr int RG = ROWS / 8;
        const int k = grp / RG, r = (grp % RG) * 8;
        *(f16x8*)(lds + k * (ROWS + 8) + r) = v;
    }
}

// the 16 x 32 MFMA fragment of tile rows [rbase, rbase + 16): lane (c = lane & 15, g = lane >> 4) gets row rbase + c,
// k = 4g..4g+3 | 16+4g..16+4g+3
template <int ROWS>
__device__ __forceinline__ f16x8 fragment(const f16* lds, bool kcontig, int rbase, int lane, int kk) {
    const int c = lane & 15, g = lane >> 4;
    g_f16x4 lo, hi;
    if (kcontig) {
        const f16* p = lds + (rbase + c) * GPITCH + 32 * kk + 4 * g;
        lo = *(const g_f16x4*)p;
        hi = *(const g_f16x4*)(p + 16);
    } else {
        const f16* p = lds + (32 * kk + 4 * g + (c >> 2)) * (ROWS + 8) + rbase + 4 * (c & 3);
        lo = g_lds_tr16(p);
        hi = g_lds_tr16(p + 16 * (ROWS + 8));
    }
    return f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

struct GemmK {
    Operand A, B;
    Gather g;
    void* C;
    int64_t sc_m, sc_n;
    int M, N;
    int64_t K, k_per_split;
    int splits, nb1, nb2;
    int64_t ba[3], bb[3], bc[3];
    float alpha;
    int out_mode;            // 0 fp16 store, 1 fp32 store, 2 fp32 atomic add, 3 fp32 += (single writer)
    int gKW;
};

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmK p) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int GA = BM * GK / 8, GB = BN * GK / 8;            // 16-byte groups per tile
    constexpr int NA = (GA + 255) / 256, NB = (GB + 255) / 256;
    __shared__ __attribute__((aligned(16))) f16 As[(BM * GPITCH > GK * (BM + 8)) ? BM * GPITCH : GK * (BM + 8)];
    __shared__ __attribute__((aligned(16))) f16 Bs[(BN * GPITCH > GK * (BN + 8)) ? BN * GPITCH : GK * (BN + 8)];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    int bz = blockIdx.z;
    const int split = bz % p.splits;
    bz /= p.splits;
    const int b2 = bz % p.nb2, b1 = (bz / p.nb2) % p.nb1, b0 = bz / (p.nb2 * p.nb1);
    Operand A = p.A, B = p.B;
    Gather g = p.g;
    A.p += b0 * p.ba[0] + b1 * p.ba[1] + b2 * p.ba[2];
    B.p += b0 * p.bb[0] + b1 * p.bb[1] + b2 * p.bb[2];
    if (g.on) { g.ky = b2 / p.gKW; g.kx = b2 % p.gKW; }
    const int64_t coff = b0 * p.bc[0] + b1 * p.bc[1] + b2 * p.bc[2];
    const bool akc = A.s_k == 1, bkc = B.s_k == 1 && !g.on;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int64_t k_begin = (int64_t)split * p.k_per_split;
    int64_t k_end = k_begin + p.k_per_split;
    if (k_end > p.K) k_end = p.K;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f16x8 ra[NA], rb[NB];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int grp = t + u * 256;
            if (GA % 256 == 0 || grp < GA) ra[u] = load_group<BM>(A, Gather{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, akc, grp, m0, p.M, k0, k_end);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int grp = t + u * 256;
            if (GB % 256 == 0 || grp < GB) rb[u] = load_group<BN>(B, g, bkc, grp, n0, p.N, k0, k_end);
        }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int64_t k0 = k_begin; k0 < k_end; k0 += GK) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int grp = t + u * 256;
            if (GA % 256 == 0 || grp < GA) store_group<BM>(As, akc, grp, ra[u]);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int grp = t + u * 256;
            if (GB % 256 == 0 || grp < GB) store_group<BN>(Bs, bkc, grp, rb[u]);
        }
        __syncthreads();
        if (k0 + GK < k_end) fetch(k0 + GK);
#pragma unroll
        for (int kk = 0; kk < GK / 32; ++kk) {
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = fragment<BM>(As, akc, wm * 16 * TM + i * 16, lane, kk);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = fragment<BN>(Bs, bkc, wn * 16 * TN + j * 16, lane, kk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (k_begin >= k_end && p.out_mode >= 2) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * TM + i * 16 + (lane >> 4) * 4 + r;
                const int n = n0 + wn * 16 * TN + j * 16 + (lane & 15);
                if (m < p.M && n < p.N) {
                    const float v = acc[i][j][r] * p.alpha;
                    const int64_t at = coff + (int64_t)m * p.sc_m + (int64_t)n * p.sc_n;
                    if (p.out_mode == 0) ((f16*)p.C)[at] = (f16)v;
                    else if (p.out_mode == 1) ((float*)p.C)[at] = v;
                    else if (p.out_mode == 3) ((float*)p.C)[at] += v;
                    else atomicAdd((float*)p.C + at, v);
                }
            }
}

}  // namespace

extern "C" int pt_gemm_f16(const pt_gemm_params* q, void* stream) {
    PT_CHECK(q->A && q->B && q->C, "pt_gemm_f16: null operand");
    PT_CHECK(q->M > 0 && q->N > 0 && q->K >= 0, "pt_gemm_f16: empty problem (M %d N %d K %lld)", q->M, q->N, (long long)q->K);
    PT_CHECK(q->sa_m == 1 || q->sa_k == 1, "pt_gemm_f16: A needs a unit stride along m or k (%lld, %lld)", (long long)q->sa_m, (long long)q->sa_k);
    PT_CHECK(q->sb_n == 1 || q->sb_k == 1, "pt_gemm_f16: B needs a unit stride along n or k (%lld, %lld)", (long long)q->sb_n, (long long)q->sb_k);
    PT_CHECK(q->out_mode >= 0 && q->out_mode <= 3, "pt_gemm_f16: out_mode %d", q->out_mode);
    PT_CHECK(q->nb0 >= 1 && q->nb1 >= 1 && q->nb2 >= 1, "pt_gemm_f16: batch counts must be >= 1");
    const int gather = q->g_H > 0;
    if (gather) {
        PT_CHECK(q->K < (1LL << 32), "pt_gemm_f16: gather: more than 2^32 output pixels");
        PT_CHECK(q->sb_n == 1, "pt_gemm_f16: the gathered operand is channels-last (unit stride along n)");
        PT_CHECK(q->g_KH >= 1 && q->g_KW >= 1 && q->nb2 == q->g_KH * q->g_KW, "pt_gemm_f16: gather: the innermost batch runs over the %d x %d taps", q->g_KH, q->g_KW);
        PT_CHECK(q->g_stride >= 1 && q->g_OH >= 1 && q->g_OW >= 1 && q->g_W >= 1, "pt_gemm_f16: gather geometry");
    }
    int splits = q->splits < 1 ? 1 : q->splits;
    PT_CHECK(splits == 1 || q->out_mode == 2, "pt_gemm_f16: split-K needs out_mode 2 (fp32 atomic accumulate)");
    int64_t kps = (q->K + splits - 1) / splits;
    kps = (kps + GK - 1) / GK * GK;
    if (kps < GK) kps = GK;
    splits = (int)((q->K + kps - 1) / kps);
    if (splits < 1) splits = 1;
    GemmK k;
    k.A = Operand{(const f16*)q->A, q->sa_m, q->sa_k};
    k.B = Operand{(const f16*)q->B, q->sb_n, q->sb_k};
    k.g = Gather{gather, q->g_H, q->g_W, q->g_OH, q->g_OW, q->g_stride, q->g_pad_h, q->g_pad_w, 0, 0, q->g_ld};
    k.gKW = gather ? q->g_KW : 1;
    k.C = q->C; k.sc_m = q->sc_m; k.sc_n = q->sc_n;
    k.M = q->M; k.N = q->N; k.K = q->K; k.k_per_split = kps; k.splits = splits;
    k.nb1 = q->nb1; k.nb2 = q->nb2;
    k.ba[0] = q->ba0; k.ba[1] = q->ba1; k.ba[2] = q->ba2;
    k.bb[0] = q->bb0; k.bb[1] = q->bb1; k.bb[2] = q->bb2;
    k.bc[0] = q->bc0; k.bc[1] = q->bc1; k.bc[2] = q->bc2;
    k.alpha = q->alpha; k.out_mode = q->out_mode;
    const int64_t nz = (int64_t)q->nb0 * q->nb1 * q->nb2 * splits;
    PT_CHECK(nz <= 65535 * 1024LL, "pt_gemm_f16: %lld batch entries x splits", (long long)nz);
    hipStream_t s = (hipStream_t)stream;
    int tm, tn;
    if (q->M <= 16 && q->N <= 16) { tm = 1; tn = 1; }
    else if (q->M <= 64 && q->N <= 64) { tm = 2; tn = 2; }
    else if (q->N <= 64) { tm = 4; tn = 2; }
    else { tm = 4; tn = 4; }
    const int bm = 32 * tm, bn = 32 * tn;
    // grid.z is limited to 65535: fold the excess into grid.y?  batches here (frames x heads, positions x heads) can reach
    // 10^5, so z carries min(nz, 65535)-sized slices and the launch is repeated over slices
    const dim3 grid_xy((q->N + bn - 1) / bn, (q->M + bm - 1) / bm, 1);
    const double flops = 2.0 * q->M * q->N * (double)q->K * q->nb0 * q->nb1 * q->nb2;
    pt_prof_begin(PT_PROF_GEMM, s, flops);
    PT_CHECK(nz <= 65535 || (splits == 1 && q->nb1 * (int64_t)q->nb2 <= 65535),
             "pt_gemm_f16: more than 65535 (batch x split) entries need nb1 * nb2 <= 65535 and no split-K");
    if (nz <= 65535) {
        dim3 grid(grid_xy.x, grid_xy.y, (unsigned)nz);
        if (tm == 1) hipLaunchKernelGGL((gemm_kernel<1, 1>), grid, dim3(256), 0, s, k);
        else if (tm == 2) hipLaunchKernelGGL((gemm_kernel<2, 2>), grid, dim3(256), 0, s, k);
        else if (tn == 2) hipLaunchKernelGGL((gemm_kernel<4, 2>), grid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((gemm_kernel<4, 4>), grid, dim3(256), 0, s, k);
    } else {                                                   // slices of whole outermost-batch entries
        const int64_t inner = (int64_t)q->nb1 * q->nb2;
        const int64_t per = 65535 / inner;
        for (int64_t b0 = 0; b0 < q->nb0; b0 += per) {
            const int64_t cnt = (q->nb0 - b0 < per) ? q->nb0 - b0 : per;
            GemmK ks = k;
            ks.A.p += b0 * k.ba[0]; ks.B.p += b0 * k.bb[0];
            ks.C = (q->out_mode == 0) ? (void*)((f16*)k.C + b0 * k.bc[0]) : (void*)((float*)k.C + b0 * k.bc[0]);
            dim3 grid(grid_xy.x, grid_xy.y, (unsigned)(cnt * inner));
            if (tm == 1) hipLaunchKernelGGL((gemm_kernel<1, 1>), grid, dim3(256), 0, s, ks);
            else if (tm == 2) hipLaunchKernelGGL((gemm_kernel<2, 2>), grid, dim3(256), 0, s, ks);
            else if (tn == 2) hipLaunchKernelGGL((gemm_kernel<4, 2>), grid, dim3(256), 0, s, ks);
            else hipLaunchKernelGGL((gemm_kernel<4, 4>), grid, dim3(256), 0, s, ks);
        }
    }
    pt_prof_end(PT_PROF_GEMM, s);
    PT_LAUNCH_CHECK("pt_gemm_f16");
    return 0;
}
