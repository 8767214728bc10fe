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
// (the next K step's global loads are issued right after the barrier and land under the current step's MFMAs; the loader -
// `Stager` - is branch-free in the regular case so that nothing waits for them before the next LDS store): an operand whose unit stride runs along k
// sits as [row][k], one whose unit stride runs along the row (dY^T, X, P^T, K as the B of dS K ...) as [k][row], and its
// v_mfma_f32_16x16x32_f16 fragments come out of ds_read_b64_tr_b16 (a 4 x 16 block of halfs delivered column-major per
// 16-lane group) - no transposing scatter.  Inside a K step of 32 lane group g then holds k = 4g..4g+3 and 16+4g..16+4g+3;
// the [row][k] operand reads the same two quads (two 8-byte reads), so both fragments agree on the k order.
// Four tile shapes (waves 2 x 2, TM x TN MFMA tiles per wave), a one-dimensional XCD-aware grid (see the kernel).
#include "pt_common.h"

namespace {

constexpr int GK = 64, GPITCH = GK + 8;          // K step; [row][k] rows: 64 + 8 halfs
constexpr int TRPAD = 16;                        // [k][row] rows: ROWS + 16 halfs - the 8 k-rows a 32-lane half of ds_read_b64_tr_b16
                                                 // touches start 8 banks apart (pitch / 2 = 8 mod 64 dwords for ROWS = 32 ... 256)

typedef _Float16 g_f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 g_hw_f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ g_f16x4 g_lds_tr16(const f16* p) {
    const g_hw_f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) g_hw_f16x4*)p);
    return __builtin_bit_cast(g_f16x4, v);
}

__device__ __attribute__((aligned(16))) unsigned g_gemm_zeros[4];      // 16 zero bytes: the source of every out-of-range group

struct Operand {
    const f16* p;
    int64_t s_r, s_k;        // element strides along the tile's row (m or n) and along k
};

// conv gather for the B operand of a weight gradient: row k = output pixel -> the input pixel the tap (ky, kx) reads
struct Gather {
    int on, H, W, OH, OW, stride, pad_h, pad_w, ky, kx;
    int64_t ld;
    int dimg, doy, dox;      // GK output pixels further, in (images, rows, columns)
};

// This thread's ROWS / 32 sixteen-byte groups (8 halfs each) of one operand's [ROWS x GK] tile, K step after K step.
// Group u is group (t + 256 u) of the tile: kcontig (8 consecutive k of one row) -> row0 + 32 u at k offset `sub`; else
// (8 consecutive rows at one k; the gathered operand is always of this form) -> rows row0.. at k offset sub + KU u.
// What does not change along k is worked out once, and what is the same for all of a thread's groups is kept once: a K
// step costs one pointer increment per operand, a compare and the load per group.  Two earlier forms of this loader are why:
//  * re-deriving row, column, the gather's pixel decomposition (two divisions) and 64-bit products per group per step, and
//  * alignment / tail branches around each load, behind which the compiler placed `s_waitcnt vmcnt(0)` - eight serialised
//    round trips to L2 per K step and nothing overlapping the MFMAs -
// held every shape of the training step at ~300 TFLOP/s whatever the tile, the LDS layout, the tile order or the prefetch
// depth (tools/micro/gemm_time.py).  FAST (every group whole and 16-byte aligned: the host checks strides, base pointers and
// M / N / K divisibility) is ONE unconditional load per group from a select-ed address - a zero block for what lies outside -
// and no arithmetic on the loaded value before it is stored to LDS.
template <int ROWS, bool FAST>
struct Stager {
    static constexpr int RG = ROWS / 8, KU = 256 / RG, NG = ROWS / 32;
    const f16* p;                      // group 0's source at the K step to be fetched next (not used by the gather)
    int sub, row;                      // group 0's k offset inside a K step; its first row
    int ox[NG], oy[NG], img[NG];       // gather: the output pixel of each group's k

    __device__ __forceinline__ void start(const Operand& o, const Gather& g, bool kc, int t, int r0, int64_t k_begin) {
        if (kc) {
            sub = (t & 7) * 8; row = r0 + (t >> 3);
            p = o.p + (int64_t)row * o.s_r + k_begin + sub;
        } else {
            sub = t / RG; row = r0 + (t % RG) * 8;
            p = o.p + (k_begin + sub) * o.s_k + row;
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            ox[u] = oy[u] = img[u] = 0;
            if (g.on) {
                const unsigned kk = (unsigned)(k_begin + sub + KU * u);      // output pixels of a batch: < 2^32 (checked by the host)
                ox[u] = (int)(kk % (unsigned)g.OW);
                const unsigned q = kk / (unsigned)g.OW;
                oy[u] = (int)(q % (unsigned)g.OH);
                img[u] = (int)(q / (unsigned)g.OH);
            }
        }
    }

    // group u of the K step that starts at k0
    __device__ __forceinline__ f16x8 load(int u, const Operand& o, const Gather& g, bool kc, int R, int64_t k0, int64_t k_end) const {
        const int r = kc ? row + 32 * u : row;
        const int64_t k = k0 + sub + (kc ? 0 : KU * u);
        bool ok = r < R && k < k_end;
        const f16* src = p + (kc ? (int64_t)(32 * u) * o.s_r : (int64_t)(KU * u) * o.s_k);
        if (g.on) {
            const int iy = oy[u] * g.stride + g.ky - g.pad_h, ix = ox[u] * g.stride + g.kx - g.pad_w;
            ok = ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
            src = o.p + (((int64_t)img[u] * g.H + iy) * g.W + ix) * g.ld + row;
        }
        if (FAST) {
            src = ok ? src : (const f16*)g_gemm_zeros;
            return *(const f16x8*)src;
        }
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f16)0.f;
        if (!ok) return v;
        const int64_t left = kc ? k_end - k : (int64_t)(R - r);               // of the 8 elements, how many exist
        if (left >= 8 && (((uintptr_t)src) & 15) == 0) v = *(const f16x8*)src;
        else
            for (int j = 0; j < 8; ++j) if (j < left) v[j] = src[j];
        return v;
    }

    // to the next K step: GK further along k; the gather's pixels move by (g.dimg, g.doy, g.dox), one carry per digit
    __device__ __forceinline__ void next(const Operand& o, const Gather& g, bool kc) {
        p += kc ? (int64_t)GK : GK * o.s_k;
        if (g.on) {
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                ox[u] += g.dox;
                const int cx = ox[u] >= g.OW;
                ox[u] -= cx ? g.OW : 0;
                oy[u] += g.doy + cx;
                const int cy = oy[u] >= g.OH;
                oy[u] -= cy ? g.OH : 0;
                img[u] += g.dimg + cy;
            }
        }
    }
};

template <int ROWS>
__device__ __forceinline__ void store_group(f16* lds, bool kcontig, int grp, f16x8 v) {
    if (kcontig) {
        *(f16x8*)(lds + (grp / (GK / 8)) * GPITCH + (grp % (GK / 8)) * 8) = v;
    } else {                                            // [k][row], pitch ROWS + TRPAD
        constexpr int RG = ROWS / 8;
        const int k = grp / RG, r = (grp % RG) * 8;
        *(f16x8*)(lds + k * (ROWS + TRPAD) + r) = v;
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
        const f16* p = lds + (32 * kk + 4 * g + (c >> 2)) * (ROWS + TRPAD) + rbase + 4 * (c & 3);
        lo = g_lds_tr16(p);
        hi = g_lds_tr16(p + 16 * (ROWS + TRPAD));
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
    int splits, nb1, nb2, tiles_m, tiles_n;
    int64_t ba[3], bb[3], bc[3];
    float alpha;
    int out_mode;            // 0 fp16 store, 1 fp32 store, 2 fp32 atomic add, 3 fp32 += (single writer)
    int gKW;
};

template <int TM, int TN, bool FAST>
__global__ __launch_bounds__(256) void gemm_kernel(GemmK p) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int DEPTH = 1;     // register sets of loads in flight; 2 was measured: -5 % at the same occupancy (252 VGPRs), -45 % at half of it
    constexpr int GA = BM * GK / 8, GB = BN * GK / 8;            // 16-byte groups per tile
    constexpr int NA = (GA + 255) / 256, NB = (GB + 255) / 256;
    __shared__ __attribute__((aligned(16))) f16 As[(BM * GPITCH > GK * (BM + TRPAD)) ? BM * GPITCH : GK * (BM + TRPAD)];
    __shared__ __attribute__((aligned(16))) f16 Bs[(BN * GPITCH > GK * (BN + TRPAD)) ? BN * GPITCH : GK * (BN + TRPAD)];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    // Tile order: the grid is one-dimensional; pt_xcd_remap hands every XCD a contiguous run of (batch entry, split, tile) ids,
    // and inside a batch entry GROUP_M consecutive M tiles share an N tile before the next N tile starts, so the ~64 workgroups
    // resident on an XCD form an 8 x 8 block of one product's tile grid and share both operand panels through that L2.  With
    // the round-robin order (x = N tiles fastest, neighbours on different XCDs) each XCD streamed nearly the whole of both
    // operands per K step: every shape of the training step sat at 32 KB per workgroup per K step x 512 workgroups
    // = 4.7 TB/s of beyond-L2 traffic, ~300 TFLOP/s whatever else changed (tools/micro/gemm_time.py).
    const int lin = pt_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    int bz = lin / tiles;
    const int within = lin - bz * tiles;
    constexpr int GROUP_M = 8;
    const int gsz = GROUP_M * p.tiles_n, grp_i = within / gsz, first_m = grp_i * GROUP_M;
    const int gmm = min(GROUP_M, p.tiles_m - first_m), w_in = within - grp_i * gsz;
    const int tile_m = first_m + w_in % gmm, tile_n = w_in / gmm;
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
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int64_t k_begin = (int64_t)split * p.k_per_split;
    int64_t k_end = k_begin + p.k_per_split;
    if (k_end > p.K) k_end = p.K;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const Gather nog{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    static_assert(GA % 256 == 0 && GB % 256 == 0 && NA == BM / 32 && NB == BN / 32, "whole groups per thread");
    Stager<BM, FAST> sa;
    Stager<BN, FAST> sb;
    sa.start(A, nog, akc, t, m0, k_begin);
    sb.start(B, g, bkc, t, n0, k_begin);
    int64_t kf = k_begin;                                      // the K step the next fetch loads
    // DEPTH K steps of global loads are in flight (one register set each) while one is computed
    f16x8 ra[DEPTH][NA], rb[DEPTH][NB];
    auto fetch = [&](f16x8 (&xa)[NA], f16x8 (&xb)[NB]) {
#pragma unroll
        for (int u = 0; u < NA; ++u) xa[u] = sa.load(u, A, nog, akc, p.M, kf, k_end);
#pragma unroll
        for (int u = 0; u < NB; ++u) xb[u] = sb.load(u, B, g, bkc, p.N, kf, k_end);
        sa.next(A, nog, akc);
        sb.next(B, g, bkc);
        kf += GK;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (k_begin + d * GK < k_end) fetch(ra[d], rb[d]);
    for (int64_t kb = k_begin; kb < k_end; kb += DEPTH * GK) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int64_t k0 = kb + d * GK;
            if (k0 >= k_end) break;
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int grp = t + u * 256;
                if (GA % 256 == 0 || grp < GA) store_group<BM>(As, akc, grp, ra[d][u]);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int grp = t + u * 256;
                if (GB % 256 == 0 || grp < GB) store_group<BN>(Bs, bkc, grp, rb[d][u]);
            }
            __syncthreads();
            if (k0 + DEPTH * GK < k_end) fetch(ra[d], rb[d]);
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
    }
    if (k_begin >= k_end && p.out_mode >= 2) return;
    if (p.out_mode == 3) {
        // `+=` by a single writer: the old values of one MFMA row block (TN x 4 per lane) are all loaded before the first
        // store - written as load / add / store per element the compiler must assume each store aliases the next load and
        // waits for every one of TM x TN x 4 round trips in turn (64 per thread for the 128 x 128 tile)
        float* Cf = (float*)p.C;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float old[TN][4];
            int64_t at[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 16 * TM + i * 16 + (lane >> 4) * 4 + r;
                    const int n = n0 + wn * 16 * TN + j * 16 + (lane & 15);
                    const bool ok = m < p.M && n < p.N;
                    at[j][r] = ok ? coff + (int64_t)m * p.sc_m + (int64_t)n * p.sc_n : -1;
                    old[j][r] = Cf[ok ? at[j][r] : coff];                 // coff: this batch entry's element (0, 0), always there
                }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (at[j][r] >= 0) Cf[at[j][r]] = old[j][r] + acc[i][j][r] * p.alpha;
        }
        return;
    }
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
    k.g = Gather{gather, q->g_H, q->g_W, q->g_OH, q->g_OW, q->g_stride, q->g_pad_h, q->g_pad_w, 0, 0, q->g_ld,
                 gather ? (GK / q->g_OW) / q->g_OH : 0, gather ? (GK / q->g_OW) % q->g_OH : 0, gather ? GK % q->g_OW : 0};
    k.gKW = gather ? q->g_KW : 1;
    k.C = q->C; k.sc_m = q->sc_m; k.sc_n = q->sc_n;
    k.M = q->M; k.N = q->N; k.K = q->K; k.k_per_split = kps; k.splits = splits;
    k.nb1 = q->nb1; k.nb2 = q->nb2;
    k.ba[0] = q->ba0; k.ba[1] = q->ba1; k.ba[2] = q->ba2;
    k.bb[0] = q->bb0; k.bb[1] = q->bb1; k.bb[2] = q->bb2;
    k.bc[0] = q->bc0; k.bc[1] = q->bc1; k.bc[2] = q->bc2;
    k.alpha = q->alpha; k.out_mode = q->out_mode;
    const int64_t nz = (int64_t)q->nb0 * q->nb1 * q->nb2 * splits;
    hipStream_t s = (hipStream_t)stream;
    int tm, tn;
    if (q->M <= 16 && q->N <= 16) { tm = 1; tn = 1; }
    else if (q->M <= 64 && q->N <= 64) { tm = 2; tn = 2; }
    else if (q->N <= 64) { tm = 4; tn = 2; }
    else { tm = 4; tn = 4; }
    const int bm = 32 * tm, bn = 32 * tn;
    k.tiles_m = (q->M + bm - 1) / bm; k.tiles_n = (q->N + bn - 1) / bn;
    const int64_t nwg = (int64_t)k.tiles_m * k.tiles_n * nz;
    PT_CHECK(nwg < (1LL << 31), "pt_gemm_f16: %lld workgroups (tiles x batch entries x splits)", (long long)nwg);
    const double flops = 2.0 * q->M * q->N * (double)q->K * q->nb0 * q->nb1 * q->nb2;
    pt_prof_begin(PT_PROF_GEMM, s, flops);
    const dim3 grid((unsigned)nwg, 1, 1);
    // the regular case: every 16-byte group of both operands is whole and aligned
    auto regular = [](const void* ptr, int64_t s_r, int64_t s_k, int rows, int64_t K, int64_t b0, int64_t b1, int64_t b2, int64_t ld) {
        const bool kc = s_k == 1 && ld == 0;
        const int64_t other = kc ? s_r : (ld ? ld : s_k);
        return ((uintptr_t)ptr & 15) == 0 && other % 8 == 0 && b0 % 8 == 0 && b1 % 8 == 0 && b2 % 8 == 0 && (kc ? K % 8 == 0 : rows % 8 == 0);
    };
    const bool fast = regular(q->A, q->sa_m, q->sa_k, q->M, q->K, q->ba0, q->ba1, q->ba2, 0) &&
                      regular(q->B, q->sb_n, q->sb_k, q->N, q->K, q->bb0, q->bb1, q->bb2, gather ? q->g_ld : 0);
#define PT_GEMM_LAUNCH(TM_, TN_)                                                                              \
    do {                                                                                                      \
        if (fast) hipLaunchKernelGGL((gemm_kernel<TM_, TN_, true>), grid, dim3(256), 0, s, k);                \
        else hipLaunchKernelGGL((gemm_kernel<TM_, TN_, false>), grid, dim3(256), 0, s, k);                    \
    } while (0)
    if (tm == 1) PT_GEMM_LAUNCH(1, 1);
    else if (tm == 2) PT_GEMM_LAUNCH(2, 2);
    else if (tn == 2) PT_GEMM_LAUNCH(4, 2);
    else PT_GEMM_LAUNCH(4, 4);
#undef PT_GEMM_LAUNCH
    pt_prof_end(PT_PROF_GEMM, s);
    PT_LAUNCH_CHECK("pt_gemm_f16");
    return 0;
}
