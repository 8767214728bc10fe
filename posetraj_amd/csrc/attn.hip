// Attention kernels of the spatio-temporal transformer, gfx950.
//
// pt_attn_spatial_f16 : flash-style softmax(QK^T)V per (image, head), head_dim 64, no mask.
//   128 queries per workgroup (4 waves x 32), key/value tiles of 64 staged by LDS-DMA into a 2-deep ring; all query
//   blocks of one (image, head) are placed on one XCD.  Defer-max softmax with the reference max in the MFMA's C operand.
//   Scores are computed TRANSPOSED (S^T = K Q^T with v_mfma_f32_32x32x16_f16) so a query's scores sit in the
//   registers of one lane pair (l, l^32): the row max needs one cross-half exchange and no LDS (the row sum is taken on the
//   matrix pipe: one v_mfma_f32_16x16x32_f16 with a 0 / 1 A operand per P fragment); the
//   exponentiated accumulator is, after a pairwise fp16 convert, directly the B operand of O^T += V^T P^T
//   (k order inside a step is the accumulator's row order, so V^T is fetched with ds_read_b64_tr_b16 in that same
//   order).  O^T leaves through an LDS transpose as 16-byte row stores.
// pt_attn_temporal_f16 : attention over the <=16 frames of one spatial position (HBM-bound, 0.05 % of the flops):
//   one wave per (clip, position, head), scores and PV on the matrix cores (v_mfma_f32_16x16x32_f16 / 16x16x16),
//   operands as (frame, 8 channels) fragments straight from global memory; details at the kernel.
#include "pt_common.h"

namespace {

// ======================================================================================= spatial
constexpr int QB = 128, KB = 64, HD = 64;
constexpr int KV_TILE = KB * HD * 2;          // 8 KiB
constexpr int OROW = 144;                     // bytes per staged output row (128 + 16 pad)

typedef f16 f16x4v __attribute__((ext_vector_type(4)));

typedef __fp16 hw_f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of halfs, delivered column-major
__device__ __forceinline__ f16x4v lds_tr16(const char* p) {
    const hw_f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hw_f16x4*)p);
    return __builtin_bit_cast(f16x4v, v);
}

// Softmax bookkeeping (VALU diet: the kernel is VALU-bound at head_dim 64 - one exp per score against half as many
// MFMA cycles per score as at head_dim 128).
//   * reference max instead of running max: scores leave the MFMA already relative to m_ref (the chain's C operand is a
//     16-register block holding -m_ref), m_ref is only re-based when some query's tile maximum exceeds it by more than
//     THR (defer-max, guide T13) - always in tile 0, almost never afterwards - so the per-tile subtraction and the
//     O-wide rescale are gone from the steady state.  P <= 2^THR = 256 is exact enough in fp16 (same 11 bits), sums are fp32.
//   * PRE: Q arrives pre-multiplied by scale * log2(e) (one rounding, in the QKV projection's epilogue: cs_scale of
//     pt_igemm_f16), so p = exp2(score) with no VALU op between the MFMA and v_exp_f32.  Otherwise p = exp2(score * c).
//   * the lane-pair exchange (query's other key half) is one v_permlane32_swap, not an LDS bpermute.
// Workgroup order: all query blocks of one (image, head) run on ONE XCD (ids equal mod 8), so its K/V (2.4 MB at
// S = 9216) are fetched into one L2 instead of eight.
// Lane-pair exchange (l, l ^ 32) without LDS: v_permlane32_swap exchanges the upper half of its first operand with
// the lower half of its second, so with both operands holding v the two registers hold {own, partner's} in every lane.
// Written as inline asm: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) used result 0 for both elements of the
// returned pair (max(r0, r1) compiled to r0, r0 + r1 to 2 r0 - every lane silently kept only the LOWER lane's value).
// The s_nop covers the VALU-write -> permlane-read hazard (2 wait states) inside the statement.
__device__ __forceinline__ void pair_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float pair_max(float v) {
    float a = v, b = v;
    pair_swap(a, b);
    return fmaxf(a, b);
}

template <bool PRE>
__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(const f16* __restrict__ qkv, int ld, int k_off,
                                                              int v_off, f16* __restrict__ out, int ldo, int S, int nqb,
                                                              int heads, int ngroups, float cexp,
                                                              const f16* __restrict__ zeros) {
    __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE + 0];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = (idx / nqb) * 8 + xcd;                  // (image, head) pair; the grid is padded to 8 pairs per round
    if (grp >= ngroups) return;
    const int qb = idx % nqb, head = grp % heads, img = grp / heads;
    const size_t row0 = (size_t)img * S;
    const int hcol = head * HD;
    constexpr float THR_LOG2 = 8.0f;
    const float thr = PRE ? THR_LOG2 : THR_LOG2 / cexp;

    // ---- Q fragments (B operand): lane (q = l & 31, hh = l >> 5) holds Q[q][16 ks + 8 hh .. +7]
    const int ql = lane & 31, hh = lane >> 5;
    const int qrow = qb * QB + wave * 32 + ql;
    f16x8 qf[4];
    {
        const f16* qp = qrow < S ? qkv + (row0 + qrow) * ld + hcol + 8 * hh : zeros;
        const int step = qrow < S ? 16 : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qp + ks * step);
    }

    // ---- staging: thread copies chunk slots t and t + 256 of the K tile and of the V tile.  Full tiles: a wave-uniform
    // tile base (scalar arithmetic) + a per-thread 32-bit offset fixed for the whole kernel; only a ragged last tile
    // pays the per-lane bounds select.
    const int cphys = t & 7, csrc = cphys ^ ((t >> 4) & 7);
    const f16* zsrc = zeros + (lane & 7) * 8;
    const unsigned loff0 = (unsigned)((t >> 3) * ld + csrc * 8), loff1 = loff0 + 32u * (unsigned)ld;   // elements
    const f16* const kbase = qkv + row0 * ld + hcol + k_off;
    const f16* const vbase = qkv + row0 * ld + hcol + v_off;
    auto stage = [&](int kt, int buf) {
        char* Ks = smem + buf * 2 * KV_TILE + wave * 1024;
        char* Vs = Ks + KV_TILE;
        const f16* kt_k = kbase + (size_t)kt * KB * ld;      // wave-uniform
        const f16* kt_v = vbase + (size_t)kt * KB * ld;
        if (kt * KB + KB <= S) {
            pt_glds16(kt_k + loff0, Ks);
            pt_glds16(kt_v + loff0, Vs);
            pt_glds16(kt_k + loff1, Ks + 4096);
            pt_glds16(kt_v + loff1, Vs + 4096);
        } else {
            const int key0 = kt * KB + (t >> 3);
            pt_glds16(key0 < S ? kt_k + loff0 : zsrc, Ks);
            pt_glds16(key0 < S ? kt_v + loff0 : zsrc, Vs);
            pt_glds16(key0 + 32 < S ? kt_k + loff1 : zsrc, Ks + 4096);
            pt_glds16(key0 + 32 < S ? kt_v + loff1 : zsrc, Vs + 4096);
        }
    };

    f32x16 ot[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[0][r] = 0.f; ot[1][r] = 0.f; negm[r] = 0.f; }
    float nm = 0.f;                                         // nm = -m_ref of this lane's query
    // Row sums on the matrix pipe: l[q] = sum_k P[q][k] is one v_mfma_f32_16x16x32_f16 per P fragment with a 0 / 1 A operand.  Fed
    // with the 32x32x16 B fragment, lane L lands in column L & 15, k group L >> 4 - i.e. queries n and n + 16 share a column, in the
    // even and odd k groups: A row 0 = ones on k groups {0, 2} (lanes 0, 32), row 1 = ones on {1, 3} (lanes 17, 49), so
    // D[0][n] = l[n], D[1][n] = l[n + 16] (both key halves summed): lanes 0 .. 15, registers 0 and 1.  32 v_add_f32 per tile leave
    // the VALU; the sums are those of the fp16 P the PV product uses.  (Round 5: at the 1 300 W this kernel draws the launch is
    // bound by ENERGY, not by a pipe - 2.0 GHz, every re-scheduling of the tile within 3 % in time and joules
    // (profiles/r05/attn/) - so what pays is fewer instructions: this, and the re-base below updating negm IN PLACE
    // (`negm[r] = nm` made hipcc carry the 16-register block through 24 v_mov per tile).)
    const bool one = lane == 0 || lane == 32 || lane == 17 || lane == 49;
    const f16 onev = one ? (f16)1.0f : (f16)0.0f;
    const f16x8 onesA = {onev, onev, onev, onev, onev, onev, onev, onev};
    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};

    // lane-constant LDS offsets
    const int kswz = (ql >> 1) & 7;                         // K row = kb*32 + ql -> (row >> 1) & 7 = (ql >> 1) & 7 (+16 kb = 0 mod 8)
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3; // transposed read: row q, columns 4p..4p+3 of the 4 x 16 block
    const int dhalf = (lane >> 4) & 1;
    int koff[4], voff[2][4][2];                              // byte offsets inside a K / V tile
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ql * 128 + (((2 * ks + hh) ^ kswz) * 16);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int dcol = db * 32 + dhalf * 16 + 4 * tp;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int ra = (k4 >> 1) * 32 + 16 * (k4 & 1) + 4 * hh + tq, rb = ra + 8;
            voff[db][k4][0] = ra * 128 + (((dcol >> 3) ^ ((ra >> 1) & 7)) * 16) + (dcol & 7) * 2;
            voff[db][k4][1] = rb * 128 + (((dcol >> 3) ^ ((rb >> 1) & 7)) * 16) + (dcol & 7) * 2;
        }
    }

    const int nkt = (S + KB - 1) / KB;
    const bool ragged = (S & (KB - 1)) != 0;
    stage(0, 0);
    __syncthreads();

    auto tile = [&](int kt, int buf) {
        const char* Ks = smem + buf * 2 * KV_TILE;
        const char* Vs = Ks + KV_TILE;
        // ---- S^T - m_ref = K Q^T + (-m_ref): two 32-key blocks, every K fragment fetched before the first MFMA
        f16x8 kf[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *(const f16x8*)(Ks + kb * 4096 + koff[ks]);
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][0], qf[0], negm, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[ks], st[kb], 0, 0, 0);
        }
        // ---- V^T fragments for the whole tile: their LDS latency hides under the softmax arithmetic below
        f16x8 vf[2][4];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const f16x4v lo = lds_tr16(Vs + voff[db][k4][0]);
                const f16x4v hi = lds_tr16(Vs + voff[db][k4][1]);
                vf[db][k4] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        // the next tile's copies are issued behind every LDS read of this one: the compiler drains the LDS-DMA queue
        // (vmcnt(0)) in front of the first LDS read that follows a copy, so issued any earlier they would be waited for
        // at once; from here they fly under the softmax and PV phases until the barrier
        if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);
        if (ragged && kt == nkt - 1) {                      // ragged last tile: keys >= S never win the softmax
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= S) st[kb][r] = -INFINITY;
                }
        }
        // ---- does any of this query's scores exceed m_ref by more than THR?  Only a POSITIVE maximum matters (the scores are relative to
        // m_ref; the re-base step is max(mx, 0)), and positive floats order like their bit patterns: the chain is v_max3_i32 on the raw
        // scores - a negative result is merely "some negative score".  fmaxf() on MFMA outputs costs a NaN-quieting `v_max_f32 x, x, x`
        // per leaf (5 of the tile's 78 VALU instructions); tile 0, which needs the true maximum, takes the float chain inside the branch.
        int mi = __float_as_int(st[0][0]);                   // (not __builtin_bit_cast on a vector ELEMENT: clang reads element 0 for every index)
#pragma unroll
        for (int r = 0; r < 16; ++r) { mi = max(mi, __float_as_int(st[0][r])); mi = max(mi, __float_as_int(st[1][r])); }
        {
            int a = mi, b = mi;                               // the partner lane's (other key half): one v_permlane32_swap
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
            mi = max(a, b);
        }
        const float mxp = __int_as_float(mi);
        if (kt == 0 || __any(mxp > thr)) {                   // re-base m_ref (wave-uniform; tile 0 always, later rarely)
            float delta;
            if (kt == 0) {
                float mx = st[0][0];
#pragma unroll
                for (int r = 0; r < 16; ++r) { mx = fmaxf(mx, st[0][r]); mx = fmaxf(mx, st[1][r]); }
                delta = pair_max(mx);
            } else {
                delta = fmaxf(mxp, 0.f);
            }
            const float alpha = __builtin_amdgcn_exp2f(PRE ? -delta : -delta * cexp);
            lacc[0] *= alpha; lacc[1] *= __shfl(alpha, (lane & 15) + 16);
            nm -= delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ot[0][r] *= alpha; ot[1][r] *= alpha;
                st[0][r] -= delta; st[1][r] -= delta;
                negm[r] -= delta;
            }
        }
        f16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(PRE ? st[kb][r] : st[kb][r] * cexp);
                pf[kb][r >> 3][r & 7] = (f16)pv;
            }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[db][k4], pf[k4 >> 1][k4 & 1], ot[db], 0, 0, 0);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) lacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(onesA, pf[k4 >> 1][k4 & 1], lacc, 0, 0, 0);
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) { tile(kt, 0); tile(kt + 1, 1); }
    if (kt < nkt) tile(kt, 0);

    // ---- normalise, transpose through LDS, 16-byte row stores
    const float l0 = __shfl(lacc[0], ql & 15), l1 = __shfl(lacc[1], ql & 15);
    const float inv = 1.0f / (ql < 16 ? l0 : l1);
    char* Os = smem + wave * (32 * OROW);                    // 4.5 KiB per wave, inside the (now idle) ring
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16x4v o4;
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = (f16)(ot[db][4 * g + j] * inv);
            *(f16x4v*)(Os + ql * OROW + (db * 32 + 8 * g + 4 * hh) * 2) = o4;
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 8 + (lane >> 3), c = lane & 7;
        const int qr = qb * QB + wave * 32 + r;
        if (qr < S) *(f16x8*)(out + (row0 + qr) * ldo + hcol + c * 8) = *(const f16x8*)(Os + r * OROW + c * 16);
    }
}

// ======================================================================================= temporal
// One wave per (clip b, position s, head): attention over the F frames of one pixel, head_dim 64 or 128 (HDIM), on the matrix
// cores.  (The first version did the 14 x 14 x 64 products on the VALU, ~1000 instructions per task, and ran
// VALU-bound at 2.6-2.7 TB/s; this one is ~150 VALU + 6 MFMAs and streams at the HBM rate.)
// Frames come in NB blocks of 16: NB = 1 for F <= 16 (SVD: 14), NB = 2 for F <= 32 (SVD-XT and the reference's in-tree
// default num_frames = 25, /root/reference/models/controlnet_sdv.py:263).  Per (key block kb, query block qb):
//   S^T = K Q^T   : 2 x v_mfma_f32_16x16x32_f16; both operands are (frame = 16 blk + (lane & 15), 8 consecutive d)
//                   fragments loaded straight from global memory (16 B per lane, frames >= F read as zero)
//   softmax over the key frame k = 16 kb + 4 (lane >> 4) + j: in-lane over (kb, j), then two cross-lane steps (xor 16, 32)
//   O^T = V^T P^T : 4 x v_mfma_f32_16x16x16_f16 per key block (one per 16 channels); P^T is used as the B operand exactly as
//                   the first product left it in the accumulator, V^T[d][k] is gathered from the V rows staged in LDS
//   store         : lane (q = lane & 15) owns 4 consecutive channels per block: 8-byte stores, 32 B per quarter-wave
constexpr int TF_MAX = 32;

template <int NB, int HDIM>
__global__ __launch_bounds__(256) void attn_temporal_kernel(const f16* __restrict__ qkv, int ld, int k_off, int v_off,
                                                            f16* __restrict__ out, int ldo, int F, int S, int heads,
                                                            int64_t ntasks, float scale) {
    constexpr int NS = HDIM / 32;                            // 32-deep steps of the score product
    constexpr int NV = NB * HDIM / 32;                       // 16-byte chunks of the [16 NB][HDIM] V image per lane
    __shared__ __attribute__((aligned(16))) f16 smem[4 * NB * 16 * HDIM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const int head = (int)(task % heads);
    const int64_t bs = task / heads;
    const int s = (int)(bs % S);
    const int64_t b = bs / S;
    const int c = lane & 15, g = lane >> 4;
    f16* T = smem + wave * (NB * 16 * HDIM);                 // V rows [frame][HDIM]
    const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    // ---- issue every global load first: Q / K fragments (NS each per frame block) and the V rows
    f16x8 qf[NB][NS], kf[NB][NS];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const int f = 16 * blk + c;
        const f16* rowc = qkv + ((b * F + (f < F ? f : 0)) * (int64_t)S + s) * ld + head * HDIM + g * 8;
#pragma unroll
        for (int h = 0; h < NS; ++h) {
            qf[blk][h] = f < F ? *(const f16x8*)(rowc + 32 * h) : zero8;
            kf[blk][h] = f < F ? *(const f16x8*)(rowc + k_off + 32 * h) : zero8;
        }
    }
    f16x8 vrow[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + 64 * i, f = idx / (HDIM / 8);  // 16-byte chunk idx of the [16 NB][HDIM] V image
        vrow[i] = f < F ? *(const f16x8*)(qkv + ((b * F + f) * (int64_t)S + s) * ld + v_off + head * HDIM + (idx % (HDIM / 8)) * 8) : zero8;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) *(f16x8*)(T + (lane + 64 * i) * 8) = vrow[i];
    // ---- S^T[k][q] per (key block, query block)  (lane: k = 16 kb + 4 g + j, q = 16 qb + c)
    f16x4 pt[NB][NB];                                        // [kb][qb]
    float inv[NB];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb) {
        float p[NB][4], mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < NS; ++h) st = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][h], qf[qb][h], st, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[kb][j] = (16 * kb + 4 * g + j < F) ? st[j] * scale : -INFINITY;
                mx = fmaxf(mx, p[kb][j]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) { p[kb][j] = __expf(p[kb][j] - mx); sum += p[kb][j]; }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        inv[qb] = 1.0f / sum;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) pt[kb][qb] = (f16x4){(f16)p[kb][0], (f16)p[kb][1], (f16)p[kb][2], (f16)p[kb][3]};
    }
    // ---- O^T[d][q] = sum_k V[k][d] P^T[k][q]
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // this wave's V rows are in LDS
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int blk = 0; blk < HDIM / 16; ++blk) {
        f16x4 vt[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) vt[kb][j] = T[(16 * kb + 4 * g + j) * HDIM + 16 * blk + c];
#pragma unroll
        for (int qb = 0; qb < NB; ++qb) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) o = __builtin_amdgcn_mfma_f32_16x16x16f16(vt[kb], pt[kb][qb], o, 0, 0, 0);
            const int f = 16 * qb + c;
            if (f < F) {
                const f16x4 r = {(f16)(o[0] * inv[qb]), (f16)(o[1] * inv[qb]), (f16)(o[2] * inv[qb]), (f16)(o[3] * inv[qb])};
                *(f16x4*)(out + ((b * F + f) * (int64_t)S + s) * ldo + head * HDIM + 4 * g + 16 * blk) = r;
            }
        }
    }
}

}  // namespace

extern "C" int pt_attn_spatial_f16(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, void* out, int32_t ldo,
                                   int32_t Nimg, int32_t S, int32_t heads, int32_t head_dim, float scale, int32_t q_prescaled,
                                   void* stream) {
    PT_CHECK(qkv && out, "pt_attn_spatial_f16: null pointer");
    PT_CHECK(head_dim == 64, "pt_attn_spatial_f16: head_dim %d unsupported (this build handles 64, the SVD value)", head_dim);
    PT_CHECK(ld % 8 == 0 && ldo % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0, "pt_attn_spatial_f16: pitches/offsets must be multiples of 8");
    PT_CHECK(Nimg > 0 && S > 0 && heads > 0 && scale > 0.f, "pt_attn_spatial_f16: bad sizes");
    PT_CHECK(pt_zero_page(), "pt_attn_spatial_f16: zero page not set");
    hipStream_t s = (hipStream_t)stream;
    const float cexp = scale * 1.4426950408889634f;          // exp(scale * x) = exp2(cexp * x)
    const int nqb = (S + QB - 1) / QB;
    const long long ngroups = (long long)Nimg * heads, nblk = (ngroups + 7) / 8 * 8 * nqb;
    PT_CHECK(nblk < (1ll << 31), "pt_attn_spatial_f16: grid too large");
    pt_prof_begin(1, s, 4.0 * (double)Nimg * heads * (double)S * (double)S * 64.0);
    if (q_prescaled)
        hipLaunchKernelGGL(attn_spatial_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, (const f16*)qkv, ld, k_off, v_off,
                           (f16*)out, ldo, S, nqb, heads, (int)ngroups, cexp, (const f16*)pt_zero_page());
    else
        hipLaunchKernelGGL(attn_spatial_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, s, (const f16*)qkv, ld, k_off, v_off,
                           (f16*)out, ldo, S, nqb, heads, (int)ngroups, cexp, (const f16*)pt_zero_page());
    pt_prof_end(1, s);
    PT_LAUNCH_CHECK("pt_attn_spatial_f16");
    return 0;
}

extern "C" int pt_attn_temporal_f16(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, void* out, int32_t ldo,
                                    int32_t B, int32_t F, int32_t S, int32_t heads, int32_t head_dim, float scale,
                                    void* stream) {
    PT_CHECK(qkv && out, "pt_attn_temporal_f16: null pointer");
    PT_CHECK(head_dim == 64 || head_dim == 128, "pt_attn_temporal_f16: head_dim %d unsupported (64, 128)", head_dim);
    PT_CHECK(F >= 1 && F <= TF_MAX, "pt_attn_temporal_f16: %d frames unsupported (1..32)", F);
    PT_CHECK(ld % 8 == 0 && ldo % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0, "pt_attn_temporal_f16: pitches/offsets must be multiples of 8");
    const int64_t ntasks = (int64_t)B * S * heads;
    PT_CHECK(ntasks > 0 && (ntasks + 3) / 4 < (1ll << 31), "pt_attn_temporal_f16: bad sizes");
    const dim3 grid((unsigned)((ntasks + 3) / 4));
#define PT_TATTN(NB_, HD_)                                                                                                    \
    hipLaunchKernelGGL((attn_temporal_kernel<NB_, HD_>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)qkv, ld, k_off,  \
                       v_off, (f16*)out, ldo, F, S, heads, ntasks, scale)
    if (head_dim == 64) { if (F <= 16) PT_TATTN(1, 64); else PT_TATTN(2, 64); }
    else                { if (F <= 16) PT_TATTN(1, 128); else PT_TATTN(2, 128); }
#undef PT_TATTN
    PT_LAUNCH_CHECK("pt_attn_temporal_f16");
    return 0;
}
