// Attention kernels of the spatio-temporal transformer, gfx950.
//
// pt_attn_spatial_f16 : flash-style softmax(QK^T)V per (image, head), head_dim 64, no mask.
//   192 queries per workgroup (4 waves x 3 blocks of 16), key/value tiles of 64 staged by LDS-DMA into a 2-deep ring; all query
//   blocks of one (image, head) are placed on one XCD.  Defer-max softmax with the reference max in the MFMA's C operand.
//   Scores are computed TRANSPOSED (S^T = K Q^T, v_mfma_f32_16x16x32_f16 - the shape that holds the higher clock under the power
//   cap, which is what bounds this kernel) so a query's scores sit in four lanes: the steady state needs no cross-lane step at all
//   (the threshold test is a wave-wide OR of lane-local maxima, the row sum an MFMA against ones); the exponentiated accumulators
//   are, after a pairwise fp16 convert, directly the B operand of O^T += V^T P^T (the k order inside a step is the accumulators' row
//   order, V^T is fetched with ds_read_b64_tr_b16 in that same order).  O^T leaves through an LDS transpose as 16-byte row stores.
// pt_attn_temporal_f16 : attention over the <=16 frames of one spatial position (HBM-bound, 0.05 % of the flops):
//   one wave per (clip, position, head), scores and PV on the matrix cores (v_mfma_f32_16x16x32_f16 / 16x16x16),
//   operands as (frame, 8 channels) fragments straight from global memory; details at the kernel.
#include "pt_common.h"
#include <stdlib.h>

namespace {

// ======================================================================================= spatial
constexpr int KB = 64, HD = 64;
constexpr int KV_TILE = KB * HD * 2;          // 8 KiB
constexpr int OROW = 144;                     // bytes per staged output row (128 + 16 pad)

typedef f16 f16x4v __attribute__((ext_vector_type(4)));

typedef __fp16 hw_f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of halfs, delivered column-major
__device__ __forceinline__ f16x4v lds_tr16(const char* p) {
    const hw_f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hw_f16x4*)p);
    return __builtin_bit_cast(f16x4v, v);
}

// Softmax bookkeeping.
//   * reference max instead of running max: scores leave the MFMA already relative to m_ref (the chains' C operand holds -m_ref),
//     m_ref is only re-based when some query's tile maximum exceeds it by more than THR (defer-max, guide T13) - always in tile 0,
//     almost never afterwards - so the per-tile subtraction and the O-wide rescale are gone from the steady state.
//     P <= 2^THR = 256 is exact enough in fp16 (same 11 bits); the row sums are those of the fp16 P, accumulated in fp32.
//   * PRE: Q arrives pre-multiplied by scale * log2(e) (one rounding, in the QKV projection's epilogue: cs_scale of
//     pt_igemm_f16), so p = exp2(score) with no VALU op between the MFMA and v_exp_f32.  Otherwise p = exp2(score * c).
// Workgroup order: all query blocks of one (image, head) run on ONE XCD (ids equal mod 8), so its K/V (2.4 MB at
// S = 9216) are fetched into one L2 instead of eight.
// Cross-lane exchanges (only in the re-base path) without LDS: v_permlane32_swap exchanges the upper half of its first operand with
// the lower half of its second (v_permlane16_swap: odd rows of 16 lanes with even rows), so with both operands holding v the two
// registers hold {own, partner's} in every lane.  Written as inline asm: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2)
// used result 0 for both elements of the returned pair (max(r0, r1) compiled to r0 - every lane silently kept only the LOWER
// lane's value).  The s_nop covers the VALU-write -> permlane-read hazard (2 wait states) inside the statement.
__device__ __forceinline__ void pair_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// The tile on v_mfma_f32_16x16x32_f16 (round 5; rounds 1-4 ran the same tile and schedule on 32x32x16, one query per lane pair): on
// random data under the power cap the 16x16x32 shape holds a 13 % higher clock at equal cycles per flop
// (profiles/r03/mfma_shape_ab.txt; MI355X_MICROARCH.md, DVFS give-back (7)), and this kernel is bound by energy: 3.15 -> 3.00 ms,
// 4.25 -> 4.07 J per level-0 launch at 2.21 instead of 2.04 GHz (profiles/r05/attn/bench_power_clock_16x16x32.txt).
// Per wave: NQB = 3 query blocks x 4 key blocks of 16 x 16 scores (every K and V^T fragment read from LDS feeds NQB MFMAs: the LDS
// reads are 5 % of the launch's joules per halving, and 3 blocks is what 2 waves per SIMD leave registers for - 225 VGPRs; 4 blocks
// spill or, with just-in-time fragment reads, fall below the power cap into latency: profiles/r05/attn/bench_query_blocks_per_wave.txt),
//   S^T[kb][qb] = K[kb] Q[qb]^T + (-m_ref[qb])      A = K fragment (key = 16 kb + l % 16, d = 32 ds + 8 g ..), shared by all qb
//   lane (q = l % 16, g = l / 16) holds S^T[key = 16 kb + 4 g + j][q], j = 0 .. 3: a query's 64 scores sit in the four lanes l % 16 + 16 g
//   P^T as the B operand of a 32-key step s: lane group g supplies [st[2s][qb][0..3], st[2s+1][qb][0..3]] - i.e. the step's k index
//   8 g + jj stands for key 32 s + 16 (jj / 4) + 4 g + jj % 4, and V^T is fetched in that order: two ds_read_b64_tr_b16 of 4 key rows
//   O^T[db][qb] += V^T[db][s] P^T[s][qb]            4 d blocks x NQB q blocks; lane (q, g) holds O[q][16 db + 4 g + j]
//   row sums: A = ones: every row of D is the sum over the step's 32 keys, so all four registers of lacc[qb] hold l[q]
// The steady state has no cross-lane step: `__any` over the lanes' LOCAL maxima is the wave's test; only a re-base combines the four
// lanes of a query (v_permlane16_swap + v_permlane32_swap).
__device__ __forceinline__ void quad_swap16(float& a, float& b) {      // rows of 16 lanes: a's odd rows <-> b's even rows
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float query_max(float v) {                  // max over the four lanes l % 16 + 16 g
    float a = v, b = v;
    quad_swap16(a, b);
    v = fmaxf(a, b);
    a = v; b = v;
    pair_swap(a, b);
    return fmaxf(a, b);
}

template <bool PRE, int NQB>
__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(const f16* __restrict__ qkv, int ld, int k_off,
                                                              int v_off, f16* __restrict__ out, int ldo, int S, int nqb,
                                                              int heads, int ngroups, float cexp,
                                                              const f16* __restrict__ zeros) {
    __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE + 0];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = (idx / nqb) * 8 + xcd;                  // (image, head) pair; the grid is padded to 8 pairs per round
    if (grp >= ngroups) return;
    const int qb0 = idx % nqb, head = grp % heads, img = grp / heads;
    const size_t row0 = (size_t)img * S;
    const int hcol = head * HD;
    constexpr float THR_LOG2 = 8.0f;
    const float thr = PRE ? THR_LOG2 : THR_LOG2 / cexp;
    const int q16 = lane & 15, g = lane >> 4;

    // ---- Q fragments (B operand): lane (q, g) holds Q[16 qb + q][32 ds + 8 g .. +7]
    constexpr int QW = 16 * NQB, QBW = 4 * QW;             // queries per wave / per workgroup
    f16x8 qf[NQB][2];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int qrow = qb0 * QBW + wave * QW + qb * 16 + q16;
        const f16* qp = qrow < S ? qkv + (row0 + qrow) * ld + hcol + 8 * g : zeros;
        const int step = qrow < S ? 32 : 0;
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) qf[qb][ds] = *(const f16x8*)(qp + ds * step);
    }

    // ---- staging (unchanged): thread copies chunk slots t and t + 256 of the K tile and of the V tile
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

    f32x4 ot[4][NQB], negm[NQB], lacc[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        negm[qb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lacc[qb] = negm[qb];
#pragma unroll
        for (int db = 0; db < 4; ++db) ot[db][qb] = negm[qb];
    }
    const f16x8 ones = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};

    // lane-constant LDS offsets.  K fragment (kb, ds): row 16 kb + q16, logical 16-B chunk 4 ds + g, rows XOR-swizzled by (row >> 1) & 7
    int koff[2];
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) koff[ds] = q16 * 128 + (((4 * ds + g) ^ ((q16 >> 1) & 7)) * 16);       // + kb * 2048 (16 rows: swizzle unchanged)
    // V^T fragment (db, s): two transposed reads of 4 key rows x 16 d columns; in a 16-lane group lane j addresses row tq = j >> 2,
    // columns 4 (j & 3) .. + 3 and receives column j.  Rows: 32 s + 16 half + 4 g + tq
    const int tq = q16 >> 2, tp = q16 & 3;
    int voff[4][2];                                          // [db][half]; + s * 4096
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = 16 * half + 4 * g + tq, dcol = 16 * db + 4 * tp;
            voff[db][half] = row * 128 + (((dcol >> 3) ^ ((row >> 1) & 7)) * 16) + (dcol & 7) * 2;
        }

    const int nkt = (S + KB - 1) / KB;
    const bool ragged = (S & (KB - 1)) != 0;
    stage(0, 0);
    __syncthreads();

    auto tile = [&](int kt, int buf) {
        const char* Ks = smem + buf * 2 * KV_TILE;
        const char* Vs = Ks + KV_TILE;
        // ---- S^T - m_ref: 8 K fragments, each feeding both query blocks
        f16x8 kf[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) kf[kb][ds] = *(const f16x8*)(Ks + kb * 2048 + koff[ds]);
        f32x4 st[4][NQB];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                st[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][0], qf[qb][0], negm[qb], 0, 0, 0);
                st[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][1], qf[qb][1], st[kb][qb], 0, 0, 0);
            }
        // ---- V^T fragments for the whole tile: their LDS latency hides under the softmax arithmetic below
        f16x8 vf[4][2];
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x4v lo = lds_tr16(Vs + s * 4096 + voff[db][0]);
                const f16x4v hi = lds_tr16(Vs + s * 4096 + voff[db][1]);
                vf[db][s] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        // the next tile's copies are issued behind every LDS read of this one: hipcc drains the LDS-DMA queue (vmcnt(0)) in front of a
        // transposed read that follows a copy, so issued any earlier they would be waited for at once; from here they fly under the
        // softmax and PV phases until the barrier
        if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);
        if (ragged && kt == nkt - 1) {                      // ragged last tile: keys >= S never win the softmax
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (kt * KB + 16 * kb + 4 * g + j >= S) {
#pragma unroll
                        for (int qb = 0; qb < NQB; ++qb) st[kb][qb][j] = -INFINITY;
                    }
        }
        // ---- does any score exceed m_ref by more than THR?  Lane-local: only a POSITIVE maximum matters and positive floats order like
        // their bit patterns (v_max3_i32 on the raw scores, no NaN-quieting v_max pairs), and __any() is the wave's OR
        int mi = __float_as_int(st[0][0][0]);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) mi = max(mi, __float_as_int(st[kb][qb][j]));
        if (kt == 0 || __any(__int_as_float(mi) > thr)) {    // re-base m_ref (wave-uniform; tile 0 always, later rarely)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                float mx = st[0][qb][0];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, st[kb][qb][j]);
                mx = query_max(mx);
                const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(PRE ? -delta : -delta * cexp);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lacc[qb][j] *= alpha; negm[qb][j] -= delta;
#pragma unroll
                    for (int db = 0; db < 4; ++db) ot[db][qb][j] *= alpha;
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) st[kb][qb][j] -= delta;
                }
            }
        }
        f16x8 pf[2][NQB];                                    // [s][qb]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float sc = st[2 * s + (jj >> 2)][qb][jj & 3];
                    pf[s][qb][jj] = (f16)__builtin_amdgcn_exp2f(PRE ? sc : sc * cexp);
                }
        // ---- O^T += V^T P^T, row sums
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
                    ot[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[db][s], pf[s][qb], ot[db][qb], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[s][qb], lacc[qb], 0, 0, 0);
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) { tile(kt, 0); tile(kt + 1, 1); }
    if (kt < nkt) tile(kt, 0);

    // ---- normalise, transpose through LDS, 16-byte row stores: lane (q, g) holds O[16 qb + q][16 db + 4 g + j]
    static_assert(4 * QW * OROW <= 4 * KV_TILE, "the staged output rows of the four waves must fit the K/V ring");
    char* Os = smem + wave * (QW * OROW);                    // 2.25 KiB per query block and wave: 6.75 KiB at NQB = 3 (27 of the ring's 32 KiB), 4.5 at NQB = 2; the ring is idle by now
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const float inv = 1.0f / lacc[qb][0];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            f16x4v o4;
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = (f16)(ot[db][qb][j] * inv);
            *(f16x4v*)(Os + (16 * qb + q16) * OROW + (16 * db + 4 * g) * 2) = o4;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int pass = 0; pass < 2 * NQB; ++pass) {
        const int r = pass * 8 + (lane >> 3), c = lane & 7;
        const int qr = qb0 * QBW + wave * QW + r;
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

namespace { int g_attn_nqb = 3; }

// test / tuning hook (like pt_igemm_force_config): 192 (3, the default) or 128 (2) queries per workgroup.  A process-wide switch that a
// captured hipGraph does not see - set it before any capture.
extern "C" int pt_attn_spatial_set_nqb(int32_t nqb) {
    PT_CHECK(nqb == 2 || nqb == 3, "pt_attn_spatial_set_nqb: %d (2 or 3)", nqb);
    g_attn_nqb = nqb;
    return 0;
}

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
    const int nq = g_attn_nqb;                               // query blocks of 16 per wave (3; 2 = the A/B build of round 5, pt_attn_spatial_set_nqb)
    const int qbw = 64 * nq;
    const int nqb = (S + qbw - 1) / qbw;
    const long long ngroups = (long long)Nimg * heads, nblk = (ngroups + 7) / 8 * 8 * nqb;
    PT_CHECK(nblk < (1ll << 31), "pt_attn_spatial_f16: grid too large");
    pt_prof_begin(1, s, 4.0 * (double)Nimg * heads * (double)S * (double)S * 64.0);
#define PT_ATTN_LAUNCH(P_, N_) hipLaunchKernelGGL((attn_spatial_kernel<P_, N_>), dim3((unsigned)nblk), dim3(256), 0, s, (const f16*)qkv, ld, k_off, v_off, \
                           (f16*)out, ldo, S, nqb, heads, (int)ngroups, cexp, (const f16*)pt_zero_page())
    if (nq == 3) { if (q_prescaled) PT_ATTN_LAUNCH(true, 3); else PT_ATTN_LAUNCH(false, 3); }
    else { if (q_prescaled) PT_ATTN_LAUNCH(true, 2); else PT_ATTN_LAUNCH(false, 2); }
#undef PT_ATTN_LAUNCH
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
