// Attention kernels of the spatio-temporal transformer, gfx950.
//
// pt_attn_spatial_f16 : flash-style softmax(QK^T)V per (image, head), head_dim 64, no mask.
//   128 queries per workgroup (4 waves x 32), key/value tiles of 64 staged by LDS-DMA into a 2-deep ring.
//   Scores are computed TRANSPOSED (S^T = K Q^T with v_mfma_f32_32x32x16_f16) so a query's scores sit in the
//   registers of one lane pair (l, l^32): the row max / sum need one cross-half exchange and no LDS; the
//   exponentiated accumulator is, after a pairwise fp16 convert, directly the B operand of O^T += V^T P^T
//   (k order inside a step is the accumulator's row order, so V^T is fetched with ds_read_b64_tr_b16 in that same
//   order).  O^T leaves through an LDS transpose as 16-byte row stores.
// pt_attn_temporal_f16 : attention over the <=16 frames of one spatial position (HBM-bound, 0.05 % of the flops):
//   one wave per (clip, position, head); the 3 x F x 128-byte rows are fetched as whole cache lines into LDS,
//   scores / softmax / PV on the VALU with lane = (query frame, quarter of head_dim).
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

__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(const f16* __restrict__ qkv, int ld, int k_off,
                                                              int v_off, f16* __restrict__ out, int ldo, int S,
                                                              float scale2, const f16* __restrict__ zeros) {
    __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE + 0];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int qb = blockIdx.x, head = blockIdx.y, img = blockIdx.z;
    const size_t row0 = (size_t)img * S;
    const int hcol = head * HD;

    // ---- Q fragments (B operand): lane (q = l & 31, hh = l >> 5) holds Q[q][16 ks + 8 hh .. +7]
    const int ql = lane & 31, hh = lane >> 5;
    const int qrow = qb * QB + wave * 32 + ql;
    f16x8 qf[4];
    {
        const f16* qp = qrow < S ? qkv + (row0 + qrow) * ld + hcol + 8 * hh : zeros;
        const int step = qrow < S ? 16 : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *(const f16x8*)(qp + ks * step);         // raw Q: the softmax scale rides in the exponent's FMA below
        }
    }

    // ---- staging: thread copies chunk slots t and t + 256 of the K tile and of the V tile
    const int cphys = t & 7, csrc = cphys ^ ((t >> 4) & 7);
    const f16* zsrc = zeros + (lane & 7) * 8;
    auto stage = [&](int kt, int buf) {
        char* Ks = smem + buf * 2 * KV_TILE;
        char* Vs = Ks + KV_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kt * KB + (t >> 3) + 32 * i;
            const f16* base = qkv + (row0 + key) * ld + hcol + csrc * 8;
            pt_glds16(key < S ? base + k_off : zsrc, Ks + (wave * 64 + 256 * i) * 16);
            pt_glds16(key < S ? base + v_off : zsrc, Vs + (wave * 64 + 256 * i) * 16);
        }
    };

    f32x16 ot[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[0][r] = 0.f; ot[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // lane-constant LDS offsets
    const int kswz = (ql >> 1) & 7;                         // K row = kb*32 + ql -> (row >> 1) & 7 = (ql >> 1) & 7 (+16 kb = 0 mod 8)
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3; // transposed read: row q, columns 4p..4p+3 of the 4 x 16 block
    const int dhalf = (lane >> 4) & 1;

    const int nkt = (S + KB - 1) / KB;
    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
        const char* Ks = smem + cur * 2 * KV_TILE;
        const char* Vs = Ks + KV_TILE;

        // ---- S^T = K Q^T : two 32-key blocks
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 kf = *(const f16x8*)(Ks + (kb * 32 + ql) * 128 + (((2 * ks + hh) ^ kswz) * 16));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], st[kb], 0, 0, 0);
            }
        }
        if (kt == nkt - 1 && (S & (KB - 1))) {              // ragged last tile: keys >= S never win the softmax
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= S) st[kb][r] = -INFINITY;
                }
        }
        // ---- online softmax (this lane: query ql, keys of half hh; partner lane ^ 32 has the other half)
        float mx = st[0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) { mx = fmaxf(mx, st[0][r]); mx = fmaxf(mx, st[1][r]); }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float nm = -m_new * scale2;
        float psum = 0.f;
        f16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[kb][r], scale2, nm));   // exp(scale * (s - max))
                psum += pv;
                pf[kb][r >> 3][r & 7] = (f16)pv;
            }
        if (__any(m_new > m_run)) {                          // the running max moved for some query of this wave: rescale
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale2);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ot[0][r] *= alpha; ot[1][r] *= alpha; }
            m_run = m_new;
        }
        l_run += psum;

        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int dcol = db * 32 + dhalf * 16 + 4 * tp;                  // first of this lane's 4 address columns
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int ra = kb * 32 + 16 * s + 4 * hh + tq, rb = ra + 8;
                    const f16x4v lo = lds_tr16(Vs + ra * 128 + ((((dcol >> 3) ^ ((ra >> 1) & 7)) * 16) + (dcol & 7) * 2));
                    const f16x4v hi = lds_tr16(Vs + rb * 128 + ((((dcol >> 3) ^ ((rb >> 1) & 7)) * 16) + (dcol & 7) * 2));
                    f16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][s], ot[db], 0, 0, 0);
                }
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- normalise, transpose through LDS, 16-byte row stores
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
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
// One wave per (clip b, position s, head): attention over the F <= 16 frames of one pixel, head_dim 64, on the matrix
// cores.  (The first version did the 14 x 14 x 64 products on the VALU, ~1000 instructions per task, and ran
// VALU-bound at 2.6-2.7 TB/s; this one is ~150 VALU + 6 MFMAs and streams at the HBM rate.)
//   S^T = K Q^T   : 2 x v_mfma_f32_16x16x32_f16; both operands are (frame = lane & 15, 8 consecutive d) fragments
//                   loaded straight from global memory (16 B per lane, frames >= F read as zero)
//   softmax over the key frame k = 4 (lane >> 4) + j: in-lane over j, then two cross-lane steps (xor 16, 32)
//   O^T = V^T P^T : 4 x v_mfma_f32_16x16x16_f16 (one per 16 channels); P^T is used as the B operand exactly as the
//                   first product left it in the accumulator, V^T[d][k] is gathered from the V rows staged in LDS
//   store         : lane (q = lane & 15) owns 4 consecutive channels per block: 8-byte stores, 32 B per quarter-wave
constexpr int TF_MAX = 16;

__global__ __launch_bounds__(256) void attn_temporal_kernel(const f16* __restrict__ qkv, int ld, int k_off, int v_off,
                                                            f16* __restrict__ out, int ldo, int F, int S, int heads,
                                                            int64_t ntasks, float scale) {
    __shared__ __attribute__((aligned(16))) f16 smem[4 * TF_MAX * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const int head = (int)(task % heads);
    const int64_t bs = task / heads;
    const int s = (int)(bs % S);
    const int64_t b = bs / S;
    const int c = lane & 15, g = lane >> 4;
    f16* T = smem + wave * (TF_MAX * 64);                    // V rows [frame][64]
    const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    // ---- issue every global load first: Q / K fragments (2 each) and the V rows (2 chunks per lane)
    const f16* rowc = qkv + ((b * F + (c < F ? c : 0)) * (int64_t)S + s) * ld + head * 64 + g * 8;
    f16x8 qf[2], kf[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        qf[h] = c < F ? *(const f16x8*)(rowc + 32 * h) : zero8;
        kf[h] = c < F ? *(const f16x8*)(rowc + k_off + 32 * h) : zero8;
    }
    f16x8 vrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = lane + 64 * i, f = idx >> 3;         // 16-byte chunk idx of the [16][64] V image
        vrow[i] = f < F ? *(const f16x8*)(qkv + ((b * F + f) * (int64_t)S + s) * ld + v_off + head * 64 + (idx & 7) * 8) : zero8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) *(f16x8*)(T + (lane + 64 * i) * 8) = vrow[i];
    // ---- S^T[k][q] (lane: k = 4 g + j, q = c)
    f32x4 st = {0.f, 0.f, 0.f, 0.f};
    st = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[0], st, 0, 0, 0);
    st = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[1], st, 0, 0, 0);
    float p[4], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p[j] = (4 * g + j < F) ? st[j] * scale : -INFINITY;
        mx = fmaxf(mx, p[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p[j] = __expf(p[j] - mx); sum += p[j]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const f16x4 pt = {(f16)p[0], (f16)p[1], (f16)p[2], (f16)p[3]};
    // ---- O^T[d][q] = sum_k V[k][d] P^T[k][q]
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // this wave's V rows are in LDS
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / sum;
    f16* orow = out + ((b * F + (c < F ? c : 0)) * (int64_t)S + s) * ldo + head * 64 + 4 * g;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        f16x4 vt;
#pragma unroll
        for (int j = 0; j < 4; ++j) vt[j] = T[(4 * g + j) * 64 + 16 * blk + c];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x16f16(vt, pt, zero4, 0, 0, 0);
        if (c < F) {
            const f16x4 r = {(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
            *(f16x4*)(orow + 16 * blk) = r;
        }
    }
}

}  // namespace

extern "C" int pt_attn_spatial_f16(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, void* out, int32_t ldo,
                                   int32_t Nimg, int32_t S, int32_t heads, int32_t head_dim, float scale, void* stream) {
    PT_CHECK(qkv && out, "pt_attn_spatial_f16: null pointer");
    PT_CHECK(head_dim == 64, "pt_attn_spatial_f16: head_dim %d unsupported (this build handles 64, the SVD value)", head_dim);
    PT_CHECK(ld % 8 == 0 && ldo % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0, "pt_attn_spatial_f16: pitches/offsets must be multiples of 8");
    PT_CHECK(Nimg > 0 && S > 0 && heads > 0 && heads <= 65535 && Nimg <= 65535, "pt_attn_spatial_f16: bad sizes");
    PT_CHECK(pt_zero_page(), "pt_attn_spatial_f16: zero page not set");
    hipStream_t s = (hipStream_t)stream;
    const float scale2 = scale * 1.4426950408889634f;
    pt_prof_begin(1, s, 4.0 * (double)Nimg * heads * (double)S * (double)S * 64.0);
    hipLaunchKernelGGL(attn_spatial_kernel, dim3((S + QB - 1) / QB, heads, Nimg), dim3(256), 0, s, (const f16*)qkv, ld,
                       k_off, v_off, (f16*)out, ldo, S, scale2, (const f16*)pt_zero_page());
    pt_prof_end(1, s);
    PT_LAUNCH_CHECK("pt_attn_spatial_f16");
    return 0;
}

extern "C" int pt_attn_temporal_f16(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, void* out, int32_t ldo,
                                    int32_t B, int32_t F, int32_t S, int32_t heads, int32_t head_dim, float scale,
                                    void* stream) {
    PT_CHECK(qkv && out, "pt_attn_temporal_f16: null pointer");
    PT_CHECK(head_dim == 64, "pt_attn_temporal_f16: head_dim %d unsupported (64 only)", head_dim);
    PT_CHECK(F >= 1 && F <= TF_MAX, "pt_attn_temporal_f16: %d frames unsupported (1..16)", F);
    PT_CHECK(ld % 8 == 0 && ldo % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0, "pt_attn_temporal_f16: pitches/offsets must be multiples of 8");
    const int64_t ntasks = (int64_t)B * S * heads;
    PT_CHECK(ntasks > 0 && (ntasks + 3) / 4 < (1ll << 31), "pt_attn_temporal_f16: bad sizes");
    hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)((ntasks + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)qkv, ld, k_off, v_off, (f16*)out, ldo, F, S, heads, ntasks, scale);
    PT_LAUNCH_CHECK("pt_attn_temporal_f16");
    return 0;
}
