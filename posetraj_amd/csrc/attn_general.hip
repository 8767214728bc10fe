// pt_attn_f16: flash-style softmax(Q K^T * scale) V per (batch, head) for the head sizes beside the U-Net's 64 - the VAE's
// single 512-wide head (AutoencoderKLTemporalDecoder mid blocks, S = h*w per frame), CLIP ViT-H's 80 (S = 257) and the 128
// of the reference's in-tree default num_attention_heads = (5,10,10,20) at level 2.  Self- or cross-attention (Sq != Sk).
//
// Two kernels: attn_general_kernel<D> (head_dim 64 / 80 / 128) below, attn_d512_kernel further down.
// attn_general_kernel: one workgroup = 4 waves x 16 queries; key / value tiles of 32 rows go global -> registers -> LDS (rows padded by 32 bytes:
// conflict-free ds_read_b128 of the K fragments and ds_read_b64_tr_b16 of the V^T fragments), the loads of tile t+1 are in
// flight while tile t is multiplied.  Everything is computed TRANSPOSED like the head_dim-64 kernel (attn.hip):
//   S^T[key][q] = K Q^T      v_mfma_f32_16x16x32_f16 (+ one 16x16x16 step when head_dim % 32 == 16); the Q fragments stay
//                             in registers for the whole kernel; lane (q = lane & 15, g = lane >> 4) ends up with the scores
//                             of keys 16 kb + 4 g + {0..3} of its query: row max = in-lane + two cross-lane steps
//   O^T[d][q]  += V^T P^T    v_mfma_f32_16x16x32_f16: the exponentiated accumulators of the two key blocks ARE the B operand
//                             (k index 8 g + j <-> key 16 (j >> 2) + 4 g + (j & 3)); V^T comes out of LDS in that same
//                             key order by two transposed reads per MFMA
// Online softmax in fp32 (running max and sum per query; O is rescaled only when some query's maximum moved), P in fp16.
// Keys >= Sk score -inf and their V rows are staged as zeros; queries >= Sq are computed on row Sq-1 and not stored.
#include "pt_common.h"

namespace {

typedef f16 f16x4v __attribute__((ext_vector_type(4)));
typedef __fp16 hw_f16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ f16x4v ag_lds_tr16(const char* p) {
    const hw_f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hw_f16x4*)p);
    return __builtin_bit_cast(f16x4v, v);
}

constexpr int AG_KT = 32;       // keys per tile
constexpr int AG_QB = 64;       // queries per workgroup

template <int D>
__global__ __launch_bounds__(256, (D > 128 ? 1 : 2)) void attn_general_kernel(
        const f16* __restrict__ q, int ldq, const f16* __restrict__ k, int ldk, const f16* __restrict__ v, int ldv,
        f16* __restrict__ out, int ldo, int Sq, int Sk, int nqb, int heads, int ngroups, float cexp, float* __restrict__ lse) {
    constexpr int PITCH = 2 * D + 32;                 // bytes per staged row
    constexpr int NS32 = D / 32, REM16 = (D % 32) / 16, NDB = D / 16;
    constexpr int CPR = D / 8;                         // 16-byte chunks per row
    constexpr int NCH = (AG_KT * CPR + 255) / 256;     // chunks per thread and operand
    extern __shared__ __attribute__((aligned(16))) char ag_smem[];
    char* const Ks = ag_smem;
    char* const Vs = ag_smem + AG_KT * PITCH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = lane & 15, g = lane >> 4;
    // all query blocks of one (batch, head) share an XCD (ids equal mod 8): its K / V are streamed into one L2
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = (idx / nqb) * 8 + xcd;
    if (grp >= ngroups) return;
    const int qb = idx % nqb, head = grp % heads, bat = grp / heads;
    const int hcol = head * D;
    const f16* const kbase = k + (size_t)bat * Sk * ldk + hcol;
    const f16* const vbase = v + (size_t)bat * Sk * ldv + hcol;

    // ---- Q fragments (B operand): lane (q = c, g) holds Q[q][32 s + 8 g .. + 7]
    int qrow = qb * AG_QB + wave * 16 + c;
    const bool qok = qrow < Sq;
    if (!qok) qrow = Sq - 1;
    const f16* qp = q + ((size_t)bat * Sq + qrow) * ldq + hcol;
    f16x8 qf[NS32 > 0 ? NS32 : 1];
    f16x4 qf4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
    for (int s = 0; s < NS32; ++s) qf[s] = *(const f16x8*)(qp + 32 * s + 8 * g);
    if (REM16) qf4 = *(const f16x4*)(qp + D - 16 + 4 * g);

    // ---- staging
    f16x8 kreg[NCH], vreg[NCH];
    const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = t + 256 * i, row = id / CPR, cc = id - row * CPR;
            const int key = kt * AG_KT + row;
            const bool ok = (AG_KT * CPR % 256 == 0 || id < AG_KT * CPR) && key < Sk;
            kreg[i] = ok ? *(const f16x8*)(kbase + (size_t)key * ldk + cc * 8) : zero8;
            vreg[i] = ok ? *(const f16x8*)(vbase + (size_t)key * ldv + cc * 8) : zero8;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = t + 256 * i, row = id / CPR, cc = id - row * CPR;
            if (AG_KT * CPR % 256 == 0 || id < AG_KT * CPR) {
                *(f16x8*)(Ks + row * PITCH + cc * 16) = kreg[i];
                *(f16x8*)(Vs + row * PITCH + cc * 16) = vreg[i];
            }
        }
    };

    f32x4 ot[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) ot[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;              // l_run: this lane's share (its 8 keys per tile) of the row sum

    const int tq = c >> 2, tp = c & 3;                 // transposed read: row tq, columns 4 tp .. + 3 of the 4 x 16 block
    const char* const kfrag = Ks + c * PITCH + 16 * g;
    const char* const vfrag = Vs + (4 * g + tq) * PITCH + 8 * tp;

    const int nkt = (Sk + AG_KT - 1) / AG_KT;
    fetch(0);
    commit();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch(kt + 1);
        // ---- S^T = K Q^T for the two 16-key blocks
        f32x4 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS32; ++s) {
                const f16x8 kf = *(const f16x8*)(kfrag + kb * 16 * PITCH + 64 * s);
                st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], st[kb], 0, 0, 0);
            }
            if (REM16) {
                const f16x4 kf = *(const f16x4*)(Ks + (kb * 16 + c) * PITCH + (D - 16 + 4 * g) * 2);
                st[kb] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qf4, st[kb], 0, 0, 0);
            }
        }
        // ---- online softmax (scores in units of log2)
        float sv[2][4], mt = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = kt * AG_KT + kb * 16 + 4 * g + i;
                sv[kb][i] = key < Sk ? st[kb][i] * cexp : -INFINITY;
                mt = fmaxf(mt, sv[kb][i]);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        f16x8 pf;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __builtin_amdgcn_exp2f(sv[kb][i] - m_new);
                psum += p;
                pf[4 * kb + i] = (f16)p;
            }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) ot[db] *= alpha;
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const f16x4v lo = ag_lds_tr16(vfrag + db * 32);
            const f16x4v hi = ag_lds_tr16(vfrag + 16 * PITCH + db * 32);
            const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            ot[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, ot[db], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            commit();
            __syncthreads();
        }
    }
    // ---- normalise and store: lane (q = c, g) owns d = 16 db + 4 g .. + 3
    float l = l_run;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    if (qok) {
        f16* op = out + ((size_t)bat * Sq + qrow) * ldo + hcol + 4 * g;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const f16x4 o4 = {(f16)(ot[db][0] * inv), (f16)(ot[db][1] * inv), (f16)(ot[db][2] * inv), (f16)(ot[db][3] * inv)};
            *(f16x4*)(op + db * 16) = o4;
        }
        // log2-domain log-sum-exp of the scaled scores: what the training step's backward (attn_bwd.hip) rebuilds P from
        if (lse && g == 0) lse[((size_t)bat * Sq + qrow) * heads + head] = m_run + __log2f(l);
    }
}

// ---------------------------------------------------------------------------------------------------- head_dim 512
// The VAE's single 512-wide head (S = 9 216 tokens per frame at 576 x 1024).  Same arithmetic as the kernel above, laid out for
// this size: 8 waves x 16 queries per workgroup (two waves per SIMD: one's softmax / LDS reads run beside the other's MFMAs);
// a K or V row is 1 024 B = exactly one 64-lane LDS-DMA, so the tiles are staged by global_load_lds straight into padded rows
// (no staging registers), double-buffered, the copies of tile t+1 in flight under tile t; the 16-step score reduction over
// head_dim is split into four independent accumulators per key block so that the matrix pipe never waits on its own result.
constexpr int A5_D = 512, A5_PITCH = 2 * A5_D + 32, A5_TILE = AG_KT * A5_PITCH, A5_QB = 128;

__global__ __launch_bounds__(512, 2) void attn_d512_kernel(
        const f16* __restrict__ q, int ldq, const f16* __restrict__ k, int ldk, const f16* __restrict__ v, int ldv,
        f16* __restrict__ out, int ldo, int Sq, int Sk, int nqb, int heads, int ngroups, float cexp, const f16* __restrict__ zeros) {
    constexpr int D = A5_D, PITCH = A5_PITCH, NS32 = D / 32, NDB = D / 16;
    extern __shared__ __attribute__((aligned(16))) char ag_smem[];     // [2 buffers][K tile | V tile]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = (idx / nqb) * 8 + xcd;
    if (grp >= ngroups) return;
    const int qb = idx % nqb, head = grp % heads, bat = grp / heads;
    const int hcol = head * D;
    const f16* const kbase = k + (size_t)bat * Sk * ldk + hcol + lane * 8;
    const f16* const vbase = v + (size_t)bat * Sk * ldv + hcol + lane * 8;
    const f16* const zsrc = zeros + (lane & 15) * 8;

    int qrow = qb * A5_QB + wave * 16 + c;
    const bool qok = qrow < Sq;
    if (!qok) qrow = Sq - 1;
    const f16* qp = q + ((size_t)bat * Sq + qrow) * ldq + hcol;
    f16x8 qf[NS32];
#pragma unroll
    for (int s = 0; s < NS32; ++s) qf[s] = *(const f16x8*)(qp + 32 * s + 8 * g);

    // wave w copies rows 4 w .. 4 w + 3 of the K tile and of the V tile: 8 LDS-DMA instructions per wave and tile
    auto stage = [&](int kt, int buf) {
        char* Kb = ag_smem + buf * 2 * A5_TILE;
        char* Vb = Kb + A5_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 4 + i, key = kt * AG_KT + row;
            const bool ok = key < Sk;
            pt_glds16(ok ? kbase + (size_t)key * ldk : zsrc, Kb + row * PITCH);
            pt_glds16(ok ? vbase + (size_t)key * ldv : zsrc, Vb + row * PITCH);
        }
    };

    f32x4 ot[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) ot[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int tq = c >> 2, tp = c & 3;
    const int koff = c * PITCH + 16 * g, voff = (4 * g + tq) * PITCH + 8 * tp;

    const int nkt = (Sk + AG_KT - 1) / AG_KT;
    stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);              // (every wave left buffer buf ^ 1 at the barrier that ended tile kt - 1)
        const char* Ks = ag_smem + buf * 2 * A5_TILE;
        const char* Vs = Ks + A5_TILE;
        // ---- S^T = K Q^T: four independent accumulators per 16-key block over the 16 head_dim steps
        f32x4 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x4 part[4];
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) part[pi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS32; ++s) {
                const f16x8 kf = *(const f16x8*)(Ks + kb * 16 * PITCH + koff + 64 * s);
                part[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], part[s & 3], 0, 0, 0);
            }
            st[kb] = (part[0] + part[1]) + (part[2] + part[3]);
        }
        float sv[2][4], mt = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = kt * AG_KT + kb * 16 + 4 * g + i;
                sv[kb][i] = key < Sk ? st[kb][i] * cexp : -INFINITY;
                mt = fmaxf(mt, sv[kb][i]);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        f16x8 pf;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __builtin_amdgcn_exp2f(sv[kb][i] - m_new);
                psum += p;
                pf[4 * kb + i] = (f16)p;
            }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) ot[db] *= alpha;
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const f16x4v lo = ag_lds_tr16(Vs + voff + db * 32);
            const f16x4v hi = ag_lds_tr16(Vs + voff + 16 * PITCH + db * 32);
            const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            ot[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, ot[db], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // this wave's copies of tile kt + 1 have landed
        __syncthreads();                                       // ... everyone's have, and everyone is done reading buffer buf
    }
    float l = l_run;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    if (qok) {
        f16* op = out + ((size_t)bat * Sq + qrow) * ldo + hcol + 4 * g;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const f16x4 o4 = {(f16)(ot[db][0] * inv), (f16)(ot[db][1] * inv), (f16)(ot[db][2] * inv), (f16)(ot[db][3] * inv)};
            *(f16x4*)(op + db * 16) = o4;
        }
    }
}

int launch_attn_d512(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int nbatch, int Sq,
                     int Sk, int heads, float scale, hipStream_t s) {
    constexpr int LDS = 4 * A5_TILE;
    static bool attr_done[64] = {};
    const int dev = pt_device();
    if (!attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)attn_d512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done[dev] = true;
    }
    PT_CHECK(pt_zero_page(), "pt_attn_f16: zero page not set");
    const int nqb = (Sq + A5_QB - 1) / A5_QB;
    const long long ngroups = (long long)nbatch * heads, nblk = (ngroups + 7) / 8 * 8 * nqb;
    PT_CHECK(nblk < (1ll << 31), "pt_attn_f16: grid too large");
    hipLaunchKernelGGL(attn_d512_kernel, dim3((unsigned)nblk), dim3(512), LDS, s, (const f16*)q, ldq, (const f16*)k, ldk, (const f16*)v,
                       ldv, (f16*)out, ldo, Sq, Sk, nqb, heads, (int)ngroups, scale * 1.4426950408889634f, (const f16*)pt_zero_page());
    return 0;
}

template <int D>
int launch_attn_general(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                        int nbatch, int Sq, int Sk, int heads, float scale, hipStream_t s, float* lse = nullptr) {
    constexpr int LDS = 2 * AG_KT * (2 * D + 32);
    static bool attr_done[64] = {};
    const int dev = pt_device();
    if (!attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)attn_general_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done[dev] = true;
    }
    const int nqb = (Sq + AG_QB - 1) / AG_QB;
    const long long ngroups = (long long)nbatch * heads, nblk = (ngroups + 7) / 8 * 8 * nqb;
    PT_CHECK(nblk < (1ll << 31), "pt_attn_f16: grid too large");
    hipLaunchKernelGGL(attn_general_kernel<D>, dim3((unsigned)nblk), dim3(256), LDS, s, (const f16*)q, ldq, (const f16*)k, ldk,
                       (const f16*)v, ldv, (f16*)out, ldo, Sq, Sk, nqb, heads, (int)ngroups, scale * 1.4426950408889634f, lse);
    return 0;
}

}  // namespace

// forward of the TRAINING step's spatial self-attention: pt_attn_f16 that also writes lse[(batch Sq + q) heads + head] =
// log2 sum_key exp2(score scale log2 e) - the one number per query the flash backward needs to rebuild P without the scores
extern "C" int pt_attn_fwd_lse_f16(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                                   int32_t ldo, int32_t nbatch, int32_t Sq, int32_t Sk, int32_t heads, int32_t head_dim, float scale,
                                   float* lse, void* stream) {
    PT_CHECK(q && k && v && out && lse, "pt_attn_fwd_lse_f16: null pointer");
    PT_CHECK(nbatch > 0 && Sq > 0 && Sk > 0 && heads > 0 && scale > 0.f, "pt_attn_fwd_lse_f16: bad sizes");
    PT_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "pt_attn_fwd_lse_f16: pitches must be multiples of 8 (out: 4)");
    auto al = [](const void* p, int a) { return ((uintptr_t)p & (a - 1)) == 0; };
    PT_CHECK(al(q, 16) && al(k, 16) && al(v, 16) && al(out, 8), "pt_attn_fwd_lse_f16: q / k / v must be 16-byte aligned, out 8-byte");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (head_dim) {
        case 64:  rc = launch_attn_general<64>(q, ldq, k, ldk, v, ldv, out, ldo, nbatch, Sq, Sk, heads, scale, s, lse); break;
        case 128: rc = launch_attn_general<128>(q, ldq, k, ldk, v, ldv, out, ldo, nbatch, Sq, Sk, heads, scale, s, lse); break;
        default:
            PT_CHECK(false, "pt_attn_fwd_lse_f16: head_dim %d unsupported (64, 128)", head_dim);
    }
    if (rc) return rc;
    PT_LAUNCH_CHECK("pt_attn_fwd_lse_f16");
    return 0;
}

extern "C" int pt_attn_f16(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                           int32_t ldo, int32_t nbatch, int32_t Sq, int32_t Sk, int32_t heads, int32_t head_dim, float scale,
                           void* stream) {
    PT_CHECK(q && k && v && out, "pt_attn_f16: null pointer");
    PT_CHECK(nbatch > 0 && Sq > 0 && Sk > 0 && heads > 0 && scale > 0.f, "pt_attn_f16: bad sizes");
    PT_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "pt_attn_f16: pitches must be multiples of 8 (out: 4)");
    auto al = [](const void* p, int a) { return ((uintptr_t)p & (a - 1)) == 0; };
    PT_CHECK(al(q, 16) && al(k, 16) && al(v, 16) && al(out, 8), "pt_attn_f16: q / k / v must be 16-byte aligned, out 8-byte");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (head_dim) {
        case 64:  rc = launch_attn_general<64>(q, ldq, k, ldk, v, ldv, out, ldo, nbatch, Sq, Sk, heads, scale, s); break;
        case 80:  rc = launch_attn_general<80>(q, ldq, k, ldk, v, ldv, out, ldo, nbatch, Sq, Sk, heads, scale, s); break;
        case 128: rc = launch_attn_general<128>(q, ldq, k, ldk, v, ldv, out, ldo, nbatch, Sq, Sk, heads, scale, s); break;
        case 512: rc = launch_attn_d512(q, ldq, k, ldk, v, ldv, out, ldo, nbatch, Sq, Sk, heads, scale, s); break;
        default:
            PT_CHECK(false, "pt_attn_f16: head_dim %d unsupported (64, 80, 128, 512)", head_dim);
    }
    if (rc) return rc;
    PT_LAUNCH_CHECK("pt_attn_f16");
    return 0;
}
