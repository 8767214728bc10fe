// Flash backward of the spatial self-attentions for the TRAINING step (scripts/train_svd_traj_VIPSeg_14.py:1414,
// `accelerator.backward(loss)` through F.scaled_dot_product_attention): no score matrix ever reaches HBM.
//
// The first version recomputed S = Q K^T, P, dP = dO V^T and dS through pt_gemm_f16 + row kernels: per level-0 layer
// (14 frames x 5 heads x 2880^2 scores) that is 2.3 GB of fp32 written and read twice, 38 ms of a 201 ms step.  Here the two
// passes of a flash backward are re-shapings of the FORWARD kernel (attn_general.hip: everything transposed, 16 queries or
// keys per wave, tiles of 32 rows staged global -> registers -> LDS, accumulators re-used as MFMA operands):
//
//  pass dQ (query-stationary: a wave owns 16 queries, streams K / V tiles)
//      S^T = K Q^T, dP^T = V dO^T                    Q and dO fragments stay in registers
//      P^T = exp2(S^T c - L[q]),  dS^T = P^T (dP^T - Dq[q])            L: log2-sum-exp from the forward, Dq = sum_d dO O
//      dQ^T[d][q] += K^T dS^T                        K^T out of the staged tile by transposed reads, dS^T from the accumulators
//  pass dK dV (key-stationary: a wave owns 16 keys, streams Q / dO tiles)
//      S = Q K^T, dP = dO V^T                        K and V fragments stay in registers; rows are queries now
//      P = exp2(S c - L[row]),  dS = P (dP - Dq[row])
//      dV^T[d][key] += dO^T P,   dK^T[d][key] += Q^T dS                 dO^T / Q^T by transposed reads of the staged tiles
//
// v_mfma_f32_16x16x32_f16 throughout; P and dS are fp16 operands (like the forward's P), sums fp32.  head_dim 64 / 128.
#include "pt_common.h"

namespace {

typedef f16 f16x4b __attribute__((ext_vector_type(4)));
typedef __fp16 hw_f16x4b __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ f16x4b ab_lds_tr16(const char* p) {
    const hw_f16x4b v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hw_f16x4b*)p);
    return __builtin_bit_cast(f16x4b, v);
}

constexpr int AB_T = 32;        // rows per streamed tile
constexpr int AB_B = 64;        // stationary rows per workgroup (4 waves x 16)

// Dq[row, head] = sum_d dO[row, head d] O[row, head d]     (one wave per row; 8 / 16 lanes share a head)
template <int D>
__global__ __launch_bounds__(256) void rowdot_kernel(const f16* __restrict__ a, int lda, const f16* __restrict__ b, int ldb, int64_t rows,
                                                     int heads, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int CH = heads * D / 8;
    for (int ch = lane; ch < (CH + 63) / 64 * 64; ch += 64) {
        float acc = 0.f;
        if (ch < CH) {
            const f16x8 x = *(const f16x8*)(a + r * lda + ch * 8), y = *(const f16x8*)(b + r * ldb + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)x[j] * (float)y[j];
        }
#pragma unroll
        for (int o = 1; o < D / 8; o <<= 1) acc += __shfl_xor(acc, o);
        if (ch < CH && (ch % (D / 8)) == 0) out[r * heads + ch / (D / 8)] = acc;
    }
}

// --------------------------------------------------------------------------------------------------------- pass dQ
template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const f16* __restrict__ q, int ldq, const f16* __restrict__ k, int ldk,
                                                             const f16* __restrict__ v, int ldv, const f16* __restrict__ dout, int ldo,
                                                             const float* __restrict__ lse, const float* __restrict__ dq_dot,
                                                             f16* __restrict__ dq, int lddq, int Sq, int Sk, int nqb, int heads, int ngroups,
                                                             float cexp, float scale) {
    constexpr int PITCH = 2 * D + 32;
    constexpr int NS32 = D / 32, NDB = D / 16, CPR = D / 8;
    constexpr int NCH = (AB_T * CPR + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char ab_smem[];
    char* const Ks = ab_smem;
    char* const Vs = ab_smem + AB_T * PITCH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = (idx / nqb) * 8 + xcd;
    if (grp >= ngroups) return;
    const int qb = idx % nqb, head = grp % heads, bat = grp / heads;
    const int hcol = head * D;
    const f16* const kbase = k + (size_t)bat * Sk * ldk + hcol;
    const f16* const vbase = v + (size_t)bat * Sk * ldv + hcol;
    int qrow = qb * AB_B + wave * 16 + c;
    const bool qok = qrow < Sq;
    if (!qok) qrow = Sq - 1;
    const size_t grow = (size_t)bat * Sq + qrow;
    f16x8 qf[NS32], dof[NS32];
#pragma unroll
    for (int s = 0; s < NS32; ++s) {
        qf[s] = *(const f16x8*)(q + grow * ldq + hcol + 32 * s + 8 * g);
        dof[s] = *(const f16x8*)(dout + grow * ldo + hcol + 32 * s + 8 * g);
    }
    const float L = lse[grow * heads + head], Dq = dq_dot[grow * heads + head];

    f16x8 kreg[NCH], vreg[NCH];
    const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = t + 256 * i, row = id / CPR, cc = id - row * CPR;
            const int key = kt * AB_T + row;
            const bool ok = (AB_T * CPR % 256 == 0 || id < AB_T * CPR) && key < Sk;
            kreg[i] = ok ? *(const f16x8*)(kbase + (size_t)key * ldk + cc * 8) : zero8;
            vreg[i] = ok ? *(const f16x8*)(vbase + (size_t)key * ldv + cc * 8) : zero8;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = t + 256 * i, row = id / CPR, cc = id - row * CPR;
            if (AB_T * CPR % 256 == 0 || id < AB_T * CPR) {
                *(f16x8*)(Ks + row * PITCH + cc * 16) = kreg[i];
                *(f16x8*)(Vs + row * PITCH + cc * 16) = vreg[i];
            }
        }
    };
    f32x4 dqt[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) dqt[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tq = c >> 2, tp = c & 3;
    const char* const kfrag = Ks + c * PITCH + 16 * g;                 // row-major fragment (A of K Q^T)
    const char* const vfragn = Vs + c * PITCH + 16 * g;                // row-major fragment (A of V dO^T)
    const char* const kfragt = Ks + (4 * g + tq) * PITCH + 8 * tp;     // transposed fragment (A of K^T dS^T)
    const int nkt = (Sk + AB_T - 1) / AB_T;
    fetch(0);
    commit();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch(kt + 1);
        f32x4 st[2], dpt[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dpt[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS32; ++s) {
                const f16x8 kf = *(const f16x8*)(kfrag + kb * 16 * PITCH + 64 * s);
                const f16x8 vf = *(const f16x8*)(vfragn + kb * 16 * PITCH + 64 * s);
                st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], st[kb], 0, 0, 0);
                dpt[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, dof[s], dpt[kb], 0, 0, 0);
            }
        }
        f16x8 dsf;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = kt * AB_T + kb * 16 + 4 * g + i;
                const float p = key < Sk ? __builtin_amdgcn_exp2f(st[kb][i] * cexp - L) : 0.f;
                dsf[4 * kb + i] = (f16)(p * (dpt[kb][i] - Dq));
            }
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const f16x4b lo = ab_lds_tr16(kfragt + db * 32);
            const f16x4b hi = ab_lds_tr16(kfragt + 16 * PITCH + db * 32);
            const f16x8 kt8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            dqt[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kt8, dsf, dqt[db], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            commit();
            __syncthreads();
        }
    }
    if (qok) {
        f16* op = dq + grow * lddq + hcol + 4 * g;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const f16x4b o4 = {(f16)(dqt[db][0] * scale), (f16)(dqt[db][1] * scale), (f16)(dqt[db][2] * scale), (f16)(dqt[db][3] * scale)};
            *(f16x4b*)(op + db * 16) = o4;
        }
    }
}

// --------------------------------------------------------------------------------------------------------- pass dK dV
template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const f16* __restrict__ q, int ldq, const f16* __restrict__ k, int ldk,
                                                              const f16* __restrict__ v, int ldv, const f16* __restrict__ dout, int ldo,
                                                              const float* __restrict__ lse, const float* __restrict__ dq_dot,
                                                              f16* __restrict__ dk, f16* __restrict__ dv, int lddk, int Sq, int Sk, int nkb,
                                                              int heads, int ngroups, float cexp, float scale) {
    constexpr int PITCH = 2 * D + 32;
    constexpr int NS32 = D / 32, NDB = D / 16, CPR = D / 8;
    constexpr int NCH = (AB_T * CPR + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char ab_smem[];
    char* const Qs = ab_smem;
    char* const Os = ab_smem + AB_T * PITCH;
    float* const Ls = (float*)(ab_smem + 2 * AB_T * PITCH);            // [32] lse, then [32] Dq of the staged queries
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = (idx / nkb) * 8 + xcd;
    if (grp >= ngroups) return;
    const int kb0 = idx % nkb, head = grp % heads, bat = grp / heads;
    const int hcol = head * D;
    const f16* const qbase = q + (size_t)bat * Sq * ldq + hcol;
    const f16* const obase = dout + (size_t)bat * Sq * ldo + hcol;
    const float* const lbase = lse + (size_t)bat * Sq * heads + head;
    const float* const dbase = dq_dot + (size_t)bat * Sq * heads + head;
    int krow = kb0 * AB_B + wave * 16 + c;
    const bool kok = krow < Sk;
    if (!kok) krow = Sk - 1;
    const size_t gk = (size_t)bat * Sk + krow;
    f16x8 kf[NS32], vf[NS32];
#pragma unroll
    for (int s = 0; s < NS32; ++s) {
        kf[s] = *(const f16x8*)(k + gk * ldk + hcol + 32 * s + 8 * g);
        vf[s] = *(const f16x8*)(v + gk * ldv + hcol + 32 * s + 8 * g);
    }
    f16x8 qreg[NCH], oreg[NCH];
    float lreg = 0.f;
    const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    auto fetch = [&](int qt) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = t + 256 * i, row = id / CPR, cc = id - row * CPR;
            const int qq = qt * AB_T + row;
            const bool ok = (AB_T * CPR % 256 == 0 || id < AB_T * CPR) && qq < Sq;
            qreg[i] = ok ? *(const f16x8*)(qbase + (size_t)qq * ldq + cc * 8) : zero8;
            oreg[i] = ok ? *(const f16x8*)(obase + (size_t)qq * ldo + cc * 8) : zero8;
        }
        if (t < 2 * AB_T) {                                             // threads 0..31: lse, 32..63: Dq of the tile's queries
            const int qq = qt * AB_T + (t & (AB_T - 1));
            lreg = qq < Sq ? (t < AB_T ? lbase : dbase)[(size_t)qq * heads] : 0.f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = t + 256 * i, row = id / CPR, cc = id - row * CPR;
            if (AB_T * CPR % 256 == 0 || id < AB_T * CPR) {
                *(f16x8*)(Qs + row * PITCH + cc * 16) = qreg[i];
                *(f16x8*)(Os + row * PITCH + cc * 16) = oreg[i];
            }
        }
        if (t < 2 * AB_T) Ls[t] = lreg;
    };
    f32x4 dkt[NDB], dvt[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) { dkt[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvt[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int tq = c >> 2, tp = c & 3;
    const char* const qfrag = Qs + c * PITCH + 16 * g;                  // row-major fragments (A of Q K^T, dO V^T)
    const char* const ofrag = Os + c * PITCH + 16 * g;
    const char* const qfragt = Qs + (4 * g + tq) * PITCH + 8 * tp;      // transposed fragments (A of Q^T dS, dO^T P)
    const char* const ofragt = Os + (4 * g + tq) * PITCH + 8 * tp;
    const int nqt = (Sq + AB_T - 1) / AB_T;
    fetch(0);
    commit();
    __syncthreads();
    for (int qt = 0; qt < nqt; ++qt) {
        if (qt + 1 < nqt) fetch(qt + 1);
        f32x4 st[2], dpt[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            st[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dpt[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS32; ++s) {
                const f16x8 qa = *(const f16x8*)(qfrag + qb * 16 * PITCH + 64 * s);
                const f16x8 oa = *(const f16x8*)(ofrag + qb * 16 * PITCH + 64 * s);
                st[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, kf[s], st[qb], 0, 0, 0);        // rows: queries, column: this lane's key
                dpt[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oa, vf[s], dpt[qb], 0, 0, 0);
            }
        }
        f16x8 pf, dsf;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const f32x4 l4 = *(const f32x4*)(Ls + qb * 16 + 4 * g), d4 = *(const f32x4*)(Ls + AB_T + qb * 16 + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qq = qt * AB_T + qb * 16 + 4 * g + i;
                const float p = qq < Sq ? __builtin_amdgcn_exp2f(st[qb][i] * cexp - l4[i]) : 0.f;
                pf[4 * qb + i] = (f16)p;
                dsf[4 * qb + i] = (f16)(p * (dpt[qb][i] - d4[i]));
            }
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const f16x4b olo = ab_lds_tr16(ofragt + db * 32), ohi = ab_lds_tr16(ofragt + 16 * PITCH + db * 32);
            const f16x8 ot8 = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
            dvt[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ot8, pf, dvt[db], 0, 0, 0);
            const f16x4b qlo = ab_lds_tr16(qfragt + db * 32), qhi = ab_lds_tr16(qfragt + 16 * PITCH + db * 32);
            const f16x8 qt8 = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
            dkt[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qt8, dsf, dkt[db], 0, 0, 0);
        }
        __syncthreads();
        if (qt + 1 < nqt) {
            commit();
            __syncthreads();
        }
    }
    if (kok) {
        f16* kp = dk + gk * lddk + hcol + 4 * g;
        f16* vp = dv + gk * lddk + hcol + 4 * g;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const f16x4b k4 = {(f16)(dkt[db][0] * scale), (f16)(dkt[db][1] * scale), (f16)(dkt[db][2] * scale), (f16)(dkt[db][3] * scale)};
            const f16x4b v4 = {(f16)dvt[db][0], (f16)dvt[db][1], (f16)dvt[db][2], (f16)dvt[db][3]};
            *(f16x4b*)(kp + db * 16) = k4;
            *(f16x4b*)(vp + db * 16) = v4;
        }
    }
}


// --------------------------------------------------------------------------------------------------------- temporal attention
// Backward of the attention over the F <= 16 frames of one spatial position (pt_attn_temporal_f16's problem: 14 x 14 scores per
// position and head).  One wave per (clip, position, head), everything in registers and 6-12 KB of LDS: the scores are formed in
// BOTH orientations with v_mfma_f32_16x16x32_f16 straight from the global-memory fragments (rows q / columns k for dK and dV,
// rows k / columns q for dQ), softmax and dS = P (dP - sum_k P dP) on the accumulators, which then are the B operands of
// v_mfma_f32_16x16x16_f16 against Q^T / K^T / dO^T gathered from the staged rows.  Replaces eight pt_gemm_f16 / row-kernel
// launches over 14 400 batch entries of 14 x 14 (0.4 ms per layer) by one pass at the HBM rate.
template <int HDIM>
__global__ __launch_bounds__(256) void attn_temporal_bwd_kernel(const f16* __restrict__ qkv, int ld, int k_off, int v_off, const f16* __restrict__ dout,
                                                                int ldo, f16* __restrict__ dqkv, int ldd, int F, int S, int heads,
                                                                int64_t ntasks, float scale) {
    constexpr int NS = HDIM / 32;
    __shared__ __attribute__((aligned(16))) f16 smem[4 * 3 * 16 * HDIM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const int head = (int)(task % heads);
    const int64_t bs = task / heads;
    const int s = (int)(bs % S);
    const int64_t b = bs / S;
    const int c = lane & 15, g = lane >> 4;
    f16* const Qs = smem + wave * (3 * 16 * HDIM);
    f16* const Ks = Qs + 16 * HDIM;
    f16* const Os = Ks + 16 * HDIM;
    const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    const bool live = c < F;                                              // this lane's frame (as a fragment row) exists
    const int64_t row = (b * F + (live ? c : 0)) * (int64_t)S + s;
    const f16* rp = qkv + row * ld + head * HDIM + g * 8;
    const f16* op = dout + row * ldo + head * HDIM + g * 8;
    f16x8 qf[NS], kf[NS], vf[NS], of[NS];
#pragma unroll
    for (int h = 0; h < NS; ++h) {
        qf[h] = live ? *(const f16x8*)(rp + 32 * h) : zero8;
        kf[h] = live ? *(const f16x8*)(rp + k_off + 32 * h) : zero8;
        vf[h] = live ? *(const f16x8*)(rp + v_off + 32 * h) : zero8;
        of[h] = live ? *(const f16x8*)(op + 32 * h) : zero8;
    }
#pragma unroll
    for (int h = 0; h < NS; ++h) {
        *(f16x8*)(Qs + c * HDIM + 32 * h + 8 * g) = qf[h];
        *(f16x8*)(Ks + c * HDIM + 32 * h + 8 * g) = kf[h];
        *(f16x8*)(Os + c * HDIM + 32 * h + 8 * g) = of[h];
    }
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = sa, sb = sa, db = sa;           // a: rows q = 4g+i, column k = c;  b: rows k = 4g+i, column q = c
#pragma unroll
    for (int h = 0; h < NS; ++h) {
        sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[h], kf[h], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_16x16x32_f16(of[h], vf[h], da, 0, 0, 0);
        sb = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[h], qf[h], sb, 0, 0, 0);
        db = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[h], of[h], db, 0, 0, 0);
    }
    // ---- orientation a: softmax over the keys = across the 16 lanes of a row group
    f16x4 pa, dsa;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = live ? sa[i] * scale : -INFINITY, m = v;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        float e = live ? __expf(v - m) : 0.f, z = e;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) z += __shfl_xor(z, o);
        const float p = (4 * g + i < F) ? e / z : 0.f;
        float dd = p * da[i];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) dd += __shfl_xor(dd, o);
        pa[i] = (f16)p;
        dsa[i] = (f16)(p * (da[i] - dd) * scale);
    }
    // ---- orientation b: softmax over the keys = over the rows (in-lane i, then the four row groups)
    f16x4 dsb;
    {
        float v[4], m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (4 * g + i < F) ? sb[i] * scale : -INFINITY; m = fmaxf(m, v[i]); }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float e[4], z = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { e[i] = (4 * g + i < F) ? __expf(v[i] - m) : 0.f; z += e[i]; }
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        float dd = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { e[i] = live ? e[i] / z : 0.f; dd += e[i] * db[i]; }
        dd += __shfl_xor(dd, 16);
        dd += __shfl_xor(dd, 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) dsb[i] = (f16)(e[i] * (db[i] - dd) * scale);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                                   // this wave's staged rows are in LDS
    __builtin_amdgcn_wave_barrier();
    f16* const gq = dqkv + row * ldd + head * HDIM + 4 * g;
#pragma unroll
    for (int blk = 0; blk < HDIM / 16; ++blk) {
        f16x4 ot, qt, kt;                                                 // A operands: (row d = 16 blk + c, k index = frame 4g + j)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ot[j] = Os[(4 * g + j) * HDIM + 16 * blk + c];
            qt[j] = Qs[(4 * g + j) * HDIM + 16 * blk + c];
            kt[j] = Ks[(4 * g + j) * HDIM + 16 * blk + c];
        }
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x16f16(ot, pa, z4, 0, 0, 0);      // dV^T[d][k] = sum_q dO^T[d][q] P[q][k]
        const f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x16f16(qt, dsa, z4, 0, 0, 0);     // dK^T[d][k] = sum_q Q^T[d][q] dS[q][k]
        const f32x4 dq = __builtin_amdgcn_mfma_f32_16x16x16f16(kt, dsb, z4, 0, 0, 0);     // dQ^T[d][q] = sum_k K^T[d][k] dS^T[k][q]
        if (live) {                                                       // column c = this lane's frame; rows d = 16 blk + 4 g + i
            *(f16x4*)(gq + 16 * blk) = (f16x4){(f16)dq[0], (f16)dq[1], (f16)dq[2], (f16)dq[3]};
            *(f16x4*)(gq + k_off + 16 * blk) = (f16x4){(f16)dk[0], (f16)dk[1], (f16)dk[2], (f16)dk[3]};
            *(f16x4*)(gq + v_off + 16 * blk) = (f16x4){(f16)dv[0], (f16)dv[1], (f16)dv[2], (f16)dv[3]};
        }
    }
}

template <int D>
int launch_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* out, int ldout, const void* dout, int ldo,
               const float* lse, float* dq_dot, void* dq, void* dk, void* dv, int ldd, int nbatch, int S, int heads, float scale, hipStream_t s) {
    constexpr int PITCH = 2 * D + 32;
    constexpr int LDS_Q = 2 * AB_T * PITCH, LDS_K = 2 * AB_T * PITCH + 2 * AB_T * (int)sizeof(float);
    static bool attr_done[64] = {};
    const int dev = pt_device();
    if (!attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_Q);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_K);
        attr_done[dev] = true;
    }
    const int64_t rows = (int64_t)nbatch * S;
    hipLaunchKernelGGL(rowdot_kernel<D>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const f16*)dout, ldo, (const f16*)out, ldout, rows, heads,
                       dq_dot);
    const int nb = (S + AB_B - 1) / AB_B;
    const long long ngroups = (long long)nbatch * heads, nblk = (ngroups + 7) / 8 * 8 * nb;
    PT_CHECK(nblk < (1ll << 31), "pt_attn_bwd_f16: grid too large");
    const float cexp = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, dim3((unsigned)nblk), dim3(256), LDS_Q, s, (const f16*)q, ldq, (const f16*)k, ldk, (const f16*)v, ldv,
                       (const f16*)dout, ldo, lse, (const float*)dq_dot, (f16*)dq, ldd, S, S, nb, heads, (int)ngroups, cexp, scale);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<D>, dim3((unsigned)nblk), dim3(256), LDS_K, s, (const f16*)q, ldq, (const f16*)k, ldk, (const f16*)v, ldv,
                       (const f16*)dout, ldo, lse, (const float*)dq_dot, (f16*)dk, (f16*)dv, ldd, S, S, nb, heads, (int)ngroups, cexp, scale);
    return 0;
}

}  // namespace

extern "C" int pt_attn_bwd_f16(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, const void* out,
                               int32_t ldout, const void* dout, int32_t ldo, const float* lse, float* dq_dot, void* dq, void* dk, void* dv,
                               int32_t ldd, int32_t nbatch, int32_t S, int32_t heads, int32_t head_dim, float scale, void* stream) {
    PT_CHECK(q && k && v && out && dout && lse && dq_dot && dq && dk && dv, "pt_attn_bwd_f16: null pointer");
    PT_CHECK(nbatch > 0 && S > 0 && heads > 0 && scale > 0.f, "pt_attn_bwd_f16: bad sizes");
    PT_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && ldout % 8 == 0 && ldd % 4 == 0, "pt_attn_bwd_f16: pitches must be multiples of 8 (gradients: 4)");
    auto al = [](const void* p, int a) { return ((uintptr_t)p & (a - 1)) == 0; };
    PT_CHECK(al(q, 16) && al(k, 16) && al(v, 16) && al(out, 16) && al(dout, 16) && al(dq, 8) && al(dk, 8) && al(dv, 8), "pt_attn_bwd_f16: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (head_dim) {
        case 64:  rc = launch_bwd<64>(q, ldq, k, ldk, v, ldv, out, ldout, dout, ldo, lse, dq_dot, dq, dk, dv, ldd, nbatch, S, heads, scale, s); break;
        case 128: rc = launch_bwd<128>(q, ldq, k, ldk, v, ldv, out, ldout, dout, ldo, lse, dq_dot, dq, dk, dv, ldd, nbatch, S, heads, scale, s); break;
        default:
            PT_CHECK(false, "pt_attn_bwd_f16: head_dim %d unsupported (64, 128)", head_dim);
    }
    if (rc) return rc;
    PT_LAUNCH_CHECK("pt_attn_bwd_f16");
    return 0;
}

extern "C" int pt_attn_temporal_bwd_f16(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, const void* dout, int32_t ldo, void* dqkv,
                                        int32_t ldd, int32_t B, int32_t F, int32_t S, int32_t heads, int32_t head_dim, float scale, void* stream) {
    PT_CHECK(qkv && dout && dqkv, "pt_attn_temporal_bwd_f16: null pointer");
    PT_CHECK(head_dim == 64 || head_dim == 128, "pt_attn_temporal_bwd_f16: head_dim %d unsupported (64, 128)", head_dim);
    PT_CHECK(F > 0 && F <= 16, "pt_attn_temporal_bwd_f16: %d frames (at most 16; longer clips go through pt_gemm_f16)", F);
    PT_CHECK(ld % 8 == 0 && ldo % 8 == 0 && ldd % 4 == 0 && k_off % 8 == 0 && v_off % 8 == 0, "pt_attn_temporal_bwd_f16: pitches / offsets must be multiples of 8 (gradient pitch: 4)");
    PT_CHECK((((uintptr_t)qkv | (uintptr_t)dout) & 15) == 0 && ((uintptr_t)dqkv & 7) == 0, "pt_attn_temporal_bwd_f16: misaligned pointer");
    PT_CHECK(B > 0 && S > 0 && heads > 0 && scale > 0.f, "pt_attn_temporal_bwd_f16: bad sizes");
    const int64_t ntasks = (int64_t)B * S * heads;
    const int64_t blocks = (ntasks + 3) / 4;
    PT_CHECK(blocks < (1ll << 31), "pt_attn_temporal_bwd_f16: grid too large");
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 64)
        hipLaunchKernelGGL(attn_temporal_bwd_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, s, (const f16*)qkv, ld, k_off, v_off, (const f16*)dout, ldo,
                           (f16*)dqkv, ldd, F, S, heads, ntasks, scale);
    else
        hipLaunchKernelGGL(attn_temporal_bwd_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, s, (const f16*)qkv, ld, k_off, v_off, (const f16*)dout, ldo,
                           (f16*)dqkv, ldd, F, S, heads, ntasks, scale);
    PT_LAUNCH_CHECK("pt_attn_temporal_bwd_f16");
    return 0;
}
