// Implicit-GEMM convolution / linear layer for gfx950 (MI355X).
//
//   D[n, m] = sum_k W[n, k] * A[m, k]       (computed transposed so each lane ends up with 4 consecutive
//                                            output channels of one pixel -> row-contiguous epilogue)
//   tile 128 (pixels) x 128 (channels) x 64 (k), 256 threads = 4 waves in a 2 x 2 grid, 64 x 64 per wave,
//   v_mfma_f32_16x16x32_f16, fp32 accumulate.
//   Both operands are K-contiguous in memory (channels-last activations, [N, K] packed weights), so both tiles
//   are staged with 16-byte LDS-DMA (global_load_lds_dwordx4) straight from a per-lane gathered source address:
//   im2col, zero padding (padded taps read a zero page), the 2-source skip concatenation and the nearest-2x
//   upsampling all live in that address computation and cost no extra pass over HBM.
//   LDS image: rows of 128 B (64 halfs), the eight 16-B chunks of row r XOR-swizzled by (r >> 1) & 7 on the
//   SOURCE side (the DMA destination is lane-linear), undone in the ds_read_b128 address: conflict-free reads.
//   Two LDS stages; the DMA of tile k+1 is in flight while the MFMAs of tile k run.
//   Epilogue: bias (+GEGLU) on the accumulators, tile transposed through LDS (fp32, padded rows), then
//   residual / broadcast row vector / AlphaBlender lerp / scale on 16-byte rows, 16-byte stores.
#include "pt_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int TILE_BYTES = BM * BK * 2;             // 16 KiB per operand tile
constexpr int EPI_LD = 68;                          // floats per staged row (64 + 4 pad)
constexpr int EPI_WAVE_BYTES = 64 * EPI_LD * 4;     // 17408
constexpr int SMEM_BYTES = 4 * EPI_WAVE_BYTES;      // 69632 >= 4 * TILE_BYTES

struct KParams {
    pt_igemm_params p;
    const f16* zeros;
    int tiles_m, tiles_n;
    int vec_ok;     // 16-byte epilogue path allowed
};

__device__ __forceinline__ int vec_index(const pt_igemm_params& p, int m) {
    if (p.vec_mode == 1) return m / p.vG;
    return ((m / p.vFS) * p.vS + m % p.vS) % p.vB;
}

template <bool FAST>
__global__ __launch_bounds__(NT, 2) void igemm_kernel(const KParams kp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const pt_igemm_params& p = kp.p;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bid = pt_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = bid % kp.tiles_n, tile_m = bid / kp.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---------------- staging set-up: this thread copies chunk slot (t + 256 i), i = 0..3, of each tile
    const int cphys = t & 7;
    const int csrc = cphys ^ ((t >> 4) & 7);                 // source chunk (row parity bits are i-independent)
    const int Ctot = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    int iy0[4], ix0[4], pix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (t >> 3) + 32 * i;
        if (m < p.M) {
            const int img = m / HWo, rem = m - img * HWo;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            iy0[i] = oy * p.stride - p.pad_h;
            ix0[i] = ox * p.stride - p.pad_w;
            pix0[i] = img * p.Hin * p.Win;
        } else {
            iy0[i] = -(1 << 28); ix0[i] = 0; pix0[i] = 0;
        }
    }
    const int Hlim = p.upsample2x ? 2 * p.Hin : p.Hin, Wlim = p.upsample2x ? 2 * p.Win : p.Win;
    const f16* zsrc = kp.zeros + (lane & 7) * 8;
    const f16* wsrc = (const f16*)p.w + (size_t)(n0 + (t >> 3)) * p.Kpad + csrc * 8;
    const size_t wrow32 = (size_t)32 * p.Kpad;
    const int nk = p.Kpad / BK;

    auto stage = [&](int kt, int buf) {
        char* As = smem + buf * 2 * TILE_BYTES;
        char* Bs = As + TILE_BYTES;
        const f16* src; int ld, cofs, ky, kx; bool kvalid = true;
        if (FAST) {
            const int k0 = kt * BK;
            const int tap = k0 / Ctot, ci0 = k0 - tap * Ctot;
            ky = tap / p.KW; kx = tap - ky * p.KW;
            if (ci0 < p.C0) { src = (const f16*)p.x0; ld = p.ld0; cofs = ci0 + csrc * 8; }
            else            { src = (const f16*)p.x1; ld = p.ld1; cofs = ci0 - p.C0 + csrc * 8; }
        } else {
            const int kg = kt * BK + csrc * 8;
            kvalid = kg < p.K;
            const int tap = kg / Ctot, ci = kg - tap * Ctot;
            ky = tap / p.KW; kx = tap - ky * p.KW;
            if (ci < p.C0) { src = (const f16*)p.x0; ld = p.ld0; cofs = ci; }
            else           { src = (const f16*)p.x1; ld = p.ld1; cofs = ci - p.C0; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = kvalid && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
            if (p.upsample2x) { iy >>= 1; ix >>= 1; }
            const f16* g = ok ? src + ((size_t)(pix0[i] + iy * p.Win + ix) * ld + cofs) : zsrc;
            pt_glds16(g, As + (wave * 64 + 256 * i) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            pt_glds16(wsrc + i * wrow32 + (size_t)kt * BK, Bs + (wave * 64 + 256 * i) * 16);
    };

    // ---------------- MFMA set-up
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int swz = frow >> 1;                               // (row >> 1) & 7 for every fragment row of this lane
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* As = smem + buf * 2 * TILE_BYTES;
        const char* Bs = As + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((fq + 4 * ks) ^ swz) * 16;
            f16x8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[i] = *(const f16x8*)(Bs + (wc * 64 + i * 16 + frow) * 128 + coff);
                xf[i] = *(const f16x8*)(As + (wr * 64 + i * 16 + frow) * 128 + coff);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
    };

    // ---------------- main loop: DMA of tile k+1 in flight under the MFMAs of tile k
    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk - 1; ++kt) {
        stage(kt + 1, cur ^ 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // ---------------- epilogue 1: bias (+ GEGLU) on the accumulators
    const f16* bias = (const f16*)p.bias;
    const int nbase = n0 + wc * 64 + 4 * fq;
    if (bias) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f16x4 b4 = *(const f16x4*)(bias + nbase + ni * 16);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ni][mi][j] += (float)b4[j];
        }
    }
    int ntl = 4;                                             // valid 16-wide column tiles of this wave
    if (p.act == 1) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[pr][mi][j] = acc[2 * pr][mi][j] * pt_gelu_erf(acc[2 * pr + 1][mi][j]);
        ntl = 2;
    } else if (p.act == 2) {                                 // SiLU (condition encoder, controlnet_sdv.py:101-106)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ni][mi][j] = pt_silu(acc[ni][mi][j]);
    }

    // ---------------- epilogue 2: transpose through LDS (each wave its own region)
    __syncthreads();                                         // every wave is done with the operand tiles
    float* E = (float*)(smem + wave * EPI_WAVE_BYTES);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        if (ni < ntl) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                *(f32x4*)(E + (mi * 16 + frow) * EPI_LD + ni * 16 + 4 * fq) = acc[ni][mi];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this wave's LDS writes have landed

    // ---------------- epilogue 3: row-contiguous fused tail
    const int ncols = ntl * 16;                              // 64 or 32 output columns per wave
    const int lpr = ncols / 8;                               // lanes per row
    const int rpp = 64 / lpr;                                // rows per pass
    const int Nout = p.act == 1 ? p.N / 2 : p.N;
    const int col0 = (p.act == 1 ? (n0 + wc * 64) / 2 : n0 + wc * 64) + (lane % lpr) * 8;
    const float alpha = p.alpha, oscale = p.out_scale;
    f16* out = (f16*)p.out;
    const f16* res = (const f16*)p.res;
    const f16* vec = (const f16*)p.vec;
    const f16* blend = (const f16*)p.blend;
    if (col0 < Nout) {
        for (int r = lane / lpr; r < 64; r += rpp) {
            const int m = m0 + wr * 64 + r;
            if (m >= p.M) break;
            const float* e = E + r * EPI_LD + (lane % lpr) * 8;
            const f32x4 v0 = *(const f32x4*)e, v1 = *(const f32x4*)(e + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (kp.vec_ok && col0 + 8 <= Nout) {
                if (res) {
                    const f16x8 r8 = *(const f16x8*)(res + (size_t)m * p.ldr + col0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)r8[j];
                }
                if (vec) {
                    const f16x8 r8 = *(const f16x8*)(vec + (size_t)vec_index(p, m) * p.ldv + col0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)r8[j];
                }
                if (blend) {
                    const f16x8 r8 = *(const f16x8*)(blend + (size_t)m * p.ldb + col0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = alpha * (float)r8[j] + (1.0f - alpha) * v[j];
                }
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)(v[j] * oscale);
                *(f16x8*)(out + (size_t)m * p.ldo + col0) = o;
            } else {
                for (int j = 0; j < 8 && col0 + j < Nout; ++j) {
                    float x = v[j];
                    if (res) x += (float)res[(size_t)m * p.ldr + col0 + j];
                    if (vec) x += (float)vec[(size_t)vec_index(p, m) * p.ldv + col0 + j];
                    if (blend) x = alpha * (float)blend[(size_t)m * p.ldb + col0 + j] + (1.0f - alpha) * x;
                    out[(size_t)m * p.ldo + col0 + j] = (f16)(x * oscale);
                }
            }
        }
    }
}

}  // namespace

extern "C" int pt_igemm_f16(const pt_igemm_params* pp, void* stream) {
    const pt_igemm_params& p = *pp;
    const int Ctot = p.C0 + p.C1;
    PT_CHECK(p.x0 && p.w && p.out, "pt_igemm_f16: null pointer");
    PT_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "pt_igemm_f16: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    PT_CHECK(p.C0 > 0 && p.C0 % 8 == 0 && p.C1 % 8 == 0, "pt_igemm_f16: channel counts must be multiples of 8 (C0=%d C1=%d)", p.C0, p.C1);
    PT_CHECK((p.C1 == 0) == (p.x1 == nullptr), "pt_igemm_f16: x1/C1 mismatch");
    PT_CHECK(p.K == p.KH * p.KW * Ctot, "pt_igemm_f16: K=%d != KH*KW*(C0+C1)=%d", p.K, p.KH * p.KW * Ctot);
    PT_CHECK(p.Kpad % BK == 0 && p.Kpad >= p.K, "pt_igemm_f16: Kpad=%d must be a multiple of 64 and >= K=%d", p.Kpad, p.K);
    PT_CHECK((long long)p.Nimg * p.Hout * p.Wout == p.M, "pt_igemm_f16: M=%d != Nimg*Hout*Wout", p.M);
    PT_CHECK(p.ld0 % 8 == 0 && p.ld1 % 8 == 0, "pt_igemm_f16: source pitches must be multiples of 8");
    PT_CHECK(p.stride == 1 || p.stride == 2, "pt_igemm_f16: stride %d", p.stride);
    PT_CHECK(!(p.upsample2x && p.stride != 1), "pt_igemm_f16: upsample2x needs stride 1");
    PT_CHECK(p.act == 0 || p.act == 2 || (p.act == 1 && p.N % 32 == 0), "pt_igemm_f16: act must be 0, 1 (GEGLU, N %% 32 == 0) or 2 (SiLU)");
    PT_CHECK(p.vec_mode == 0 || p.vec, "pt_igemm_f16: vec_mode without vec");
    PT_CHECK(pt_zero_page(), "pt_igemm_f16: zero page not set (pt_set_zero_page)");
    KParams kp;
    kp.p = p;
    if (!p.vec) kp.p.vec_mode = 0;
    kp.zeros = (const f16*)pt_zero_page();
    kp.tiles_m = (p.M + BM - 1) / BM;
    kp.tiles_n = (p.N + BN - 1) / BN;
    const int nout = p.act == 1 ? p.N / 2 : p.N;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    kp.vec_ok = (nout % 8 == 0) && (p.ldo % 8 == 0) && al16(p.out) && (!p.res || (p.ldr % 8 == 0 && al16(p.res))) &&
                (!p.vec || (p.ldv % 8 == 0 && al16(p.vec))) && (!p.blend || (p.ldb % 8 == 0 && al16(p.blend)));
    const bool fast = (Ctot % BK == 0) && (p.C0 % BK == 0) && (p.Kpad == p.K);
    const long long nblk = (long long)kp.tiles_m * kp.tiles_n;
    PT_CHECK(nblk < (1ll << 31), "pt_igemm_f16: grid too large");
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        (void)hipFuncSetAttribute((const void*)igemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        attr_done = true;
    }
    pt_prof_begin(0, s, 2.0 * (double)p.M * (double)p.N * (double)p.K);
    if (fast) hipLaunchKernelGGL(igemm_kernel<true>, dim3((unsigned)nblk), dim3(NT), SMEM_BYTES, s, kp);
    else      hipLaunchKernelGGL(igemm_kernel<false>, dim3((unsigned)nblk), dim3(NT), SMEM_BYTES, s, kp);
    pt_prof_end(0, s);
    PT_LAUNCH_CHECK("pt_igemm_f16");
    return 0;
}
