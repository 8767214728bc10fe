// Implicit-GEMM convolution / linear layer for gfx950 (MI355X).
//
//   D[n, m] = sum_k W[n, k] * A[m, k]       (computed transposed so each lane ends up with 4 consecutive
//                                            output channels of one pixel -> row-contiguous epilogue)
//   v_mfma_f32_16x16x32_f16, fp32 accumulate, K step 64.  Three tile configurations (waves WM x WN, per-wave
//   tile TM x TN of 16 x 16 accumulators), chosen per shape by choose_cfg():
//       256 x 256  (8 waves 4 x 2, 64 x 128 per wave)   the large-N linears / convs
//       128 x 320  (8 waves 2 x 4,  64 x  80 per wave)  N = 320 / 960 / ...: SVD's level-0 width without padding waste
//       128 x 128  (4 waves 2 x 2,  64 x 64 per wave)   small or ragged problems, 2 workgroups per CU
//   Both operands are K-contiguous in memory (channels-last activations, [N, K] packed weights), so both tiles
//   are staged with 16-byte LDS-DMA (global_load_lds_dwordx4) straight from a per-lane gathered source address:
//   im2col, zero padding (padded taps read a zero page), the 2-source skip concatenation and the nearest-2x
//   upsampling all live in that address computation and cost no extra pass over HBM.
//   LDS image: rows of 128 B (64 halfs), the eight 16-B chunks of row r XOR-swizzled by (r >> 1) & 7 on the
//   SOURCE side (the DMA destination is lane-linear), undone in the ds_read_b128 address: conflict-free reads.
//   Two LDS stages; the DMA of tile k+1 is in flight while the MFMAs of tile k run.
//   Epilogue: bias (+GEGLU / SiLU) on the accumulators, then the wave's full width transposed through LDS in row
//   chunks (fp32, padded rows) and finished row-wise in >= 128-byte segments: residual / broadcast row vector / AlphaBlender lerp / scale, 16-byte loads+stores.
#include "pt_common.h"

namespace {

constexpr int BK = 64;

struct KParams {
    pt_igemm_params p;
    const f16* zeros;
    int tiles_m, tiles_n;
    int npad;       // rows of the packed weight image
    int vec_ok;     // 16-byte epilogue path allowed
    int gm;         // M tiles per rasterisation group (see the kernel's tile-order comment)
};

template <int WM_, int WN_, int TM_, int TN_>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = WM * WN * 64;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_SLOTS = BM * 8 / NT, B_SLOTS = BN * 8 / NT;
    static constexpr int EPI_LD = TN * 16 + 4;                 // floats per staged row: the wave's width + 4 pad
    static constexpr int EPI_RH = (WM * WN * TM * 16 * EPI_LD * 4 <= 144 * 1024) ? TM * 16 : ((WM * WN * TM * 8 * EPI_LD * 4 <= 144 * 1024) ? TM * 8 : TM * 4);
    static constexpr int EPI_WAVE_BYTES = EPI_RH * EPI_LD * 4;
    static constexpr int SMEM = (2 * STAGE > WM * WN * EPI_WAVE_BYTES) ? 2 * STAGE : WM * WN * EPI_WAVE_BYTES;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile rows must divide over the threads");
    static_assert((NT / 16) % 8 == 0, "row swizzle must be slot-group independent");
};

__device__ __forceinline__ int vec_index(const pt_igemm_params& p, int m) {
    if (p.vec_mode == 1) return m / p.vG;
    return ((m / p.vFS) * p.vS + m % p.vS) % p.vB;
}

template <class CF, bool FAST>
__global__ __launch_bounds__(CF::NT, 2) void igemm_kernel(const KParams kp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TM = CF::TM, TN = CF::TN, NT = CF::NT, BM = CF::BM, BN = CF::BN;
    const pt_igemm_params& p = kp.p;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // Tile order: XCD-contiguous chunks (pt_xcd_remap), and inside them "grouped" rasterisation - kp.gm consecutive
    // M tiles share one N tile (one weight panel) before the next N tile starts, so the ~32 workgroups an XCD runs at
    // once form a gm x (32 / gm) block of the tile grid and both operand panels are shared through its L2.  The plain
    // N-fastest order re-streamed the whole weight matrix from beyond L2 once per M-tile row (FETCH_SIZE: 26x the
    // algorithmic bytes on the 16128 x 10240 x 1280 GEGLU projection).
    const int bid = pt_xcd_remap(blockIdx.x, gridDim.x);
    const int gsz = kp.gm * kp.tiles_n, grp = bid / gsz, first_m = grp * kp.gm;
    const int gm = min(kp.gm, kp.tiles_m - first_m), within = bid - grp * gsz;
    const int tile_m = first_m + within % gm, tile_n = within / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---------------- staging set-up: this thread copies chunk slot (t + NT i) of each tile
    const int csrc = (t & 7) ^ ((t >> 4) & 7);               // source chunk (row parity bits are i-independent)
    const int Ctot = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    int iyx[CF::A_SLOTS], pix0[CF::A_SLOTS];                  // (iy0 << 16 | ix0 & 0xffff): top-left tap of the output pixel
#pragma unroll
    for (int i = 0; i < CF::A_SLOTS; ++i) {
        const int m = m0 + (t >> 3) + (NT / 8) * i;
        if (m < p.M) {
            const int img = m / HWo, rem = m - img * HWo;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            iyx[i] = ((oy * p.stride - p.pad_h) << 16) | ((ox * p.stride - p.pad_w) & 0xffff);
            pix0[i] = img * p.Hin * p.Win;
        } else {
            iyx[i] = (int)0xC0000000; pix0[i] = 0;            // iy0 = -16384: every tap is out of bounds
        }
    }
    const int Hlim = p.upsample2x ? 2 * p.Hin : p.Hin, Wlim = p.upsample2x ? 2 * p.Win : p.Win;
    const f16* zsrc = kp.zeros + (lane & 7) * 8;
    int woff[CF::B_SLOTS];
#pragma unroll
    for (int i = 0; i < CF::B_SLOTS; ++i) {
        int wrow = n0 + (t >> 3) + (NT / 8) * i;
        if (wrow >= kp.npad) wrow = kp.npad - 1;             // columns >= N are never stored; keep the read in bounds
        woff[i] = wrow * p.Kpad + csrc * 8;
    }
    const f16* wbase = (const f16*)p.w;
    const int nk = p.Kpad / BK;

    // FAST path (every 64-wide K tile lies inside one tap of one source): the im2col gather is kept as one source
    // pointer per slot, recomputed only when the tap or the source changes and otherwise just advanced by the channel
    // offset - the full address arithmetic per copy (~25 VALU) was costing as many issue cycles as the MFMAs.
    const f16* aptr[CF::A_SLOTS];
    unsigned avalid = 0;
    int s_tap = 0, s_src = 0, s_ci = 0;                      // wave-uniform position of the next tile to stage
    auto retarget = [&]() {
        const int ky = s_tap / p.KW, kx = s_tap - ky * p.KW;
        const f16* src = s_src ? (const f16*)p.x1 : (const f16*)p.x0;
        const int ld = s_src ? p.ld1 : p.ld0;
        avalid = 0;
#pragma unroll
        for (int i = 0; i < CF::A_SLOTS; ++i) {
            int iy = (iyx[i] >> 16) + ky, ix = (int)(short)(iyx[i] & 0xffff) + kx;
            const bool ok = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
            if (p.upsample2x) { iy >>= 1; ix >>= 1; }
            aptr[i] = ok ? src + ((size_t)(pix0[i] + iy * p.Win + ix) * ld + csrc * 8) : zsrc;
            avalid |= ok ? (1u << i) : 0u;
        }
    };

    auto stage = [&](int kt, int buf) {
        char* As = smem + buf * CF::STAGE;
        char* Bs = As + CF::A_BYTES;
        if (FAST) {
            if (s_ci == 0) retarget();
#pragma unroll
            for (int i = 0; i < CF::A_SLOTS; ++i)
                pt_glds16(aptr[i] + (((avalid >> i) & 1u) ? s_ci : 0), As + (wave * 64 + NT * i) * 16);
            s_ci += BK;
            if (s_ci == (s_src ? p.C1 : p.C0)) {
                s_ci = 0;
                if (s_src == 0 && p.C1 > 0) s_src = 1;
                else { s_src = 0; ++s_tap; }
            }
        } else {
            const int kg = kt * BK + csrc * 8;
            const bool kvalid = kg < p.K;
            const int tap = kg / Ctot, ci = kg - tap * Ctot;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const f16* src; int ld, cofs;
            if (ci < p.C0) { src = (const f16*)p.x0; ld = p.ld0; cofs = ci; }
            else           { src = (const f16*)p.x1; ld = p.ld1; cofs = ci - p.C0; }
#pragma unroll
            for (int i = 0; i < CF::A_SLOTS; ++i) {
                int iy = (iyx[i] >> 16) + ky, ix = (int)(short)(iyx[i] & 0xffff) + kx;
                const bool ok = kvalid && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
                if (p.upsample2x) { iy >>= 1; ix >>= 1; }
                const f16* g = ok ? src + ((size_t)(pix0[i] + iy * p.Win + ix) * ld + cofs) : zsrc;
                pt_glds16(g, As + (wave * 64 + NT * i) * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < CF::B_SLOTS; ++i)
            pt_glds16(wbase + (woff[i] + kt * BK), Bs + (wave * 64 + NT * i) * 16);
    };

    // ---------------- MFMA set-up
    const int wr = wave / CF::WN, wc = wave % CF::WN;
    const int frow = lane & 15, fq = lane >> 4;
    const int swz = frow >> 1;                               // (row >> 1) & 7 for every fragment row of this lane
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Tiles with register headroom (DB) issue the ds_reads of BOTH 32-deep fragment sets before the first MFMA (the
    // sched_barrier keeps the compiler from sinking the second set next to its use): the second set's LDS latency -
    // ~300 cycles with all 8 waves reading at once - then hides under the first set's MFMAs.
    constexpr bool DB = (TM * TN * 4 + 2 * (TM + TN) * 4) <= 190;
    auto compute = [&](int buf) {
        const char* As = smem + buf * CF::STAGE + (wr * TM * 16 + frow) * 128;
        const char* Bs = smem + buf * CF::STAGE + CF::A_BYTES + (wc * TN * 16 + frow) * 128;
        if (DB) {
            f16x8 xf[2][TM], wf[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int coff = ((fq + 4 * ks) ^ swz) * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) xf[ks][i] = *(const f16x8*)(As + i * 2048 + coff);
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[ks][i] = *(const f16x8*)(Bs + i * 2048 + coff);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][ni], xf[ks][mi], acc[ni][mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int coff = ((fq + 4 * ks) ^ swz) * 16;
                f16x8 xf[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) xf[i] = *(const f16x8*)(As + i * 2048 + coff);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const f16x8 wf = *(const f16x8*)(Bs + ni * 2048 + coff);
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mi], acc[ni][mi], 0, 0, 0);
                }
            }
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

    // ---------------- epilogue 1: bias (+ GEGLU / SiLU) on the accumulators
    const f16* bias = (const f16*)p.bias;
    if (bias) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            int nb = n0 + (wc * TN + ni) * 16 + 4 * fq;
            if (nb > kp.npad - 4) nb = kp.npad - 4;
            const f16x4 b4 = *(const f16x4*)(bias + nb);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ni][mi][j] += (float)b4[j];
        }
    }
    int ntl = TN;                                            // valid 16-wide column tiles of this wave
    if (p.act == 1) {
#pragma unroll
        for (int pr = 0; pr < TN / 2; ++pr)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[pr][mi][j] = acc[2 * pr][mi][j] * pt_gelu_erf(acc[2 * pr + 1][mi][j]);
        ntl = TN / 2;
    } else if (p.act == 2) {                                 // SiLU (condition encoder, controlnet_sdv.py:101-106)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ni][mi][j] = pt_silu(acc[ni][mi][j]);
    }

    // ---------------- epilogue 2: the wave's full width, RH rows at a time, through LDS; row-wise fused tail.
    // Every row segment written is >= 128 contiguous bytes (64 for the GEGLU half-width of the 128-wide tiles).
    __syncthreads();                                         // every wave is done with the operand tiles
    constexpr int RH = CF::EPI_RH, ELD = CF::EPI_LD;
    float* E = (float*)(smem + wave * CF::EPI_WAVE_BYTES);
    const int Nout = p.act == 1 ? p.N / 2 : p.N;
    const int wcol0 = p.act == 1 ? (n0 + wc * TN * 16) / 2 : n0 + wc * TN * 16;
    const float alpha = p.alpha, oscale = p.out_scale;
    f16* out = (f16*)p.out;
    const f16* res = (const f16*)p.res;
    const f16* vec = (const f16*)p.vec;
    const f16* blend = (const f16*)p.blend;
    const int lpr = ntl * 2;                                 // lanes per row, 8 columns each
    const int rpp = 64 / lpr;                                // rows per pass (lanes >= rpp * lpr idle)
    const int lrow = lane / lpr, lcol = (lane - lrow * lpr) * 8;
    const int col0 = wcol0 + lcol;
#pragma unroll
    for (int rc = 0; rc < TM * 16 / RH; ++rc) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            if (ni < ntl) {
#pragma unroll
                for (int mi = 0; mi < RH / 16; ++mi)
                    *(f32x4*)(E + (mi * 16 + frow) * ELD + ni * 16 + 4 * fq) = acc[ni][rc * (RH / 16) + mi];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's LDS writes have landed
        if (lrow < rpp && col0 < Nout) {
            for (int r = lrow; r < RH; r += rpp) {
                const int m = m0 + wr * TM * 16 + rc * RH + r;
                if (m >= p.M) break;
                const float* e = E + r * ELD + lcol;
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
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // reads done before the next chunk overwrites E
    }
}

using CfgBig = Cfg<4, 2, 4, 8>;     // 256 x 256: 64 x 128 per wave
using CfgW320 = Cfg<2, 4, 4, 5>;    // 128 x 320 (never used with GEGLU: odd TN)
using CfgSmall = Cfg<2, 2, 4, 4>;   // 128 x 128

// Pick the tile configuration: useful flops / (machine time in units of a full wave of tiles).
int choose_cfg(int M, int N, int act) {
    struct Opt { int bm, bn, slots; double speed; };
    static const Opt opts[3] = {{256, 256, 256, 1.0}, {128, 320, 256, 1.0}, {128, 128, 512, 0.92}};   // speeds: profiles/r01/igemm_cfg_sweep_v11.txt
    int best = 2; double best_t = 1e300;
    for (int i = 0; i < 3; ++i) {
        if (i == 1 && act == 1) continue;
        const double tiles = (double)((M + opts[i].bm - 1) / opts[i].bm) * ((N + opts[i].bn - 1) / opts[i].bn);
        const double waves = (double)(long long)((tiles + opts[i].slots - 1) / opts[i].slots);
        const double tcost = waves * opts[i].slots * (double)opts[i].bm * opts[i].bn / opts[i].speed;
        if (tcost < best_t * 0.999) { best_t = tcost; best = i; }
    }
    return best;
}

// Rasterisation group size: minimise the modelled beyond-L2 traffic  A * ceil(tiles_n / n_c) + W * ceil(tiles_m / gm),
// where the `conc` workgroups co-resident on one XCD cover gm M tiles x n_c = conc / gm N tiles.
int choose_group(int tiles_m, int tiles_n, double a_bytes, double w_bytes, int conc) {
    int best = 1; double best_c = 1e300;
    for (int gm = 1; gm <= conc; gm *= 2) {
        const int g = gm < tiles_m ? gm : tiles_m;
        int nc = conc / g; if (nc > tiles_n) nc = tiles_n; if (nc < 1) nc = 1;
        const double c = a_bytes * ((tiles_n + nc - 1) / nc) + w_bytes * ((tiles_m + g - 1) / g);
        if (c < best_c * 0.999) { best_c = c; best = g; }
        if (gm >= tiles_m) break;
    }
    return best;
}

template <class CF>
void launch(const KParams& kp, bool fast, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<CF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
        (void)hipFuncSetAttribute((const void*)igemm_kernel<CF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
        attr_done = true;
    }
    const unsigned nblk = (unsigned)(kp.tiles_m * kp.tiles_n);
    if (fast) hipLaunchKernelGGL((igemm_kernel<CF, true>), dim3(nblk), dim3(CF::NT), CF::SMEM, s, kp);
    else      hipLaunchKernelGGL((igemm_kernel<CF, false>), dim3(nblk), dim3(CF::NT), CF::SMEM, s, kp);
}

int g_force_cfg = -1;

}  // namespace

// test hook: force a tile configuration (0 = 256x256, 1 = 128x320, 2 = 128x128, -1 = automatic)
extern "C" int pt_igemm_force_config(int32_t cfg) {
    PT_CHECK(cfg >= -1 && cfg <= 2, "pt_igemm_force_config: %d", cfg);
    g_force_cfg = cfg;
    return 0;
}

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
    // the kernel packs each output pixel's top-left tap as two signed 16-bit coordinates
    PT_CHECK((long long)p.Hout * p.stride + p.KH < 32000 && (long long)p.Wout * p.stride + p.KW < 32000,
             "pt_igemm_f16: output extent %d x %d too large (a linear layer is Nimg = M, H = W = 1)", p.Hout, p.Wout);
    PT_CHECK(!(p.upsample2x && p.stride != 1), "pt_igemm_f16: upsample2x needs stride 1");
    PT_CHECK(p.act == 0 || p.act == 2 || (p.act == 1 && p.N % 32 == 0), "pt_igemm_f16: act must be 0, 1 (GEGLU, N %% 32 == 0) or 2 (SiLU)");
    PT_CHECK(p.vec_mode == 0 || p.vec, "pt_igemm_f16: vec_mode without vec");
    PT_CHECK(pt_zero_page(), "pt_igemm_f16: zero page not set (pt_set_zero_page)");
    KParams kp;
    kp.p = p;
    if (!p.vec) kp.p.vec_mode = 0;
    kp.zeros = (const f16*)pt_zero_page();
    kp.npad = (p.N + 127) / 128 * 128;
    const int nout = p.act == 1 ? p.N / 2 : p.N;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    kp.vec_ok = (nout % 8 == 0) && (p.ldo % 8 == 0) && al16(p.out) && (!p.res || (p.ldr % 8 == 0 && al16(p.res))) &&
                (!p.vec || (p.ldv % 8 == 0 && al16(p.vec))) && (!p.blend || (p.ldb % 8 == 0 && al16(p.blend)));
    const bool fast = (Ctot % BK == 0) && (p.C0 % BK == 0) && (p.Kpad == p.K);
    const int cfg = g_force_cfg >= 0 ? g_force_cfg : choose_cfg(p.M, p.N, p.act);
    PT_CHECK(!(cfg == 1 && p.act == 1), "pt_igemm_f16: the 128x320 configuration does not support GEGLU");
    const int bm = cfg == 0 ? 256 : 128, bn = cfg == 0 ? 256 : (cfg == 1 ? 320 : 128);
    kp.tiles_m = (p.M + bm - 1) / bm;
    kp.tiles_n = (p.N + bn - 1) / bn;
    PT_CHECK((long long)kp.tiles_m * kp.tiles_n < (1ll << 31), "pt_igemm_f16: grid too large");
    {
        static const int gm_env = getenv("PT_IGEMM_GROUP_M") ? atoi(getenv("PT_IGEMM_GROUP_M")) : 0;   // tuning override
        const double in_px = p.upsample2x ? p.M / 4.0 : (double)p.M * p.stride * p.stride;
        kp.gm = gm_env > 0 ? (gm_env < kp.tiles_m ? gm_env : kp.tiles_m)
                           : choose_group(kp.tiles_m, kp.tiles_n, in_px * Ctot * 2.0, (double)p.N * p.K * 2.0, cfg == 2 ? 64 : 32);
    }
    hipStream_t s = (hipStream_t)stream;
    pt_prof_begin(0, s, 2.0 * (double)p.M * (double)p.N * (double)p.K);
    if (cfg == 0) launch<CfgBig>(kp, fast, s);
    else if (cfg == 1) launch<CfgW320>(kp, fast, s);
    else launch<CfgSmall>(kp, fast, s);
    pt_prof_end(0, s);
    PT_LAUNCH_CHECK("pt_igemm_f16");
    return 0;
}
