// Fused GEGLU feed-forward at C = 320 for gfx950 (MI355X):   out = tail( W2 . geglu(W1 . x + b1) + b2 )
//
//   BasicTransformerBlock.ff / TemporalBasicTransformerBlock.ff_in / .ff of the level-0 transformers (diffusers FeedForward
//   with GEGLU, called at /root/reference/models/modified_svd.py:97-103 and from BasicTransformerBlock.forward, :193-196):
//   Linear(320, 2560) -> h * gelu_erf(g) -> Linear(1280, 320) -> + residual (+ frame row vector | AlphaBlender).
//   As two pt_igemm_f16 launches the [M, 1280] intermediate (660 MB at 14 x 576 x 1024) is written by the first and read back
//   by the second, 21 times per loop iteration, and the first launch spends a third of its life in a store-bound GELU epilogue.
//   Here a workgroup owns 128 whole rows; the intermediate never leaves the CU.
//
//   Workgroup = 128 rows x all 320 output columns, 8 waves as 4 row pairs (32 rows) x 2 halves (s).  The hidden dimension is
//   walked in chunks of 64 (= 128 rows of the GEGLU-interleaved W1 pack):
//     stage 1  wave (pair, s): H1[32 rows, 64 W1 rows of half s] = X[32, 320] . W1c^T   - X lives in REGISTERS for the whole kernel
//              (20 B fragments loaded straight from global memory), W1 comes through LDS in five 64-deep K tiles;
//     GEGLU    value x gelu_erf(gate) on the accumulators -> fp16 -> the pair's rows of a [128, 64] LDS tile (natural k order);
//     stage 2  wave (pair, s): acc2[32 rows, 160 columns of half s] += h[32, 64] . W2c^T  - h from LDS, W2 in five 64-row pieces
//              laid out like igemm10_kernel's weight pieces (32 rows per wave half).
//   v_mfma_f32_16x16x32_f16 throughout, products transposed (a lane ends with 4 consecutive channels of one pixel), 0.53 LDS
//   fragment reads per MFMA (the 256 x 320 kernel: 0.35; sixteen private rows per wave would need 1.0).
//   Weights arrive by LDS-DMA into a ring that holds one whole chunk (5 x 16 KiB of W1 + 5 x 8 KiB of W2); every slot is
//   refilled for the next chunk two phases after its last read and waited for five phases later (one counted vmcnt(9) per
//   phase, raw s_barrier).  Phases of 16 MFMAs, two barriers each, the two wave groups (waves 0-3 / 4-7: one of each per SIMD,
//   a row pair never straddles them) one barrier apart as in igemm10_kernel: one group's MFMA cluster runs beside the other's
//   fragment reads, copies and GELU arithmetic.
//   Epilogue: the implicit-GEMM kernels' own (igemm_tail.h) - bias in the accumulators, rows through LDS, residual / row vector /
//   blend, 16-byte stores.  Same operands, same roundings (h goes to fp16 where the first launch stored it), same fp32
//   accumulation order in both products (ascending k in 32-deep MFMA steps): results are bit-identical to the two-launch form,
//   asserted in tests/test_kernels_gpu.py.
#include "igemm_tail.h"

namespace {

struct FParams {
    KParams kp;             // the SECOND linear layer as the epilogue sees it: N = 320, bias = b2, res / vec / blend / out ...
    const f16* x;           // [M, ldx] fp16: the LayerNorm output
    int ldx;
    const f16* w1;          // GEGLU-interleaved pack [2 * inner, kpad1]
    const f16* b1;          // [2 * inner] interleaved like w1, or null
    int kpad1;              // 320
    int nchunks;            // inner / 64
    int kpad2;              // row pitch of the w2 pack (= inner)
    // PRE (ffn320_kernel<.., PRE = true>): the attention output projection, its residual / row vector and the LayerNorm in front of
    // the feed-forward run in the kernel's prologue: x holds the ATTENTION OUTPUT rows
    const f16* wo;          // out-projection pack [384, kpado] (plain), bias [384] or null
    const f16* bo;
    int kpado;
    const f16* pres;        // the projection's residual (the block's h / u) [M, ldpr]
    int ldpr;
    const f16* pvec;        // its row vector (the collapsed cross-attention), indexed like pt_igemm_params.vec
    int ldpv, pvec_mode, pvG, pvFS, pvS, pvB;
    const f16* ln_g;        // LayerNorm gamma / beta [320]
    const f16* ln_b;
    float ln_eps;
};

using CFF = Cfg<4, 2, 2, 10>;                                // 128 x 320: 32 x 160 per wave
constexpr int F_W1T = 16384, F_W2P = 8192;
constexpr int F_W2_OFF = 5 * F_W1T, F_H_OFF = F_W2_OFF + 5 * F_W2P, F_B1_OFF = F_H_OFF + 16384, F_TRASH_OFF = F_B1_OFF + 5120;
constexpr int F_LN_OFF = F_TRASH_OFF + 8192;                 // PRE: LayerNorm gamma (1 KiB slot) | beta (1 KiB slot)
constexpr int F_SMEM = F_LN_OFF + 2048;                      // 154 624 B
constexpr int F_WO_BUF = 40960, F_YPITCH = 656, F_Y_OFF = 40960;   // PRE: one K tile of the out-projection (5 pieces of 8 KiB); the y tile: row pitch, offset
static_assert(F_Y_OFF + 128 * F_YPITCH <= F_B1_OFF && 2 * F_W1T <= F_Y_OFF, "the y tile lies behind the first two W1 tiles and in front of b1, where the LayerNorm statistics are exchanged");
static_assert(F_SMEM <= 160 * 1024 && CFF::SMEM <= F_B1_OFF, "LDS budget");

// ST: tuning build that writes s_memtime stamps of chunk 2's phases (pt_igemm_set_stamps; tools/ffn_stamps.py; slots 4 - 8 are the
// shared epilogue's)
// PRE: the kernel starts one step earlier in the transformer block - x holds the ATTENTION OUTPUT; the prologue computes
//   h = x . Wo^T + bo + res + vec[idx(m)]  (fp32, 32 rows x 160 columns per wave in the accumulators that later collect stage 2),
//   y = LayerNorm(h) (statistics over the pair's two waves through LDS), y -> fp16 -> an LDS tile -> the X fragments,
// and the feed-forward's residual is h itself: the accumulators start from h + b2 and the tail adds no residual.  Replaces
// attn1.to_out / attn1 of the temporal block + `+ residual` + norm3 (modified_svd.py:79-82,93-104; BasicTransformerBlock.forward) in
// front of ff.  h stays in fp32 where the three-launch form rounds it to fp16 twice (store, LayerNorm output is rounded either way).
template <int VAR, bool PRE = false, bool ST = false>
__global__ __launch_bounds__(512, 2) void ffn320_kernel(const FParams fp) {
    using CF = CFF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TN = 10;
    const KParams& kp = fp.kp;
    const pt_igemm_params& p = kp.p;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bid = pt_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = bid * 128;
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    // ---------------- X fragments: this wave's 32 rows x 320 channels, B-operand layout (pixel frow, k = 32 t + 8 fq ..)
    f16x8 Xf[2][10];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = min(m0 + wr * 32 + r * 16 + frow, p.M - 1);
        const f16* xp = fp.x + (size_t)row * fp.ldx + fq * 8;
#pragma unroll
        for (int tt = 0; tt < 10; ++tt) Xf[r][tt] = *(const f16x8*)(xp + 32 * tt);
    }
    const int swz = frow >> 1;
    const int c0 = (fq ^ swz) * 16, c1 = ((fq + 4) ^ swz) * 16;                     // byte offsets of the two 32-deep k halves
    const int csrc = (t & 7) ^ ((t >> 4) & 7);               // logical chunk a thread's copy stores (rows XOR-swizzled by (row >> 1) & 7)
    const int lr = t >> 3;                                   // 0 .. 63
    // W2 piece j of chunk c: LDS row (s, f, r) = (lr >> 5, (lr >> 4) & 1, lr & 15) holds W2 row 160 s + 32 j + 16 f + r
    const int w2row0 = (lr >> 5) * 160 + ((lr >> 4) & 1) * 16 + (lr & 15);
    // ---------------- LDS-DMA set-up.  One copy per thread moves 8 KiB: LDS row t >> 3, physical 16-B chunk t & 7
    // W1 K tile kt of chunk c: LDS rows 0 .. 127 = W1 rows 128 c + row (rows 0 .. 63 feed half s = 0, 64 .. 127 half s = 1)
    const int w1off = lr * fp.kpad1 + csrc * 8;              // + (128 c + 64 u) * kpad1 + 64 kt   (u = second copy)
    char* const dma0 = smem + wave * 1024;
    char* const trash = smem + F_TRASH_OFF + wave * 1024;
    const int nch = fp.nchunks;
    auto stageW1 = [&](int kt, int c) {                      // both copies of K tile kt of chunk c
        const bool live = c < nch;
        const f16* src = fp.w1 + (w1off + (size_t)(live ? c : nch - 1) * 128 * fp.kpad1 + 64 * kt);
        pt_glds16(src, live ? dma0 + kt * F_W1T : trash);
        pt_glds16(src + 64 * fp.kpad1, live ? dma0 + kt * F_W1T + 8192 : trash);
    };
    f32x4 acc2[TN][2];
    if constexpr (PRE) {
        // ---- bias of the projection; gamma / beta into LDS; its first two K tiles in flight
        f16x4 bo4[TN];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            bo4[ni] = fp.bo ? *(const f16x4*)(fp.bo + wc * 160 + ni * 16 + 4 * fq) : (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if (wave == 5) pt_glds16(fp.ln_g + min(lane, 39) * 8, smem + F_LN_OFF);
        if (wave == 6) pt_glds16(fp.ln_b + min(lane, 39) * 8, smem + F_LN_OFF + 1024);
        int wooff[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) wooff[j] = min(w2row0 + 32 * j, 383) * fp.kpado + csrc * 8;
        auto stageWo = [&](int kt, int buf) {
#pragma unroll
            for (int j = 0; j < 5; ++j) pt_glds16(fp.wo + (wooff[j] + 64 * kt), smem + buf * F_WO_BUF + j * 8192 + wave * 1024);
        };
        // the projection's residual and row vector are loaded with the rows and become the accumulators' INITIAL value (bo + res + vec):
        // their latency lies under the first copies' instead of behind the last MFMA
        f16x4 rs4[2][TN], vc4[2][TN];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = min(m0 + wr * 32 + r * 16 + frow, p.M - 1);
            const f16* rp = fp.pres + (size_t)row * fp.ldpr + wc * 160 + 4 * fq;
            int vi = 0;
            if (fp.pvec) vi = fp.pvec_mode == 1 ? row / fp.pvG : ((row / fp.pvFS) * fp.pvS + row % fp.pvS) % fp.pvB;
            const f16* vp = fp.pvec ? fp.pvec + (size_t)vi * fp.ldpv + wc * 160 + 4 * fq : (const f16*)kp.zeros;
            const int vstep = fp.pvec ? 16 : 0;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) { rs4[r][ni] = *(const f16x4*)(rp + ni * 16); vc4[r][ni] = *(const f16x4*)(vp + ni * vstep); }
        }
        stageWo(0, 0); stageWo(1, 1); stageWo(2, 2);
        __builtin_amdgcn_s_waitcnt(0x0F7F);                  // vmcnt(15): the attention rows, bo, residual and row vector have landed, the 15 copies fly
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[ni][r][j] = ((float)bo4[ni][j] + (float)rs4[r][ni][j]) + (float)vc4[r][ni][j];
        const char* const word = smem + (wc * 32 + frow) * 128;
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) {
            // K tile kt has landed (this wave's copies): tiles kt + 1, kt + 2 (5 copies each) may still fly
            if (kt < 3) __builtin_amdgcn_s_waitcnt(0x0F7A); else if (kt == 3) __builtin_amdgcn_s_waitcnt(0x0F75); else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            if (kt == 4) { stageW1(0, 0); stageW1(1, 0); }   // buffer 0 (K tile 3) is read: the feed-forward's first two W1 tiles land under the LayerNorm
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                f16x8 Wq[TN];
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    Wq[ni] = *(const f16x8*)(word + (kt % 3) * F_WO_BUF + (ni >> 1) * 8192 + (ni & 1) * 2048 + (kh ? c1 : c0));
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc2[ni][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wq[ni], Xf[r][2 * kt + kh], acc2[ni][r], 0, 0, 0);
            }
            if (kt + 3 < 5) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();                // every wave is done with buffer kt % 3
                stageWo(kt + 3, kt % 3);
            }
        }
        // ---- LayerNorm statistics of h (two passes, the pair's halves exchanged through LDS)
        float* const part = (float*)(smem + F_B1_OFF);       // [2 statistics][128 rows][2 halves]; b1 is copied in behind the prologue
        float mean[2], rstd[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float sum = 0.f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += acc2[ni][r][j];
            sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
            if (fq == 0) part[(wr * 32 + r * 16 + frow) * 2 + wc] = sum;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rl = wr * 32 + r * 16 + frow;
            mean[r] = (part[rl * 2] + part[rl * 2 + 1]) * (1.0f / 320.0f);
            float sq = 0.f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = acc2[ni][r][j] - mean[r]; sq += d * d; }
            sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
            if (fq == 0) part[256 + rl * 2 + wc] = sq;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        // ---- y = LayerNorm(h) -> fp16 -> the y tile (row pitch 656 B: the fragment reads below touch every bank once)
        const char* const gl = smem + F_LN_OFF + (wc * 160 + 4 * fq) * 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rl = wr * 32 + r * 16 + frow;
            rstd[r] = rsqrtf((part[256 + rl * 2] + part[256 + rl * 2 + 1]) * (1.0f / 320.0f) + fp.ln_eps);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const f16x4 g = *(const f16x4*)(gl + ni * 32), b = *(const f16x4*)(gl + 1024 + ni * 32);
                f16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (f16)((acc2[ni][r][j] - mean[r]) * rstd[r] * (float)g[j] + (float)b[j]);
                *(f16x4*)(smem + F_Y_OFF + rl * F_YPITCH + (wc * 160 + ni * 16 + 4 * fq) * 2) = o;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int tt = 0; tt < 10; ++tt) Xf[r][tt] = *(const f16x8*)(smem + F_Y_OFF + (wr * 32 + r * 16 + frow) * F_YPITCH + (32 * tt + 8 * fq) * 2);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                        // the y tile is read: the weight ring may land on it
    }
    f16x4 b4[TN];
    bias_issue<CF>(kp, 0, wave, lane, b4);
    if constexpr (ST) ig_stamp(kp, wave, lane, 0);
#define FF_ST(slot) if constexpr (ST) { if (c == 2) ig_stamp(kp, wave, lane, slot); }

    int w2off[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        int wrow = w2row0 + 32 * j;
        if (wrow >= kp.npad) wrow = kp.npad - 1;
        w2off[j] = wrow * fp.kpad2 + csrc * 8;
    }
    auto stageW2 = [&](int j, int c) {
        const bool live = c < nch;
        pt_glds16((const f16*)p.w + (w2off[j] + 64 * (live ? c : nch - 1)), live ? dma0 + F_W2_OFF + j * F_W2P : trash);
    };

    // ---------------- prologue: b1 into LDS, the first chunk's W1 tiles and W2 pieces 0, 1 in flight
    if (fp.b1) {
        if (wave < 5) pt_glds16(fp.b1 + t * 8, smem + F_B1_OFF + wave * 1024);
    } else if (t < 320) {
        *(f32x4*)(smem + F_B1_OFF + t * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kt = PRE ? 2 : 0; kt < 5; ++kt) stageW1(kt, 0);
    stageW2(0, 0); stageW2(1, 0);
    if constexpr (PRE) {                                     // the residual of the feed-forward is h itself: the accumulators continue from h + b2
        __builtin_amdgcn_s_waitcnt(0x0F78);                  // vmcnt(8): b2 (W1 tiles 2 .. 4 and the two W2 pieces stay in flight)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[ni][r][j] += (float)b4[ni][j];
    } else {
        bias_init<CF, 12>(b4, acc2);                         // X, b2 (and the b1 copy) have landed; the 12 weight copies stay in flight
    }
    __builtin_amdgcn_s_waitcnt(0x0F79);                      // vmcnt(9): K tile 0 (and 1) of W1
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    if constexpr (ST) ig_stamp(kp, wave, lane, 1);

    const char* const w1rd = smem + (wc * 64 + frow) * 128;                        // + kt * F_W1T + b * 2048
    const char* const w2rd = smem + F_W2_OFF + (wc * 32 + frow) * 128;             // + j * F_W2P + f * 2048
    const char* const hrd = smem + F_H_OFF + (wr * 32 + frow) * 128;               // + r * 2048
    char* const hwr = smem + F_H_OFF + (wr * 32 + frow) * 128 + 8 * (fq & 1);      // + r * 2048 + ((4 s + 2 hb + (fq >> 1)) ^ swz) * 16
    const char* const b1rd = smem + F_B1_OFF + (wc * 64 + 4 * fq) * 2;             // + (128 c + 16 b) * 2
    f32x4 acc1[4][2];
    f16x8 Wf[4][2], Hf[2][2];

#define FF_PHASE_MMA(body0, body1)                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    __builtin_amdgcn_s_barrier();                                                              \
    __builtin_amdgcn_s_waitcnt(0xC47F);                      /* lgkmcnt(4): the first k halves */ \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    body0                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    body1                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    __builtin_amdgcn_s_barrier();
#define FF_READ_W1(kt)                                                                         \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                            \
        _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                        \
            Wf[b_][h_] = *(const f16x8*)(w1rd + (kt) * F_W1T + b_ * 2048 + (h_ ? c1 : c0));
#define FF_MMA1(kt, h_)                                                                        \
    _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                            \
        _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                        \
            acc1[b_][r_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[b_][h_], Xf[r_][2 * (kt) + (h_)], acc1[b_][r_], 0, 0, 0);
#define FF_READ_W2(j0)        /* pieces j0 and j0 + 1: fragments (piece, f) -> Wf[2 * piece' + f] */ \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                            \
        _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                        \
            Wf[b_][h_] = *(const f16x8*)(w2rd + ((j0) + (b_ >> 1)) * F_W2P + (b_ & 1) * 2048 + (h_ ? c1 : c0));
#define FF_MMA2(j0, nb, h_)                                                                    \
    _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                            \
        _Pragma("unroll") for (int b_ = 0; b_ < (nb); ++b_)                                     \
            acc2[2 * (j0) + b_][r_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[b_][h_], Hf[r_][h_], acc2[2 * (j0) + b_][r_], 0, 0, 0);
#define FF_VMWAIT __builtin_amdgcn_s_waitcnt(0x0F79);        /* vmcnt(9): everything issued five phases ago has landed */

    const bool late = wave >= 4;
    if (late) __builtin_amdgcn_s_barrier();

    for (int c = 0; c < nch; ++c) {
        // ---- phase 0: acc1 = b1; stage 1, K tile 0
        FF_ST(2)
        {
            f16x4 bb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bb[b] = *(const f16x4*)(b1rd + (128 * c + 16 * b) * 2);
            FF_READ_W1(0)
            stageW2(2, c); stageW2(3, c);
            __builtin_amdgcn_s_waitcnt(0xC87F);              // lgkmcnt(8): the four bias reads
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 v = {(float)bb[b][0], (float)bb[b][1], (float)bb[b][2], (float)bb[b][3]};
                acc1[b][0] = v; acc1[b][1] = v;
            }
            FF_VMWAIT
        }
        FF_ST(3)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        FF_ST(9)
        __builtin_amdgcn_s_waitcnt(0xC47F);
        __builtin_amdgcn_sched_barrier(0);
        FF_MMA1(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        FF_MMA1(0, 1)
        __builtin_amdgcn_sched_barrier(0);
        
        __builtin_amdgcn_s_barrier();
        FF_ST(10)
        // ---- phase 1
        FF_READ_W1(1)
        stageW2(4, c);
        FF_VMWAIT
        FF_PHASE_MMA(FF_MMA1(1, 0), FF_MMA1(1, 1))
        // ---- phase 2
        FF_READ_W1(2)
        stageW1(0, c + 1);
        FF_VMWAIT
        FF_PHASE_MMA(FF_MMA1(2, 0), FF_MMA1(2, 1))
        // ---- phase 3
        FF_READ_W1(3)
        stageW1(1, c + 1);
        FF_VMWAIT
        FF_PHASE_MMA(FF_MMA1(3, 0), FF_MMA1(3, 1))
        // ---- phase 4
        FF_READ_W1(4)
        stageW1(2, c + 1);
        FF_VMWAIT
        FF_PHASE_MMA(FF_MMA1(4, 0), FF_MMA1(4, 1))
        FF_ST(11)
        // ---- phase 5: GEGLU -> this pair's rows of the h tile; stage 2, pieces 0 and 1
        {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const f32x4 val = acc1[2 * hb][r], gate = acc1[2 * hb + 1][r];
                    const f32x2 g01 = pt_gelu_erf2((f32x2){gate[0], gate[1]}), g23 = pt_gelu_erf2((f32x2){gate[2], gate[3]});
                    // the product is rounded to fp32 FIRST and to fp16 second, like the two-launch form (whose epilogue stages the
                    // fp32 product through LDS): left to itself hipcc folds multiply + convert into v_fma_mixlo_f16, one rounding
                    // of the exact product - more accurate, but then 0.08 % of the outputs differ from pt_igemm_f16's by an ulp
                    float p0 = val[0] * g01[0], p1 = val[1] * g01[1], p2 = val[2] * g23[0], p3 = val[3] * g23[1];
                    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                    const f16x4 o = {(f16)p0, (f16)p1, (f16)p2, (f16)p3};
                    *(f16x4*)(hwr + r * 2048 + ((4 * wc + 2 * hb + (fq >> 1)) ^ swz) * 16) = o;
                }
            __builtin_amdgcn_sched_barrier(0);
            FF_READ_W2(0)
            stageW1(3, c + 1);
            FF_VMWAIT
            __builtin_amdgcn_sched_barrier(0);               // the count below is only right with all 8 W2 reads ISSUED in front of it
            __builtin_amdgcn_s_waitcnt(0xC87F);              // lgkmcnt(8): this wave's h stores are in LDS (the W2 reads may still fly)
            FF_ST(12)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            
#pragma unroll
            for (int h_ = 0; h_ < 2; ++h_)
#pragma unroll
                for (int r = 0; r < 2; ++r) Hf[r][h_] = *(const f16x8*)(hrd + r * 2048 + (h_ ? c1 : c0));
            __builtin_amdgcn_s_waitcnt(0xC27F);              // lgkmcnt(2): W2 fragments and the first k half of h
            __builtin_amdgcn_sched_barrier(0);
            FF_MMA2(0, 4, 0)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            FF_MMA2(0, 4, 1)
            __builtin_amdgcn_sched_barrier(0);
            
            __builtin_amdgcn_s_barrier();
            FF_ST(13)
        }
        // ---- phase 6: pieces 2 and 3
        FF_READ_W2(2)
        stageW1(4, c + 1);
        FF_VMWAIT
        FF_PHASE_MMA(FF_MMA2(2, 4, 0), FF_MMA2(2, 4, 1))
        // ---- phase 7: piece 4
#pragma unroll
        for (int h_ = 0; h_ < 2; ++h_)
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) Wf[b_][h_] = *(const f16x8*)(w2rd + 4 * F_W2P + b_ * 2048 + (h_ ? c1 : c0));
        stageW2(0, c + 1); stageW2(1, c + 1);
        FF_VMWAIT
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_waitcnt(0xC27F);
        __builtin_amdgcn_sched_barrier(0);
        FF_MMA2(4, 2, 0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        FF_MMA2(4, 2, 1)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        FF_ST(14)
    }
    if (!late) __builtin_amdgcn_s_barrier();
#undef FF_PHASE_MMA
#undef FF_READ_W1
#undef FF_MMA1
#undef FF_READ_W2
#undef FF_MMA2
#undef FF_VMWAIT
    {
        int lane_t = lane;                                   // (opaque: keeps the tail's lane-derived values below the chunk loop)
        asm volatile("" : "+v"(lane_t));
        igemm_epilogue<CF, VAR>(kp, acc2, smem, m0, 0, wave, lane_t);
    }
#undef FF_ST
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // no LDS-DMA may outlive the wave
}

typedef void (*ffn_kernel_t)(const FParams);

}  // namespace

extern unsigned long long* g_stamps;                         // igemm.hip (pt_igemm_set_stamps)
extern long long g_stamps_cap;

extern "C" int pt_ffn_geglu_f16(const pt_ffn_params* pp, void* stream) {
    const pt_ffn_params& q = *pp;
    PT_CHECK(q.x && q.w1 && q.w2 && q.out, "pt_ffn_geglu_f16: null pointer");
    PT_CHECK(q.C == 320, "pt_ffn_geglu_f16: C = %d (built for the 320-channel level of the SVD U-Net; wider levels keep the two-launch form)", q.C);
    PT_CHECK(q.inner > 0 && q.inner % 64 == 0, "pt_ffn_geglu_f16: inner = %d must be a positive multiple of 64", q.inner);
    PT_CHECK(q.kpad1 == 320 && q.kpad2 >= q.inner && q.kpad2 % 64 == 0, "pt_ffn_geglu_f16: weight pitches %d / %d", q.kpad1, q.kpad2);
    PT_CHECK(q.M > 0 && q.ldx % 8 == 0 && q.ldo % 8 == 0, "pt_ffn_geglu_f16: M = %d, pitches must be multiples of 8", q.M);
    PT_CHECK(pt_zero_page(), "pt_ffn_geglu_f16: zero page not set (pt_set_zero_page)");
    auto al16 = [](const void* a) { return ((uintptr_t)a & 15) == 0; };
    PT_CHECK(al16(q.x) && al16(q.out) && al16(q.res) && al16(q.vec) && al16(q.blend) && al16(q.b1) && al16(q.w1) && al16(q.w2),
             "pt_ffn_geglu_f16: operands must be 16-byte aligned");
    PT_CHECK((!q.res || q.ldr % 8 == 0) && (!q.vec || q.ldv % 8 == 0) && (!q.blend || q.ldb % 8 == 0), "pt_ffn_geglu_f16: side-input pitches must be multiples of 8");
    PT_CHECK(q.vec_mode == 0 || q.vec, "pt_ffn_geglu_f16: vec_mode without vec");
    PT_CHECK((q.res ? 1 : 0) + (q.vec ? 1 : 0) + (q.blend ? 1 : 0) <= 2, "pt_ffn_geglu_f16: at most two side inputs");
    FParams fp;
    memset(&fp, 0, sizeof(fp));
    pt_igemm_params& p = fp.kp.p;
    p.M = q.M; p.N = q.C; p.K = q.inner; p.Kpad = q.kpad2;
    p.Nimg = q.M; p.Hin = p.Win = p.Hout = p.Wout = 1; p.KH = p.KW = 1; p.stride = 1;
    p.C0 = q.inner;
    p.w = q.w2; p.bias = q.b2;
    p.out = q.out; p.ldo = q.ldo;
    p.res = q.res; p.ldr = q.ldr;
    p.vec = q.vec; p.ldv = q.ldv; p.vec_mode = q.vec ? q.vec_mode : 0; p.vG = q.vG; p.vFS = q.vFS; p.vS = q.vS; p.vB = q.vB;
    p.blend = q.blend; p.ldb = q.ldb; p.alpha = q.alpha;
    p.out_scale = 1.0f; p.cs_scale = 1.0f;
    fp.kp.zeros = (const f16*)pt_zero_page();
    fp.kp.tiles_m = (q.M + 127) / 128; fp.kp.tiles_n = 1;
    fp.kp.npad = (q.C + 127) / 128 * 128;
    fp.kp.vec_ok = 1; fp.kp.gm = 1; fp.kp.splits = 1;

    fp.x = (const f16*)q.x; fp.ldx = q.ldx;
    fp.w1 = (const f16*)q.w1; fp.b1 = (const f16*)q.b1; fp.kpad1 = q.kpad1; fp.kpad2 = q.kpad2;
    fp.nchunks = q.inner / 64;
    const bool pre = q.pre_w != nullptr;
    if (pre) {
        PT_CHECK(q.pre_res && q.ln_gamma && q.ln_beta && !q.res, "pt_ffn_geglu_f16: pre_w needs pre_res, ln_gamma, ln_beta and no `res` (the block's own h is the residual)");
        PT_CHECK(q.pre_kpad == 320 && q.pre_ldr % 8 == 0 && (!q.pre_vec || (q.pre_ldv % 8 == 0 && (q.pre_vec_mode == 1 ? q.pre_vG > 0 : (q.pre_vec_mode == 2 && q.pre_vFS > 0 && q.pre_vS > 0 && q.pre_vB > 0)))),
                 "pt_ffn_geglu_f16: bad pre-projection arguments");
        PT_CHECK(al16(q.pre_w) && al16(q.pre_b) && al16(q.pre_res) && al16(q.pre_vec) && al16(q.ln_gamma) && al16(q.ln_beta), "pt_ffn_geglu_f16: pre-projection operands must be 16-byte aligned");
        fp.wo = (const f16*)q.pre_w; fp.bo = (const f16*)q.pre_b; fp.kpado = q.pre_kpad;
        fp.pres = (const f16*)q.pre_res; fp.ldpr = q.pre_ldr;
        fp.pvec = (const f16*)q.pre_vec; fp.ldpv = q.pre_ldv; fp.pvec_mode = q.pre_vec_mode; fp.pvG = q.pre_vG; fp.pvFS = q.pre_vFS; fp.pvS = q.pre_vS; fp.pvB = q.pre_vB;
        fp.ln_g = (const f16*)q.ln_gamma; fp.ln_b = (const f16*)q.ln_beta; fp.ln_eps = q.ln_eps;
    } else {
        PT_CHECK(!q.ln_gamma && !q.ln_beta && !q.pre_res && !q.pre_vec, "pt_ffn_geglu_f16: ln_gamma / pre_res / pre_vec without pre_w");
    }
    static const ffn_kernel_t table[2][3] = {{ffn320_kernel<V_P0>, ffn320_kernel<V_P1>, ffn320_kernel<V_P2>},
                                             {ffn320_kernel<V_P0, true>, ffn320_kernel<V_P1, true>, ffn320_kernel<V_P2, true>}};
    const int var = tail_variant(p);
    PT_CHECK(var >= V_P0 && var <= V_P2, "pt_ffn_geglu_f16: unsupported tail variant %d", var);
    static bool attr_done[64][2][3] = {};
    const int dev = pt_device();
    if (!attr_done[dev][pre][var]) {
        (void)hipFuncSetAttribute((const void*)table[pre][var], hipFuncAttributeMaxDynamicSharedMemorySize, F_SMEM);
        attr_done[dev][pre][var] = true;
    }
    hipStream_t s = (hipStream_t)stream;
    // counted with the implicit-GEMM family (bench.py's roofline leg): both products' algorithmic flops
    pt_prof_begin(PT_PROF_IGEMM, s, 2.0 * (double)q.M * (2.0 * q.inner) * q.C + 2.0 * (double)q.M * q.C * q.inner + (pre ? 2.0 * (double)q.M * q.C * q.C : 0.0));
    if (g_stamps && var == V_P1 && !pre) {                           // tuning: the stamped build of the residual-only variant
        fp.kp.stamps = g_stamps; fp.kp.stamps_cap = g_stamps_cap;
        static bool st_attr[64] = {};
        if (!st_attr[dev]) {
            (void)hipFuncSetAttribute((const void*)ffn320_kernel<V_P1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F_SMEM);
            st_attr[dev] = true;
        }
        hipLaunchKernelGGL((ffn320_kernel<V_P1, false, true>), dim3((unsigned)fp.kp.tiles_m), dim3(512), F_SMEM, s, fp);
    } else {
        hipLaunchKernelGGL(table[pre][var], dim3((unsigned)fp.kp.tiles_m), dim3(512), F_SMEM, s, fp);
    }
    pt_prof_end(PT_PROF_IGEMM, s);
    PT_LAUNCH_CHECK("pt_ffn_geglu_f16");
    return 0;
}
