// LayerNorm + linear layer at K = 320 as ONE kernel for gfx950 (MI355X):   out = LN(x; gamma, beta, eps) . W^T   (round 6)
//
//   norm1 -> attn1.to_q | to_k | to_v of BasicTransformerBlock and TemporalBasicTransformerBlock at the 320-channel level
//   (/root/reference/models/modified_svd.py:79-81 and BasicTransformerBlock.forward; bias-free projections, stacked [960, 320]).
//   As launches that was a LayerNorm pass (reads 165 MB, writes 165 MB at 14 x 576 x 1024) and a pt_igemm_f16 launch whose K = 320
//   loop cannot hide its prologue and its 500 MB store burst: 55 + 238 us, 27 % matrix-pipe utilisation (profiles/r06/mfma_busy_L_r06a.txt).
//   Here a workgroup owns 128 whole rows, so the statistics are a wave-local affair, the normalised rows never exist in memory
//   and the stores are spread over the whole kernel.  It is stage 1 of ffn320_kernel (csrc/ffn.hip) with a store where the GELU was.
//
//   Workgroup = 128 rows, 8 waves as 4 row pairs (32 rows each) x 2 halves (s).  Prologue: each wave loads its pair's 32 rows x 320
//   channels straight into B-operand fragments (80 VGPRs; a row's 320 channels sit in 4 lanes, both waves of a pair hold the same
//   rows), takes the row sums and sums of squares from 40 MFMAs on those very fragments (ones . X^T and the diagonal of X . X^T: the
//   VALU form cost 6.7 us per workgroup) and rewrites the fragments as y = (x - mean) * rstd * gamma + beta rounded to fp16 - the value
//   pt_layernorm_f16 would have stored, up to the summation order of the statistics.
//   The N output columns are walked in chunks of 128 weight rows (64 per half s): five 64-deep K tiles of 16 KiB each come through
//   LDS by LDS-DMA into a TEN-slot ring (two whole chunks, all of the CU's 160 KiB: slot = (chunk parity, K tile)), each slot refilled
//   two phases after its read with the tile of the chunk after next - EIGHT phases before it is needed - and waited for with ONE counted
//   vmcnt per phase.  The depth is there for the stores: a wave's four output stores per chunk sit in the same in-order counter as
//   its copies (MI355X_MICROARCH.md "vmcnt": loads, stores and LDS-DMA retire in issue order), so "the copy for the next phase has
//   landed" also means "every older store has been acknowledged"; with the five-slot ring of ffn320_kernel (copies two phases ahead)
//   a phase's wait depends on stores a phase or two old; eight phases ahead the youngest store a wait can depend on is seven phases
//   old.  (The first build - five slots, and a prologue that spilled values of the main loop, whose reloads there each came with a
//   vmcnt(0) - ran at 586 TFLOP/s, slower than the launch it replaces: profiles/r06/lnlin_bench_alone_first_build.txt.)  Raw s_barrier, two wave groups one barrier apart as in igemm10_kernel / ffn320_kernel.  16 x v_mfma_f32_16x16x32_f16 per phase and wave, products transposed (a lane ends with 4
//   consecutive channels of one pixel); two column blocks are combined by v_permlane16_swap so that a lane stores 16 bytes and a
//   row receives 64 contiguous bytes per instruction - no LDS staging, no barrier for the output.
//   Rounding points are those of the two-launch form (y to fp16, fp32 accumulation in ascending k, one rounding of acc * column
//   scale to fp16); the statistics are summed in another order than pt_layernorm_f16's, so y can differ from the stand-alone pass
//   by an fp16 ulp where a value sits on a rounding boundary (tests: <= 1e-3 of the two-launch result, ~1e-4 of fp32).
#include "igemm_tail.h"

namespace {

struct LParams {
    const f16* x; int ldx;      // [M, ldx] fp16
    int M, N;                   // N output columns (N % 8 == 0)
    const f16* w; int kpad;     // plain pack [Npad, kpad], Npad = ceil(N / 128) * 128 (zero rows behind N), kpad == 320
    const f16* g; const f16* b; // LayerNorm gamma / beta [320]
    float eps;
    f16* out; int ldo;
    int cs_cols; float cs_scale;   // columns < cs_cols leave multiplied by cs_scale (cs_cols % 64 == 0): the attention's pre-scaled Q
    int nchunks;                // Npad / 128
};

constexpr int L_SLOT = 16384, L_SMEM = 10 * L_SLOT;      // 160 KiB: the ring is the whole LDS
constexpr int L_LN_OFF = 9 * L_SLOT;                     // gamma | beta lie in the slot that is first written in phase 1 of chunk 0, long after the prologue read them

__device__ __forceinline__ void lq_swap16(uint32_t& a, uint32_t& b) {      // rows of 16 lanes: a's odd rows <-> b's even rows
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// output stores a wave has issued in the seven phase slots behind slot tau = 5 c + p (a slot has one iff tau >= 5 and tau % 5 != 4)
constexpr int lq_stores_behind(int tau) {
    int k = 0;
    for (int i = 1; i <= 7; ++i) k += (tau - i >= 5 && (tau - i) % 5 != 4) ? 1 : 0;
    return k;
}
static_assert(lq_stores_behind(5) == 0 && lq_stores_behind(9) == 4 && lq_stores_behind(10) == 4 && lq_stores_behind(12) == 6 &&
              lq_stores_behind(15) == 5 && lq_stores_behind(16) == 5 && lq_stores_behind(17) == 6 && lq_stores_behind(20) == 5 &&
              lq_stores_behind(21) == lq_stores_behind(16) && lq_stores_behind(24) == lq_stores_behind(19), "store count table");

// DBG: tuning ablations, compiled as instances of their own (pt_ln_linear_set_ablation; results are wrong): 1 = no output stores, 2 = no weight copies
// behind the prologue's, 4 = no MFMAs, 8 = return behind the prologue.  The product is DBG = 0.
template <int DBG>
__global__ __launch_bounds__(512, 2) void lnlin320_kernel(const LParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bid = pt_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = bid * 128;
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    // ---------------- this wave's 32 rows x 320 channels, B-operand layout (pixel frow, k = 32 t + 8 fq ..); gamma / beta likewise
    f16x8 Xf[2][10];
    int grow[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        grow[r] = m0 + wr * 32 + r * 16 + frow;
        const f16* xp = lp.x + (size_t)min(grow[r], lp.M - 1) * lp.ldx + fq * 8;
#pragma unroll
        for (int tt = 0; tt < 10; ++tt) Xf[r][tt] = *(const f16x8*)(xp + 32 * tt);
    }
    // gamma | beta into LDS (one LDS-DMA each: 40 lanes x 16 bytes; they would cost 80 registers beside the rows)
    if (wave == 5) pt_glds16(lp.g + min(lane, 39) * 8, smem + L_LN_OFF);
    if (wave == 6) pt_glds16(lp.b + min(lane, 39) * 8, smem + L_LN_OFF + 1024);

    // ---------------- LDS-DMA set-up (as ffn320_kernel's W1 ring).  One copy per thread moves 8 KiB: LDS row t >> 3, physical chunk t & 7
    const int swz = frow >> 1;
    const int c0 = (fq ^ swz) * 16, c1 = ((fq + 4) ^ swz) * 16;                     // byte offsets of the two 32-deep k halves
    const int csrc = (t & 7) ^ ((t >> 4) & 7);               // logical chunk a thread's copy stores (rows XOR-swizzled by (row >> 1) & 7)
    const int lr = t >> 3;                                   // 0 .. 63
    const int woff = lr * lp.kpad + csrc * 8;                // + (128 c + 64 u) * kpad + 64 kt   (u = second copy)
    char* const dma0 = smem + wave * 1024;
    const int nch = lp.nchunks;
    bool copies_on = true;
    auto stageW = [&](int kt, int c, int par) {              // both copies of K tile kt of chunk c -> slot (par, kt), par = c & 1 at compile time;
        if (!copies_on) return;
        const f16* src = lp.w + (woff + (size_t)min(c, nch - 1) * 128 * lp.kpad + 64 * kt);   // past the end: the last chunk again, into a slot nobody reads any more
        pt_glds16(src, dma0 + (par * 5 + kt) * L_SLOT);
        pt_glds16(src + 64 * lp.kpad, dma0 + (par * 5 + kt) * L_SLOT + 8192);
    };
    stageW(0, 0, 0); stageW(1, 0, 0); stageW(2, 0, 0); stageW(3, 0, 0); stageW(4, 0, 0);
    stageW(0, 1, 1); stageW(1, 1, 1); stageW(2, 1, 1);       // (the copy issued in phase p of chunk c is tile 5 c + p + 8 of the stream)

    // ---------------- LayerNorm statistics on the matrix pipe.  A wave64 VALU instruction takes four cycles: two passes of convert / add /
    // subtract / multiply-add over a wave's 2 x 80 values per lane were ~800 instructions, and with two waves per SIMD the prologue was
    // 6.7 us of nothing but that (53 us of the launch: profiles/r06/lnlin_ablations.txt).  Instead, with the fragments as they are:
    //   D = ones . X^T : every row of D is sum_k x[pixel][k] - each lane reads its pixel's sum in D[.][frow];
    //   D = X . X^T    : A and B operands share one register layout, and the DIAGONAL D[p][p] = sum_k x[p][k]^2 (fp16 products are exact in
    //                    fp32, fp32 accumulation) sits in lane (frow = p, fq = p >> 2), element p & 3 - fetched by the pixel's other lanes.
    // var = E[x^2] - mean^2 in fp32 (hidden states are centred to within a few sigma; the two-pass form of pt_layernorm_f16 differs from
    // this by ~1e-6 relative in rstd, an fp16 ulp on isolated outputs).
    float mean[2], rstd[2];
    {
        const f16x8 ones = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
        const int diag_lane = (frow >> 2) * 16 + frow;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tt = 0; tt < 10; ++tt) {
                s4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, Xf[r][tt], s4, 0, 0, 0);
                q4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Xf[r][tt], Xf[r][tt], q4, 0, 0, 0);
            }
            const int j = frow & 3;
            const float dg = j == 0 ? q4[0] : (j == 1 ? q4[1] : (j == 2 ? q4[2] : q4[3]));
            const float sq = __shfl(dg, diag_lane);
            mean[r] = s4[0] * (1.0f / 320.0f);
            rstd[r] = rsqrtf(fmaxf(sq * (1.0f / 320.0f) - mean[r] * mean[r], 0.f) + lp.eps);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // (gamma / beta of waves 5 and 6 have landed; so has everything else)
    __builtin_amdgcn_s_barrier();
    {
        const char* const gl = smem + L_LN_OFF + fq * 16;
#pragma unroll
        for (int tt = 0; tt < 10; ++tt) {
            const f16x8 g = *(const f16x8*)(gl + tt * 64), b = *(const f16x8*)(gl + 1024 + tt * 64);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)(((float)Xf[r][tt][j] - mean[r]) * rstd[r] * (float)g[j] + (float)b[j]);
                Xf[r][tt] = o;
            }
        }
    }
    // (the first eight tiles landed with the wait above and every wave has passed the barrier behind it: no second one is needed)

    if (DBG & 8) return;
    copies_on = !(DBG & 2);
    const char* const wrd = smem + (wc * 64 + frow) * 128;                         // + kt * L_SLOT + b * 2048
    f32x4 acc[4][2];
    f16x8 Wf[4][2];
    // output addressing: after the swap lane (frow, fq) owns 8 channels of block 2 bp + (fq & 1): columns 16 * that + 8 * (fq >> 1)
    const int ocol = 64 * wc + 16 * (fq & 1) + 8 * (fq >> 1);                      // + 128 c + 32 bp
    f16* const orow0 = lp.out + (size_t)min(grow[0], lp.M - 1) * lp.ldo + ocol;
    f16* const orow1 = lp.out + (size_t)min(grow[1], lp.M - 1) * lp.ldo + ocol;
    const bool ok0 = grow[0] < lp.M, ok1 = grow[1] < lp.M;
    const bool rows_full = m0 + wr * 32 + 32 <= lp.M;          // (wave-uniform)
    const bool st_counted = rows_full && !(DBG & 3);

#define LQ_PHASE_MMA(body0, body1)                                                             \
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
#define LQ_READ_W(par, kt)                                                                     \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                            \
        _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                        \
            Wf[b_][h_] = *(const f16x8*)(wrd + ((par) * 5 + (kt)) * L_SLOT + b_ * 2048 + (h_ ? c1 : c0));
#define LQ_MMA(kt, h_)                                                                         \
    _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                            \
        _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                        \
            if (!(DBG & 4)) acc[b_][r_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[b_][h_], Xf[r_][2 * (kt) + (h_)], acc[b_][r_], 0, 0, 0);
    // One counted wait per phase: the tile the NEXT phase reads (stream index 5 c + p + 1, issued eight phases before its use) has landed
    // when all but the operations younger than it have retired: the 7 x 2 copies issued since, plus ONE per output store issued since.
    // The four stores of a finished chunk leave one per phase - phases 0 .. 3 of the next chunk, each right BEHIND its phase's wait (all
    // four in phase 0 cost the workgroup ~2 000 cycles per chunk: 32 KiB through a 16-byte-per-clock store path while every wave stood
    // at its stores and the phase's barrier waited for them: 262 us against 204 without stores, profiles/r06/lnlin_bench_alone_ring10_burst.txt).
    // Phase slot tau = 5 c + p has a store iff tau >= 5 and tau % 5 != 4; the wait of slot tau looks back over slots tau - 7 .. tau - 1.
    // A wave that may have skipped stores (rows beyond M) counts none: it then waits for a store or two - slower, never early.  In-loop
    // stores are always whole in the columns (chunk c - 1 <= nch - 2 lies below N).
    // Counted over the slots behind a wait (lq_stores_behind, evaluated at compile time: the first three chunks are peeled, from the fourth
    // on the count is periodic):   chunk 0: 0 0 0 0 0   chunk 1: 0 1 2 3 4   chunk 2: 4 5 6 6 6   chunk >= 3: 5 5 6 6 6   (phases 0 .. 4)
#define LQ_WAIT_IMM(n_) ((n_) < 16 ? (0x0F70 | (n_)) : (0x4F70 | ((n_) - 16)))                   /* s_waitcnt vmcnt(n): bits 3:0 and 15:14 */
#define LQ_VMWAIT(pp, ci)                                                                      \
    if (st_counted) { __builtin_amdgcn_s_waitcnt(LQ_WAIT_IMM(14 + lq_stores_behind(5 * (ci) + (pp)))); } else { __builtin_amdgcn_s_waitcnt(0x0F7E); }
    // the finished chunk: fp32 -> (x column scale) -> fp16, pairs of column blocks exchanged across the 16-lane rows -> four 16-byte
    // values per lane (ov[2 bp + r]), stored one per phase
    u32x4 ov[4];
#define LQ_CONVERT(cc)                                                                         \
    {                                                                                          \
        const float sc_ = 128 * (cc) + 64 * wc < lp.cs_cols ? lp.cs_scale : 1.0f;              \
        _Pragma("unroll") for (int bp_ = 0; bp_ < 2; ++bp_)                                     \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                  \
                const f32x4 a_ = acc[2 * bp_][r_], b_ = acc[2 * bp_ + 1][r_];                  \
                union { f16x4 h; uint32_t u[2]; } pa_, pb_;                                    \
                pa_.h = (f16x4){(f16)(a_[0] * sc_), (f16)(a_[1] * sc_), (f16)(a_[2] * sc_), (f16)(a_[3] * sc_)}; \
                pb_.h = (f16x4){(f16)(b_[0] * sc_), (f16)(b_[1] * sc_), (f16)(b_[2] * sc_), (f16)(b_[3] * sc_)}; \
                lq_swap16(pa_.u[0], pb_.u[0]);                                                 \
                lq_swap16(pa_.u[1], pb_.u[1]);                                                 \
                ov[2 * bp_ + r_] = (u32x4){pa_.u[0], pa_.u[1], pb_.u[0], pb_.u[1]};            \
            }                                                                                  \
    }
#define LQ_PUT(cc, s_)                                                                         \
    {                                                                                          \
        const int col_ = 128 * (cc) + 32 * ((s_) >> 1);                                        \
        if (128 * (cc) + 64 * wc < lp.N && !(DBG & 1) && (((s_) & 1) ? ok1 : ok0) && col_ + ocol + 8 <= lp.N)  \
            *(u32x4*)((((s_) & 1) ? orow1 : orow0) + col_) = ov[s_];                           \
    }

    const bool late = wave >= 4;
    if (late) __builtin_amdgcn_s_barrier();

    // one chunk: five phases; P = c & 1 and CI = min(c, 3) at compile time (chunks 0 .. 2 peeled, then the loop unrolled by two)
#define LQ_CHUNK(P, CI)                                                                        \
    {                                                                                          \
        /* ---- phase 0: K tile 0; the previous chunk's accumulators become four 16-byte values, the first one leaves */ \
        LQ_READ_W(P, 0)                                                                        \
        stageW(3, c + 1, 1 - (P));                                                             \
        LQ_VMWAIT(0, CI)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((CI) > 0) { LQ_CONVERT(c - 1) LQ_PUT(c - 1, 0) }                                   \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        _Pragma("unroll") for (int b = 0; b < 4; ++b) { acc[b][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[b][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
        LQ_PHASE_MMA(LQ_MMA(0, 0), LQ_MMA(0, 1))                                               \
        /* ---- phase 1 */                                                                     \
        LQ_READ_W(P, 1)                                                                        \
        stageW(4, c + 1, 1 - (P));                                                             \
        LQ_VMWAIT(1, CI)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((CI) > 0) LQ_PUT(c - 1, 1)                                                         \
        LQ_PHASE_MMA(LQ_MMA(1, 0), LQ_MMA(1, 1))                                               \
        /* ---- phase 2 */                                                                     \
        LQ_READ_W(P, 2)                                                                        \
        stageW(0, c + 2, P);                                                                   \
        LQ_VMWAIT(2, CI)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((CI) > 0) LQ_PUT(c - 1, 2)                                                         \
        LQ_PHASE_MMA(LQ_MMA(2, 0), LQ_MMA(2, 1))                                               \
        /* ---- phase 3 */                                                                     \
        LQ_READ_W(P, 3)                                                                        \
        stageW(1, c + 2, P);                                                                   \
        LQ_VMWAIT(3, CI)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((CI) > 0) LQ_PUT(c - 1, 3)                                                         \
        LQ_PHASE_MMA(LQ_MMA(3, 0), LQ_MMA(3, 1))                                               \
        /* ---- phase 4 */                                                                     \
        LQ_READ_W(P, 4)                                                                        \
        stageW(2, c + 2, P);                                                                   \
        LQ_VMWAIT(4, CI)                                                                       \
        LQ_PHASE_MMA(LQ_MMA(4, 0), LQ_MMA(4, 1))                                               \
    }
    {
        int c = 0;
        LQ_CHUNK(0, 0)
        if (nch > 1) { c = 1; LQ_CHUNK(1, 1) }
        if (nch > 2) { c = 2; LQ_CHUNK(0, 2) }
        for (c = 3; c < nch; ++c) {
            LQ_CHUNK(1, 3)
            if (++c >= nch) break;
            LQ_CHUNK(0, 3)
        }
    }
#undef LQ_CHUNK
    if (!late) __builtin_amdgcn_s_barrier();
    LQ_CONVERT(nch - 1)
    LQ_PUT(nch - 1, 0) LQ_PUT(nch - 1, 1) LQ_PUT(nch - 1, 2) LQ_PUT(nch - 1, 3)
#undef LQ_PHASE_MMA
#undef LQ_READ_W
#undef LQ_MMA
#undef LQ_VMWAIT
#undef LQ_WAIT_IMM
#undef LQ_CONVERT
#undef LQ_PUT
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // no LDS-DMA may outlive the wave
}

}  // namespace

namespace { int g_lnlin_ablation = 0; }

// tuning hook (like pt_igemm_force_config; tools/lnlin_bench.py): run an ablation instance of the kernel - 1 = no output stores, 2 = no weight
// copies behind the prologue's, 3 = both, 4 = no MFMAs, 8 = return behind the prologue.  Results are WRONG unless 0 (the product).
extern "C" int pt_ln_linear_set_ablation(int32_t bits) {
    PT_CHECK(bits == 0 || bits == 1 || bits == 2 || bits == 3 || bits == 4 || bits == 8, "pt_ln_linear_set_ablation: %d (0, 1, 2, 3, 4 or 8)", bits);
    g_lnlin_ablation = bits;
    return 0;
}

extern "C" int pt_ln_linear_f16(const pt_lnlin_params* pp, void* stream) {
    const pt_lnlin_params& q = *pp;
    PT_CHECK(q.x && q.w && q.out && q.ln_gamma && q.ln_beta, "pt_ln_linear_f16: null pointer");
    PT_CHECK(q.K == 320 && q.kpad == 320, "pt_ln_linear_f16: K = %d, kpad = %d (built for the 320-channel level of the SVD U-Net)", q.K, q.kpad);
    PT_CHECK(q.M > 0 && q.N > 0 && q.N % 8 == 0, "pt_ln_linear_f16: M = %d, N = %d (N must be a positive multiple of 8)", q.M, q.N);
    PT_CHECK(q.ldx % 8 == 0 && q.ldx >= 320 && q.ldo % 8 == 0 && q.ldo >= q.N, "pt_ln_linear_f16: pitches %d / %d", q.ldx, q.ldo);
    PT_CHECK(q.cs_cols >= 0 && q.cs_cols % 64 == 0, "pt_ln_linear_f16: cs_cols = %d must be a non-negative multiple of 64", q.cs_cols);
    PT_CHECK(!q.bias, "pt_ln_linear_f16: a bias is not offered (the attention projections it serves have none)");
    auto al16 = [](const void* a) { return ((uintptr_t)a & 15) == 0; };
    PT_CHECK(al16(q.x) && al16(q.w) && al16(q.out) && al16(q.ln_gamma) && al16(q.ln_beta), "pt_ln_linear_f16: operands must be 16-byte aligned");
    LParams lp;
    lp.x = (const f16*)q.x; lp.ldx = q.ldx; lp.M = q.M; lp.N = q.N;
    lp.w = (const f16*)q.w; lp.kpad = q.kpad;
    lp.g = (const f16*)q.ln_gamma; lp.b = (const f16*)q.ln_beta; lp.eps = q.ln_eps;
    lp.out = (f16*)q.out; lp.ldo = q.ldo;
    lp.cs_cols = q.cs_cols; lp.cs_scale = q.cs_cols > 0 ? q.cs_scale : 1.0f;
    lp.nchunks = (q.N + 127) / 128;
    const int dbg = g_lnlin_ablation;                        // tuning ablations (pt_ln_linear_set_ablation): results are wrong unless 0
    typedef void (*lq_kernel_t)(const LParams);
    static const int dbg_ids[6] = {0, 1, 2, 3, 4, 8};
    static const lq_kernel_t table[6] = {lnlin320_kernel<0>, lnlin320_kernel<1>, lnlin320_kernel<2>, lnlin320_kernel<3>, lnlin320_kernel<4>, lnlin320_kernel<8>};
    int ki = 0;
    for (int i = 0; i < 6; ++i) if (dbg_ids[i] == dbg) ki = i;
    static bool attr_done[64][6] = {};
    const int dev = pt_device();
    if (!attr_done[dev][ki]) {
        (void)hipFuncSetAttribute((const void*)table[ki], hipFuncAttributeMaxDynamicSharedMemorySize, L_SMEM);
        attr_done[dev][ki] = true;
    }
    hipStream_t s = (hipStream_t)stream;
    pt_prof_begin(PT_PROF_IGEMM, s, 2.0 * (double)q.M * q.N * q.K);          // counted with the implicit-GEMM family (bench.py's roofline leg)
    hipLaunchKernelGGL(table[ki], dim3((unsigned)((q.M + 127) / 128)), dim3(512), L_SMEM, s, lp);
    pt_prof_end(PT_PROF_IGEMM, s);
    PT_LAUNCH_CHECK("pt_ln_linear_f16");
    return 0;
}
