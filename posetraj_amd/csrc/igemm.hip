// Implicit-GEMM convolution / linear layer for gfx950 (MI355X).
//
//   D[n, m] = sum_k W[n, k] * A[m, k]       (computed transposed so each lane ends up with 4 consecutive
//                                            output channels of one pixel -> row-contiguous epilogue)
//   v_mfma_f32_16x16x32_f16, fp32 accumulate, K step 64.  Six tile configurations (waves WM x WN, per-wave
//   tile TM x TN of 16 x 16 accumulators), chosen per shape by choose_cfg():
//       256 x 320  (8 waves 4 x 2, 64 x 160 per wave)   igemm10_kernel: every channel count of the network is a
//                                                       multiple of 320 - the workhorse (10-phase ping-pong loop)
//       256 x 256  (8 waves 4 x 2, 64 x 128 per wave)   igemm8_kernel (8-phase ping-pong) when K tiles are channel
//                                                       aligned, else the plain loop below
//       128 x 320  (8 waves 2 x 4,  64 x  80 per wave)  plain 2-stage loop: generic-K layers, N = 320 k
//       128 x 160  (4 waves 4 x 1,  32 x 160 per wave)  plain loop, 2 workgroups per CU
//       128 x 128  (4 waves 2 x 2,  64 x 64 per wave)   small or ragged problems, 2 workgroups per CU
//       256 x  32  (4 waves 4 x 1,  64 x 32 per wave)   plain loop: N <= 96 (condition encoder, conv_out)
//   Both operands are K-contiguous in memory (channels-last activations, [N, K] packed weights), so both tiles
//   are staged with 16-byte LDS-DMA (global_load_lds_dwordx4) straight from a per-lane gathered source address:
//   im2col, zero padding (padded taps read a zero page), the 2-source skip concatenation and the nearest-2x
//   upsampling all live in that address computation and cost no extra pass over HBM.
//   LDS image: rows of 128 B (64 halfs), the eight 16-B chunks of row r XOR-swizzled by (r >> 1) & 7 on the
//   SOURCE side (the DMA destination is lane-linear), undone in the ds_read_b128 address: conflict-free reads.
//   Two LDS stages; the DMA of tile k+1 is in flight while the MFMAs of tile k run.
//   Epilogue: bias (+GEGLU / SiLU) on the accumulators, then the wave's full width transposed through LDS in row
//   chunks (fp32, padded rows) and finished row-wise in >= 128-byte segments: residual / broadcast row vector / AlphaBlender lerp / scale, 16-byte loads+stores.
#include "igemm_tail.h"

namespace {

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
    f16x4 b4[TN];
    bias_issue<CF>(kp, n0, wave, lane, b4);

    // ---------------- staging set-up: this thread copies chunk slot (t + NT i) of each tile
    const int csrc = (t & 7) ^ ((t >> 4) & 7);               // source chunk (row parity bits are i-independent)
    const int Ctot = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    int iyx[CF::A_SLOTS], pix0[CF::A_SLOTS];                  // (iy0 << 16 | ix0 & 0xffff): top-left tap of the output pixel
#pragma unroll
    for (int i = 0; i < CF::A_SLOTS; ++i) {
        const int m = m0 + (t >> 3) + (NT / 8) * i;
        if (m < p.M) {
            int img = m, oy = 0, ox = 0;                     // linear layers (one-pixel images) skip the two divisions
            if (HWo != 1) {
                img = m / HWo;
                const int rem = m - img * HWo;
                oy = rem / p.Wout; ox = rem - oy * p.Wout;
            }
            iyx[i] = ((oy * p.stride - p.pad_h) << 16) | ((kp.foldx ? 0 : ox * p.stride - p.pad_w) & 0xffff);
            pix0[i] = img * p.Hin * p.Win + (kp.foldx ? ox : 0);
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
    bias_init<CF, CF::A_SLOTS + CF::B_SLOTS>(b4, acc);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk - 1; ++kt) {
        stage(kt + 1, cur ^ 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    igemm_epilogue<CF, V_RT>(kp, acc, smem, m0, n0, wave, lane);
}

// ============================================================================ 256 x 256, 8-phase ping-pong main loop
// The plain loop above issues a K tile's 8 LDS-DMA copies per wave back to back; with all 8 waves doing so at once
// the CU's single L2 -> LDS path (~56 B/clk) needs ~1100 cycles to take them, during which the waves cannot issue
// MFMAs (ablation: profiles/r01/igemm_ablation_v13.txt - removing the copies alone made the loop 1.45-2.0x faster).
// This loop splits each K tile into four half-operand pieces (16 KiB: 2 copies per thread) and four phases of 16
// MFMAs (one quadrant of the wave's 64 x 128 block), and staggers the two wave groups (waves 0-3 / 4-7, one of each
// per SIMD) by one barrier, so that while one group runs its MFMA cluster the other issues its fragment reads and
// its two copies; three pieces stay in flight across the barriers (counted vmcnt, raw s_barrier).
//   LDS: 2 buffers x [X0 | X1 | W0 | W1] x 16 KiB.  X_h = pixel rows {wr*64 + h*32 + i}, W_h = weight rows
//   {wc*128 + h*64 + j}: each piece holds, for every wave, exactly the rows of one operand half.
//   Per K tile t (buffer b = t & 1), pieces staged / fragments read / quadrant multiplied:
//     phase 1: read X0, W0 (b)   stage W1(t+1) -> b^1    MFMA X0 x W0
//     phase 2: read X1           stage X0(t+2) -> b      MFMA X1 x W0
//     phase 3: read W1           stage W0(t+2) -> b      MFMA X1 x W1
//     phase 4:                   stage X1(t+2) -> b      MFMA X0 x W1     vmcnt(6): all of tile t+1 has landed
//   Hazards.  A piece is overwritten two phases after its last read, or one phase after when those reads were retired
//   before the reading phase's first barrier (X0: its 4 reads are issued first, lgkmcnt(8) of 12).  A buffer is read
//   one phase after the wait that retires it (phase 4's wait, phase 1's reads), on the far side of a barrier every
//   wave passed after its own wait.  Copies past the last K tile are still issued so the counts stay uniform; they
//   land in per-wave trash rows behind the ring, their X copies double as the L2 prefetch of the epilogue's residual
//   and blend rows (advanceA), and nothing waits for them until the wave ends.
template <int VAR>
__global__ __launch_bounds__(512, 2) void igemm8_kernel(const KParams kp) {
    using CF = Cfg<4, 2, 4, 8>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TM = 4, TN = 8, BM = 256, BN = 256;
    constexpr int PIECE = 16384, BUF = 4 * PIECE;
    const pt_igemm_params& p = kp.p;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bid = pt_xcd_remap(blockIdx.x, gridDim.x);
    const int gsz = kp.gm * kp.tiles_n, grp = bid / gsz, first_m = grp * kp.gm;
    const int gm = min(kp.gm, kp.tiles_m - first_m), within = bid - grp * gsz;
    const int tile_m = first_m + within % gm, tile_n = within / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f16x4 b4[TN];
    bias_issue<CF>(kp, n0, wave, lane, b4);
    ig_stamp(kp, wave, lane, 0);

    // ---------------- staging set-up.  Slot a = 2 h + s: piece h, copy s of this thread (chunk t + 512 s of the piece)
    const int csrc = (t & 7) ^ ((t >> 4) & 7);
    const int HWo = p.Hout * p.Wout;
    int iyx[4], pix0[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int h = a >> 1, sl = a & 1;
        const int m = m0 + ((t >> 8) + 2 * sl) * 64 + h * 32 + ((t >> 3) & 31);
        if (m < p.M) {
            int img = m, oy = 0, ox = 0;                     // linear layers (one-pixel images) skip the two divisions
            if (HWo != 1) {
                img = m / HWo;
                const int rem = m - img * HWo;
                oy = rem / p.Wout; ox = rem - oy * p.Wout;
            }
            iyx[a] = ((oy * p.stride - p.pad_h) << 16) | ((kp.foldx ? 0 : ox * p.stride - p.pad_w) & 0xffff);
            pix0[a] = img * p.Hin * p.Win + (kp.foldx ? ox : 0);
        } else {
            iyx[a] = (int)0xC0000000; pix0[a] = 0;
        }
    }
    const int Hlim = p.upsample2x ? 2 * p.Hin : p.Hin, Wlim = p.upsample2x ? 2 * p.Win : p.Win;
    const f16* zsrc = kp.zeros + (lane & 7) * 8;
    int woff[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int wrow = n0 + (a & 1) * 128 + (a >> 1) * 64 + (t >> 3);
        if (wrow >= kp.npad) wrow = kp.npad - 1;
        woff[a] = wrow * p.Kpad + csrc * 8;
    }
    const f16* wbase = (const f16*)p.w;
    const int nk = p.Kpad / BK;

    const f16* aptr[4];
    unsigned avalid = 0;
    int s_tap = 0, s_src = 0, s_ci = 0, ci_cur = 0, st_tile = 0;   // wave-uniform position of the next K tile to stage
    bool a_past = false;                                     // the X pieces being staged lie past the last K tile
    auto advanceA = [&]() {                                  // fix the source pointers of K tile st_tile (both X pieces)
        if (st_tile >= nk) {
            // Past the last K tile the X copies carry no operand data: aim them at this wave's block of the residual
            // (first dummy tile) and of the blend input (second), one 128-byte line per lane, so the epilogue's side
            // loads - issued ~2 K tiles later - hit L2 instead of paying an HBM round trip per row pass.
            const f16* side = st_tile == nk ? (const f16*)p.res : (const f16*)p.blend;
            const int ldside = st_tile == nk ? p.ldr : p.ldb;
            const int nw = p.act == 1 ? TN * 8 : TN * 16;                              // this wave's output columns
            const int wcol = p.act == 1 ? (n0 + (wave & 1) * TN * 16) / 2 : n0 + (wave & 1) * TN * 16;
            const int nout = p.act == 1 ? p.N / 2 : p.N;
            const int mrow = min(m0 + (wave >> 1) * 64 + lane, p.M - 1);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int c = wcol + min(a * 64, nw - 8);
                aptr[a] = (side && c + 8 <= nout) ? side + ((size_t)mrow * ldside + c) : zsrc;
            }
            avalid = 0; ci_cur = 0; a_past = true;
        } else {
            if (s_ci == 0) {
                const int ky = s_tap / p.KW, kx = s_tap - ky * p.KW;
                const f16* src = s_src ? (const f16*)p.x1 : (const f16*)p.x0;
                const int ld = s_src ? p.ld1 : p.ld0;
                avalid = 0;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    int iy = (iyx[a] >> 16) + ky, ix = (int)(short)(iyx[a] & 0xffff) + kx;
                    const bool ok = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
                    if (p.upsample2x) { iy >>= 1; ix >>= 1; }
                    aptr[a] = ok ? src + ((size_t)(pix0[a] + iy * p.Win + ix) * ld + csrc * 8) : zsrc;
                    avalid |= ok ? (1u << a) : 0u;
                }
            }
            ci_cur = s_ci;
            s_ci += BK;
            if (s_ci == (s_src ? p.C1 : p.C0)) {
                s_ci = 0;
                if (s_src == 0 && p.C1 > 0) s_src = 1;
                else { s_src = 0; ++s_tap; }
            }
        }
        ++st_tile;
    };
    char* const dma0 = smem + wave * 1024;                   // this wave's 1 KiB landing row inside a piece half
    char* const trash = smem + CF::SMEM + wave * 1024;       // past-the-end copies land here, clear of ring and epilogue
    auto stageX = [&](int h, int bo) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int a = 2 * h + sl;
            pt_glds16(aptr[a] + (((avalid >> a) & 1u) ? ci_cur : 0), a_past ? trash : dma0 + bo + h * PIECE + sl * 8192);
        }
    };
    auto stageW = [&](int h, int bo, int kt) {
        const int ko = (kt < nk ? kt : nk - 1) * BK;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
            pt_glds16(wbase + (woff[2 * h + sl] + ko), kt < nk ? dma0 + bo + (2 + h) * PIECE + sl * 8192 : trash);
    };

    // ---------------- MFMA set-up
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int swz = frow >> 1;
    const int c0 = (fq ^ swz) * 16, c1 = ((fq + 4) ^ swz) * 16;            // byte offsets of the two 32-deep k halves
    const char* const xrd = smem + (wr * 32 + frow) * 128;
    const char* const wrd = smem + 2 * PIECE + (wc * 64 + frow) * 128;
    f32x4 acc[TN][TM];
    f16x8 X0[2][2], X1[2][2], Wf[4][2];                                    // [fragment][k half]

    // (no s_setprio around the MFMA clusters: see igemm10_kernel; the split wait of that kernel measured +-0 here)
#define IG8_READX(dst, h, bo)                                                                  \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                        \
            dst[i_][h_] = *(const f16x8*)(xrd + (bo) + (h) * PIECE + i_ * 2048 + (h_ ? c1 : c0));
#define IG8_READW(h, bo)                                                                       \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                        \
            Wf[i_][h_] = *(const f16x8*)(wrd + (bo) + (h) * PIECE + i_ * 2048 + (h_ ? c1 : c0));
#define IG8_MMA_HALF(Xv, hx, hw, ks_)                                                          \
        _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_)                                        \
            _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                    \
                acc[(hw) * 4 + n_][(hx) * 2 + m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(    \
                    Wf[n_][ks_], Xv[m_][ks_], acc[(hw) * 4 + n_][(hx) * 2 + m_], 0, 0, 0);
#define IG8_MMA(Xv, hx, hw)                                                                    \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    IG8_MMA_HALF(Xv, hx, hw, 0)                                                                \
    IG8_MMA_HALF(Xv, hx, hw, 1)                                                                \
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- prologue: K tile 0 complete, the first three pieces of K tile 1 in flight
    advanceA(); stageX(0, 0); stageW(0, 0, 0); stageX(1, 0); stageW(1, 0, 0);
    advanceA(); stageX(0, BUF); stageW(0, BUF, 1); stageX(1, BUF);
    bias_init<CF, 14>(b4, acc);
    __builtin_amdgcn_s_waitcnt(0x0F76);                      // vmcnt(6)
    __builtin_amdgcn_s_barrier();
    ig_stamp(kp, wave, lane, 1);
    const bool late = wave >= 4;                             // the group that runs one barrier behind
    if (late) __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < nk; ++kt) {
        const int bo = (kt & 1) * BUF, bo1 = bo ^ BUF;
        // ---- phase 1
        IG8_READX(X0, 0, bo)
        __builtin_amdgcn_sched_barrier(0);
        IG8_READW(0, bo)
        stageW(1, bo1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC87F);                  // lgkmcnt(8): the X0 reads are done before anyone restages X0
        __builtin_amdgcn_s_barrier();
        IG8_MMA(X0, 0, 0)
        __builtin_amdgcn_s_barrier();
        // ---- phase 2
        IG8_READX(X1, 1, bo)
        advanceA();
        stageX(0, bo);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        IG8_MMA(X1, 1, 0)
        __builtin_amdgcn_s_barrier();
        // ---- phase 3
        IG8_READW(1, bo)
        stageW(0, bo, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        IG8_MMA(X1, 1, 1)
        __builtin_amdgcn_s_barrier();
        // ---- phase 4
        stageX(1, bo);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6): K tile kt+1 has landed (this wave's copies)
        __builtin_amdgcn_s_barrier();
        IG8_MMA(X0, 0, 1)
        __builtin_amdgcn_s_barrier();
    }
    if (!late) __builtin_amdgcn_s_barrier();
#undef IG8_READX
#undef IG8_READW
#undef IG8_MMA
#undef IG8_MMA_HALF
    ig_stamp(kp, wave, lane, 2);
    {
        // the tail's lane-derived values (row / column of every pass, side-input addresses) must not be hoisted above the K
        // loop, where every register is taken: make the lane id opaque here
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        igemm_epilogue<CF, VAR>(kp, acc, smem, m0, n0, wave, lane_t);
    }
    ig_stamp(kp, wave, lane, 3);
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // no LDS-DMA may outlive the wave
}

// ============================================================================ 256 x 320, 10-phase ping-pong main loop
// Every channel count of the SVD U-Net is a multiple of 320 and every pixel count of the 576 x 1024 workload a
// multiple of 256 x 63, so this tile wastes nothing on any level-0..2 layer.  8 waves as 4 (M) x 2 (N), 64 x 160 per
// wave = 160 accumulator registers; the wave's four pixel fragments stay in registers for the whole K tile and the
// weights come in five pieces W0..W4 of 64 rows (for each wave column its fragments 2j, 2j+1: one GEGLU pair), one
// phase of 16 MFMAs per piece.  Same ping-pong of the two wave groups as igemm8_kernel.
//   LDS: 2 buffers x [X0 | X1 (128 pixel rows each, 16 KiB) | W0 .. W4 (8 KiB each)] = 2 x 72 KiB.
//   Per K tile t (buffer b = t & 1):
//     phase 1: read X (8), W0 (4)   stage W3(t+1) -> b^1                 MFMA X x W0
//     phase 2: read W1              stage W4(t+1) -> b^1                 MFMA X x W1
//     phase 3: read W2              stage X0(t+2) -> b                   MFMA X x W2
//     phase 4: read W3              stage X1(t+2), W0(t+2) -> b          MFMA X x W3
//     phase 5: read W4              stage W1(t+2), W2(t+2) -> b          MFMA X x W4    vmcnt(7): tile t+1 landed
//   Every piece is overwritten at least two phases after its last read; the buffer is read one phase after the wait.
template <int VAR>
__global__ __launch_bounds__(512, 2) void igemm10_kernel(const KParams kp) {
    using CF = Cfg<4, 2, 4, 10>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TM = 4, TN = 10, BM = 256, BN = 320;
    constexpr int XP = 16384, WP = 8192, BUF = 2 * XP + 5 * WP;
    const pt_igemm_params& p = kp.p;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // Split-K (small-M layers: level 3 has 64 tiles of 256 x 320 for 256 CUs): the XCD-contiguous id space is cut into
    // `splits` runs of one full tile grid each, so the workgroups an XCD runs together work on the same K slice and
    // share its weight columns through that L2 (the unsplit launch made every XCD stream the whole weight matrix).
    const int ntiles = kp.tiles_m * kp.tiles_n;
    const int bid_all = pt_xcd_remap(blockIdx.x, gridDim.x);
    const int split = bid_all / ntiles, bid = bid_all - split * ntiles;
    const int gsz = kp.gm * kp.tiles_n, grp = bid / gsz, first_m = grp * kp.gm;
    const int gm = min(kp.gm, kp.tiles_m - first_m), within = bid - grp * gsz;
    const int tile_m = first_m + within % gm, tile_n = within / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f16x4 b4[TN];
    bias_issue<CF>(kp, n0, wave, lane, b4);
    ig_stamp(kp, wave, lane, 0);

    // ---------------- staging set-up.  X slot a = 2 h + s: pixel row a*64 + (t >> 3); W piece j: one copy per thread
    const int csrc = (t & 7) ^ ((t >> 4) & 7);
    const int HWo = p.Hout * p.Wout;
    int iyx[4], pix0[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int m = m0 + a * 64 + (t >> 3);
        if (m < p.M) {
            int img = m, oy = 0, ox = 0;                     // linear layers (one-pixel images) skip the two divisions
            if (HWo != 1) {
                img = m / HWo;
                const int rem = m - img * HWo;
                oy = rem / p.Wout; ox = rem - oy * p.Wout;
            }
            iyx[a] = ((oy * p.stride - p.pad_h) << 16) | ((kp.foldx ? 0 : ox * p.stride - p.pad_w) & 0xffff);
            pix0[a] = img * p.Hin * p.Win + (kp.foldx ? ox : 0);
        } else {
            iyx[a] = (int)0xC0000000; pix0[a] = 0;
        }
    }
    const int Hlim = p.upsample2x ? 2 * p.Hin : p.Hin, Wlim = p.upsample2x ? 2 * p.Win : p.Win;
    const f16* zsrc = kp.zeros + (lane & 7) * 8;
    // weight row of this thread's chunk in piece j: n0 + wc*160 + (2j + f)*16 + r with (wc, f, r) from LDS row t >> 3
    const int wlr = t >> 3;
    const int wrow0 = n0 + (wlr >> 5) * 160 + ((wlr >> 4) & 1) * 16 + (wlr & 15);
    int woff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        int wrow = wrow0 + 32 * j;
        if (wrow >= kp.npad) wrow = kp.npad - 1;
        woff[j] = wrow * p.Kpad + csrc * 8;
    }
    const f16* wbase = (const f16*)p.w;
    const int nk_all = p.Kpad / BK;
    const int per_split = (nk_all + kp.splits - 1) / kp.splits;
    const int kbeg = split * per_split;                      // this workgroup's K tiles: [kbeg, kbeg + nk)
    const int nk = min(per_split, nk_all - kbeg);

    const f16* aptr[4];
    unsigned avalid = 0;
    int s_tap = 0, s_src = 0, s_ci = 0, ci_cur = 0, st_tile = 0;
    bool a_past = false, a_fresh = true;                     // a_fresh: the source pointers have not been set yet
    if (kbeg > 0) {                                          // position of K tile kbeg inside the (tap, source, channel) walk
        const int Ct = p.C0 + p.C1, k = kbeg * BK;
        s_tap = k / Ct;
        const int c = k - s_tap * Ct;
        s_src = c >= p.C0 ? 1 : 0;
        s_ci = s_src ? c - p.C0 : c;
    }
    auto advanceA = [&]() {
        if (st_tile >= nk) {
            // Past the last K tile the X copies carry no operand data: aim them at this wave's block of the residual
            // (first dummy tile) and of the blend input (second), one 128-byte line per lane, so the epilogue's side
            // loads - issued ~2 K tiles later - hit L2 instead of paying an HBM round trip per row pass.
            const f16* side = st_tile == nk ? (const f16*)p.res : (const f16*)p.blend;
            const int ldside = st_tile == nk ? p.ldr : p.ldb;
            const int nw = p.act == 1 ? TN * 8 : TN * 16;                              // this wave's output columns
            const int wcol = p.act == 1 ? (n0 + (wave & 1) * TN * 16) / 2 : n0 + (wave & 1) * TN * 16;
            const int nout = p.act == 1 ? p.N / 2 : p.N;
            const int mrow = min(m0 + (wave >> 1) * 64 + lane, p.M - 1);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int c = wcol + min(a * 64, nw - 8);
                aptr[a] = (side && c + 8 <= nout) ? side + ((size_t)mrow * ldside + c) : zsrc;
            }
            avalid = 0; ci_cur = 0; a_past = true;
        } else {
            if (s_ci == 0 || a_fresh) {
                a_fresh = false;
                const int ky = s_tap / p.KW, kx = s_tap - ky * p.KW;
                const f16* src = s_src ? (const f16*)p.x1 : (const f16*)p.x0;
                const int ld = s_src ? p.ld1 : p.ld0;
                avalid = 0;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    int iy = (iyx[a] >> 16) + ky, ix = (int)(short)(iyx[a] & 0xffff) + kx;
                    const bool ok = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
                    if (p.upsample2x) { iy >>= 1; ix >>= 1; }
                    aptr[a] = ok ? src + ((size_t)(pix0[a] + iy * p.Win + ix) * ld + csrc * 8) : zsrc;
                    avalid |= ok ? (1u << a) : 0u;
                }
            }
            ci_cur = s_ci;
            s_ci += BK;
            if (s_ci == (s_src ? p.C1 : p.C0)) {
                s_ci = 0;
                if (s_src == 0 && p.C1 > 0) s_src = 1;
                else { s_src = 0; ++s_tap; }
            }
        }
        ++st_tile;
    };
    char* const dma0 = smem + wave * 1024;
    char* const trash = smem + CF::SMEM + wave * 1024;
    auto stageX = [&](int h, int bo) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int a = 2 * h + sl;
            pt_glds16(aptr[a] + (((avalid >> a) & 1u) ? ci_cur : 0), a_past ? trash : dma0 + bo + h * XP + sl * 8192);
        }
    };
    auto stageW = [&](int j, int bo, int kt) {
        const int ko = (kbeg + (kt < nk ? kt : nk - 1)) * BK;
        pt_glds16(wbase + (woff[j] + ko), kt < nk ? dma0 + bo + 2 * XP + j * WP : trash);
    };

    // ---------------- MFMA set-up
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int swz = frow >> 1;
    const int c0 = (fq ^ swz) * 16, c1 = ((fq + 4) ^ swz) * 16;
    const char* const xrd = smem + (wr * 64 + frow) * 128;
    const char* const wrd = smem + 2 * XP + (wc * 32 + frow) * 128;
    f32x4 acc[TN][TM];
    f16x8 Xf[4][2], Wf[2][2];                                              // [fragment][k half]

    // Fragment reads are issued k-half-major and the wait behind the barrier is split: the first 8 MFMAs of a phase need
    // only the first k half of every fragment and start while the second halves are still on their way.  No s_setprio around
    // the MFMA clusters: with it the co-resident wave's fragment reads and copies started late (A/B on one box,
    // profiles/r02/igemm_mainloop_ab.txt: -1.5 ... -3.7 % on the long- and mid-K shapes for both changes together).
#define IG10_READW(j, bo)                                                                      \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                        \
            Wf[i_][h_] = *(const f16x8*)(wrd + (bo) + (j) * WP + i_ * 2048 + (h_ ? c1 : c0));
#define IG10_MMA_HALF(j, ks_)    /* m-major: consecutive MFMAs alternate the weight fragment (-1 % against n-major) */ \
        _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_)                                        \
            _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_)                                    \
                acc[2 * (j) + n_][m_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                \
                    Wf[n_][ks_], Xf[m_][ks_], acc[2 * (j) + n_][m_], 0, 0, 0);
#define IG10_PHASE_END(j)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    __builtin_amdgcn_s_barrier();                                                              \
    __builtin_amdgcn_s_waitcnt((j) == 0 ? 0xC67F : 0xC27F);   /* lgkmcnt(6 | 2): the first k halves */ \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    IG10_MMA_HALF(j, 0)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    IG10_MMA_HALF(j, 1)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    __builtin_amdgcn_s_barrier();

    // ---------------- prologue: K tile 0 complete, the first seven copies of K tile 1 in flight
    advanceA(); stageX(0, 0); stageX(1, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) stageW(j, 0, 0);
    advanceA(); stageX(0, BUF); stageX(1, BUF); stageW(0, BUF, 1); stageW(1, BUF, 1); stageW(2, BUF, 1);
    bias_init<CF, 16>(b4, acc);
    __builtin_amdgcn_s_waitcnt(0x0F77);                      // vmcnt(7)
    __builtin_amdgcn_s_barrier();
    ig_stamp(kp, wave, lane, 1);
    const bool late = wave >= 4;
    if (late) __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < nk; ++kt) {
        const int bo = (kt & 1) * BUF, bo1 = BUF - bo;
        // ---- phase 1 (reads: first k half of X and W0, then the second)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Xf[i][h] = *(const f16x8*)(xrd + bo + i * 2048 + (h ? c1 : c0));
#pragma unroll
            for (int i = 0; i < 2; ++i) Wf[i][h] = *(const f16x8*)(wrd + bo + i * 2048 + (h ? c1 : c0));
        }
        stageW(3, bo1, kt + 1);
        IG10_PHASE_END(0)
        // ---- phase 2
        IG10_READW(1, bo)
        stageW(4, bo1, kt + 1);
        IG10_PHASE_END(1)
        // ---- phase 3
        IG10_READW(2, bo)
        advanceA();
        stageX(0, bo);
        IG10_PHASE_END(2)
        // ---- phase 4
        IG10_READW(3, bo)
        stageX(1, bo);
        stageW(0, bo, kt + 2);
        IG10_PHASE_END(3)
        // ---- phase 5
        IG10_READW(4, bo)
        stageW(1, bo, kt + 2);
        stageW(2, bo, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F77);  // vmcnt(7): K tile kt+1 has landed (this wave's copies)
        IG10_PHASE_END(4)
    }
    if (!late) __builtin_amdgcn_s_barrier();
#undef IG10_READW
#undef IG10_PHASE_END
#undef IG10_MMA_HALF
    ig_stamp(kp, wave, lane, 2);
    if constexpr (VAR == V_SPLITK) {
        // split-K: raw fp32 partial sums in the accumulators' own layout (a lane's 4 consecutive channels of one pixel =
        // one 16-byte store; a wave row covers 64 contiguous bytes).  Bias, activation and the side inputs belong to
        // splitk_reduce_kernel, which adds the slabs in a fixed order.
        float* wsp = kp.ws + (size_t)split * p.M * p.N;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m0 + wr * 64 + mi * 16 + frow;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wc * 160 + ni * 16 + 4 * fq;
                if (m < p.M && n < p.N) *(f32x4*)(wsp + (size_t)m * p.N + n) = acc[ni][mi];
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
    } else {
        int lane_t = lane;                                   // (opaque: keeps the tail's lane-derived values below the K loop)
        asm volatile("" : "+v"(lane_t));
        igemm_epilogue<CF, VAR>(kp, acc, smem, m0, n0, wave, lane_t);
        ig_stamp(kp, wave, lane, 3);
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // no LDS-DMA may outlive the wave
    }
}


// Second half of a split-K product: out = epilogue(sum over slabs, in slab order).  One thread per (pixel, 8 channels).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const KParams kp) {
    const pt_igemm_params& p = kp.p;
    const int n8 = p.N >> 3;
    const long long total = (long long)p.M * n8;
    const float oscale0 = p.out_scale;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / n8), c0 = (int)(i - (long long)m * n8) * 8;
        float v[8];
        {
            const f16x8 b = p.bias ? *(const f16x8*)((const f16*)p.bias + c0) : (f16x8){};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)b[j];
        }
        {
            const float* w = kp.ws + (size_t)m * p.N + c0;
            const size_t slab = (size_t)p.M * p.N;
            int s = 0;
            for (; s + 3 < kp.splits; s += 4) {              // eight loads in flight, added in slab order
                f32x4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[u] = *(const f32x4*)(w + (s + u) * slab); b[u] = *(const f32x4*)(w + (s + u) * slab + 4); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[0] += a[u][0]; v[1] += a[u][1]; v[2] += a[u][2]; v[3] += a[u][3];
                    v[4] += b[u][0]; v[5] += b[u][1]; v[6] += b[u][2]; v[7] += b[u][3];
                }
            }
            for (; s < kp.splits; ++s) {
                const f32x4 a = *(const f32x4*)(w + s * slab), b = *(const f32x4*)(w + s * slab + 4);
                v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
                v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
            }
        }
        if (p.act == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pt_silu(v[j]);
        }
        float r[8] = {};
        if (p.res) {
            const f16x8 rh = *(const f16x8*)((const f16*)p.res + (size_t)m * p.ldr + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (float)rh[j];
            if (p.res_lo) {                                  // the residual as an fp16 pair
                const f16x8 rl = *(const f16x8*)((const f16*)p.res_lo + (size_t)m * p.ldr + c0);
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] += (float)rl[j];
            }
        }
        if (p.res && !p.res_post) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
        }
        if (p.vec) {
            const f16x8 e = *(const f16x8*)((const f16*)p.vec + (size_t)vec_index(p, m) * p.ldv + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)e[j];
        }
        if (p.blend) {
            const f16x8 e = *(const f16x8*)((const f16*)p.blend + (size_t)m * p.ldb + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p.alpha * (float)e[j] + (1.0f - p.alpha) * v[j];
        }
        const float oscale = oscale0 * (c0 < p.cs_cols ? p.cs_scale : 1.0f);
        f16x8 o, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[j] * oscale + ((p.res && p.res_post) ? r[j] : 0.f);
            o[j] = (f16)x;
            lo[j] = (f16)(x - (float)o[j]);
        }
        *(f16x8*)((f16*)p.out + (size_t)m * p.ldo + c0) = o;
        if (p.out_lo) *(f16x8*)((f16*)p.out_lo + (size_t)m * p.ldo + c0) = lo;
    }
}

using CfgBig = Cfg<4, 2, 4, 8>;     // 256 x 256: 64 x 128 per wave
using CfgW320 = Cfg<2, 4, 4, 5>;    // 128 x 320 (never used with GEGLU: odd TN)
using CfgSmall = Cfg<2, 2, 4, 4>;   // 128 x 128
using CfgN160 = Cfg<4, 1, 2, 10, 72 * 1024>;   // 128 x 160: 4 waves of 32 x 160, 72 KiB of LDS -> 2 workgroups per CU
using CfgN32 = Cfg<4, 1, 4, 2>;                // 256 x 32: 4 waves of 64 x 32 - the condition encoder's 16- / 32-channel convolutions over
                                               // 2.6 M pixels (training runs them forward AND backward every step) wasted 3/4 of a 128-wide tile

// Pick the tile configuration by modelled cycles: rounds of co-resident tiles x (prologue + K tiles x loop cycles per
// K tile + epilogue), with the per-configuration constants read off the s_memtime stamps / sweeps in
// profiles/r01/igemm_stamps_v14.txt and igemm_cfg_sweep_v14.txt.  `fast` = channel-aligned K tiles (the pipelined
// 256 x 256 and 256 x 320 kernels need it).
double g_last_tile_cycles = 0.0, g_last_rounds = 0.0;       // of the configuration choose_cfg returned (model cycles per tile, rounds)
int choose_cfg(int M, int N, int nk, int act, bool has_side, bool fast) {
    struct Opt { int bm, bn, slots; double pro, loop, epi, epi_geglu, epi_side; };
    static const Opt pipelined[5] = {
        {256, 256, 256, 5000, 2650, 10500, 8700, 4000},      // 8-phase ping-pong
        {128, 320, 256, 3000, 2330, 9000, 9000, 4000},       // plain 2-stage loop
        {128, 128, 512, 3000, 1900, 7000, 6000, 2000},       // plain loop, 2 workgroups per CU (1900 with every slot busy)
        {256, 320, 256, 5500, 3300, 16000, 11720, 9000},     // 10-phase ping-pong (GEGLU tail: calibrated so that K = 320 stays on
                                                             // 256 x 256 and K = 640 / 1280 move here, profiles/r03/igemm_cfg_sweep_duo_v*.txt;
                                                             // plain tail 19000 -> 16000 in round 6: 64512 x 1920 x 640 belongs here, -11 %)
        {128, 160, 512, 3000, 2430, 8000, 7000, 3000},       // plain loop, 32 x 160 per wave, 2 workgroups per CU (2150 -> 2430 in round 6: its K tile
                                                             // costs 0.68 of the 10-phase kernel's per output, not 0.77 - 80640 x 320 x 2880 / 5760, the
                                                             // level-0 convolutions of the 320 x 576 workload, ran 13 / 20 % slower here than on 256 x 320)
    };   // round 6: with these constants and the split rule below the model picks the fastest measured configuration for 59 of the 60
         // shapes of profiles/r06/igemm_cfg_sweep_{L,M}_r06c.txt (tools/cfg_model_check.py restates the model and scores it, CPU only)
    int best = 2; double best_t = 1e300;
    for (int i = 0; i < 5; ++i) {
        if (i == 1 && act == 1) continue;                    // odd TN: no GEGLU pairs
        if (i == 3 && !fast) continue;
        const Opt& o = pipelined[i];
        const double tiles = (double)((M + o.bm - 1) / o.bm) * ((N + o.bn - 1) / o.bn);
        const double rounds = (double)(long long)((tiles + o.slots - 1) / o.slots);
        const double loop = (i == 0 && !fast) ? 3400 : o.loop;                 // the plain 256 x 256 loop
        const double tile = o.pro + nk * loop + (act == 1 ? o.epi_geglu : o.epi) + (has_side ? o.epi_side : 0);
        // a wave whose column range is not whole finishes element by element (igemm_tail): N = 960 on the 256-wide tile
        // (wave width 128) ran 20 % slower than on the 320-wide one
        const int wave_w = o.bn / (i == 0 || i == 3 ? 2 : (i == 1 ? 4 : (i == 2 ? 2 : 1)));
        const double ragged = (N > wave_w && N % wave_w != 0) ? 1.3 : 1.0;
        const double t = rounds * tile * ragged;
        if (t < best_t * 0.999) { best_t = t; best = i; g_last_tile_cycles = tile; g_last_rounds = tiles / o.slots; }
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
    static bool attr_done[64] = {};                          // the opt-in is per device
    const int dev = pt_device();
    if (!attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<CF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
        (void)hipFuncSetAttribute((const void*)igemm_kernel<CF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
        attr_done[dev] = true;
    }
    const unsigned nblk = (unsigned)(kp.tiles_m * kp.tiles_n);
    if (fast) hipLaunchKernelGGL((igemm_kernel<CF, true>), dim3(nblk), dim3(CF::NT), CF::SMEM, s, kp);
    else      hipLaunchKernelGGL((igemm_kernel<CF, false>), dim3(nblk), dim3(CF::NT), CF::SMEM, s, kp);
}

typedef void (*pipe_kernel_t)(const KParams);

void launch8(const KParams& kp, hipStream_t s) {
    static const pipe_kernel_t table[V_COUNT] = {igemm8_kernel<V_P0>, igemm8_kernel<V_P1>, igemm8_kernel<V_P2>, igemm8_kernel<V_EW>,
                                                 igemm8_kernel<V_W0>, igemm8_kernel<V_W1>, igemm8_kernel<V_W2>, igemm8_kernel<V_G0>,
                                                 igemm8_kernel<V_GEW>, nullptr};
    static bool attr_done[64][V_COUNT] = {};
    const int dev = pt_device(), var = tail_variant(kp.p);
    if (!attr_done[dev][var]) {
        (void)hipFuncSetAttribute((const void*)table[var], hipFuncAttributeMaxDynamicSharedMemorySize, CfgBig::SMEM + TRASH);
        attr_done[dev][var] = true;
    }
    hipLaunchKernelGGL(table[var], dim3((unsigned)(kp.tiles_m * kp.tiles_n)), dim3(512), CfgBig::SMEM + TRASH, s, kp);
}

using CfgT320 = Cfg<4, 2, 4, 10>;   // 256 x 320: 64 x 160 per wave (igemm10_kernel)
void launch10(const KParams& kp, hipStream_t s) {
    static const pipe_kernel_t table[V_COUNT] = {igemm10_kernel<V_P0>, igemm10_kernel<V_P1>, igemm10_kernel<V_P2>, igemm10_kernel<V_EW>,
                                                 igemm10_kernel<V_W0>, igemm10_kernel<V_W1>, igemm10_kernel<V_W2>, igemm10_kernel<V_G0>,
                                                 igemm10_kernel<V_GEW>, igemm10_kernel<V_SPLITK>};
    static bool attr_done[64][V_COUNT] = {};
    const int dev = pt_device(), var = kp.ws ? V_SPLITK : tail_variant(kp.p);
    if (!attr_done[dev][var]) {
        (void)hipFuncSetAttribute((const void*)table[var], hipFuncAttributeMaxDynamicSharedMemorySize, CfgT320::SMEM + TRASH);
        attr_done[dev][var] = true;
    }
    hipLaunchKernelGGL(table[var], dim3((unsigned)(kp.tiles_m * kp.tiles_n * kp.splits)), dim3(512), CfgT320::SMEM + TRASH, s, kp);
}

int g_force_cfg = -1;

}  // namespace

unsigned long long* g_stamps = nullptr;
long long g_stamps_cap = 0;
extern "C" int pt_igemm_set_stamps(void* buf, int64_t capacity) {
    g_stamps = (unsigned long long*)buf;
    g_stamps_cap = buf ? capacity : 0;
    return 0;
}

// test hook: force a tile configuration (0 = 256x256, 1 = 128x320, 2 = 128x128, 3 = 256x320, 4 = 128x160, 5 = 256x32, -1 = automatic)
extern "C" int pt_igemm_force_config(int32_t cfg) {
    PT_CHECK(cfg >= -1 && cfg <= 5, "pt_igemm_force_config: %d", cfg);
    g_force_cfg = cfg;
    return 0;
}

// Split-K plan for small-M problems (level 3: 64 tiles of 256 x 320 on 256 CUs; 20 tiles at the 320 x 576 workload):
// `splits` workgroups per output tile, each >= 6 K tiles, until the launch has about one workgroup per CU.  Only the
// 256 x 320 kernel implements it (channel-aligned K, no GEGLU, 16-byte-aligned rows for the reducer).  Short reductions
// (K < 3072) stay un-split: there the 128-row tiles already give one workgroup per CU and the two launches + fp32 slab round
// trip of split-K lose to them (2520 x 1280 x 1280: 38.7 us split, 19.7 us on 128 x 128 tiles; tools/micro/igemm_cfg_sweep.py).
static int plan_splits(const pt_igemm_params& p, bool fast, bool vec_ok) {
    static const int off = getenv("PT_IGEMM_NO_SPLITK") ? atoi(getenv("PT_IGEMM_NO_SPLITK")) : 0;
    if (off || !fast || !vec_ok || p.act == 1 || p.N % 8 != 0) return 1;
    const int tiles = ((p.M + 255) / 256) * ((p.N + 319) / 320), nk = p.Kpad / BK;
    if (tiles > 128 || nk < 48) return 1;
    // round 6: where the 128 x 128 tiles alone give (nearly) one workgroup per slot - >= 300 tiles for the 512 slots - reductions up to
    // K = 5120 stay un-split: 4032 x 1280 x 3840 55.6 us against 66.0 split in 4, 5040 x 1280 x 3840 62.1 against 75.6, x 5120 78.4
    // against 87.1 (profiles/r06/igemm_cfg_sweep_*_r06c.txt); at K = 11520 and for 1260 rows (100 tiles) split-K still wins
    if ((long long)((p.M + 127) / 128) * ((p.N + 127) / 128) >= 300 && nk <= 80) return 1;
    int s = 256 / tiles;
    if (s > nk / 6) s = nk / 6;
    if (s > 16) s = 16;
    if (s < 2) return 1;
    const int per = (nk + s - 1) / s;                        // the kernel deals ceil(nk / s) K tiles per split: drop the
    s = (nk + per - 1) / per;                                // splits that would get none (nk = 180, s = 16 -> 15)
    return s < 2 ? 1 : s;
}

extern "C" int64_t pt_igemm_splitk_ws_bytes(const pt_igemm_params* pp) {
    const pt_igemm_params& p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.Kpad <= 0 || p.C0 <= 0) return 0;
    const int Ctot = p.C0 + p.C1;
    const bool fast = (Ctot % BK == 0) && (p.C0 % BK == 0) && (p.Kpad == p.K);
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool vec_ok = !p.out_f32 && (p.N % 8 == 0) && (p.ldo % 8 == 0) && al16(p.out) && (!p.res || (p.ldr % 8 == 0 && al16(p.res))) && al16(p.res_lo) && al16(p.out_lo) &&
                        (!p.vec || (p.ldv % 8 == 0 && al16(p.vec))) && (!p.blend || (p.ldb % 8 == 0 && al16(p.blend)));
    if (g_force_cfg >= 0 && g_force_cfg != 3) return 0;
    const int s = plan_splits(p, fast, vec_ok);
    return s > 1 ? (int64_t)s * p.M * p.N * 4 : 0;
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
    // the kernel packs each output pixel's top-left tap as two signed 16-bit coordinates; one-column kernels need no x
    // coordinate at all (foldx) and may be as wide as the 31-bit pixel index allows
    const bool foldx = p.KW == 1 && p.pad_w == 0 && p.stride == 1 && !p.upsample2x && p.Wout == p.Win;
    PT_CHECK((long long)p.Hout * p.stride + p.KH < 32000 && (foldx || (long long)p.Wout * p.stride + p.KW < 32000),
             "pt_igemm_f16: output extent %d x %d too large (a linear layer is Nimg = M, H = W = 1)", p.Hout, p.Wout);
    PT_CHECK((long long)p.Nimg * p.Hin * p.Win < (1ll << 31), "pt_igemm_f16: more than 2^31 input pixels");
    PT_CHECK(!(p.upsample2x && p.stride != 1), "pt_igemm_f16: upsample2x needs stride 1");
    PT_CHECK(p.act == 0 || p.act == 2 || (p.act == 1 && p.N % 32 == 0), "pt_igemm_f16: act must be 0, 1 (GEGLU, N %% 32 == 0) or 2 (SiLU)");
    PT_CHECK(p.vec_mode == 0 || p.vec, "pt_igemm_f16: vec_mode without vec");
    PT_CHECK(pt_zero_page(), "pt_igemm_f16: zero page not set (pt_set_zero_page)");
    KParams kp;
    kp.p = p;
    if (!p.vec) kp.p.vec_mode = 0;
    kp.zeros = (const f16*)pt_zero_page();
    kp.npad = (p.N + 127) / 128 * 128;
    kp.stamps = g_stamps; kp.stamps_cap = g_stamps_cap;
    kp.foldx = foldx ? 1 : 0;
    const int nout = p.act == 1 ? p.N / 2 : p.N;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    PT_CHECK(!(p.res_post && !p.res), "pt_igemm_f16: res_post without res");
    PT_CHECK(!(p.res_lo && !p.res), "pt_igemm_f16: res_lo without res");
    PT_CHECK(!(p.out_lo && (p.out_f32 || p.act == 1)), "pt_igemm_f16: out_lo needs an fp16, non-GEGLU output");
    PT_CHECK(p.cs_cols >= 0 && p.cs_cols % 8 == 0, "pt_igemm_f16: cs_cols=%d must be a non-negative multiple of 8", p.cs_cols);
    kp.vec_ok = !p.out_f32 && (nout % 8 == 0) && (p.ldo % 8 == 0) && al16(p.out) && (!p.res || (p.ldr % 8 == 0 && al16(p.res))) && al16(p.res_lo) && al16(p.out_lo) &&
                (!p.vec || (p.ldv % 8 == 0 && al16(p.vec))) && (!p.blend || (p.ldb % 8 == 0 && al16(p.blend)));
    const bool fast = (Ctot % BK == 0) && (p.C0 % BK == 0) && (p.Kpad == p.K);
    const int force = g_force_cfg;
    int cfg = force >= 0 ? force : choose_cfg(p.M, p.N, p.Kpad / BK, p.act, p.res || p.blend, fast);
    int splits = (force < 0 || force == 3) ? plan_splits(p, fast, kp.vec_ok) : 1;
    if (splits > 1 && !(p.splitk_ws && p.splitk_ws_bytes >= (int64_t)splits * p.M * p.N * 4)) splits = 1;   // no workspace offered
    if (splits > 1) cfg = 3;
    if (force < 0 && p.N <= 96 && p.act != 1 && splits == 1) cfg = 5;      // tools/micro/igemm_n32_sweep.py: 2.6-4 x on N = 16 / 32, 1.5 x on 96
    if (cfg == 3 && !fast) cfg = p.act == 1 ? 0 : 1;         // the 256x320 kernel has no generic-K gather
    PT_CHECK(!(cfg == 5 && p.act == 1), "pt_igemm_f16: the 256x32 configuration does not support GEGLU");
    PT_CHECK(!(cfg == 1 && p.act == 1), "pt_igemm_f16: the 128x320 configuration does not support GEGLU");
    const int bm = (cfg == 0 || cfg == 3 || cfg == 5) ? 256 : 128, bn = cfg == 0 ? 256 : (cfg == 2 ? 128 : (cfg == 4 ? 160 : (cfg == 5 ? 32 : 320)));
    kp.tiles_m = (p.M + bm - 1) / bm;
    kp.tiles_n = (p.N + bn - 1) / bn;
    PT_CHECK((long long)kp.tiles_m * kp.tiles_n < (1ll << 31), "pt_igemm_f16: grid too large");
    {
        static const int gm_env = getenv("PT_IGEMM_GROUP_M") ? atoi(getenv("PT_IGEMM_GROUP_M")) : 0;   // tuning override
        const double in_px = p.upsample2x ? p.M / 4.0 : (double)p.M * p.stride * p.stride;
        kp.gm = gm_env > 0 ? (gm_env < kp.tiles_m ? gm_env : kp.tiles_m)
                           : choose_group(kp.tiles_m, kp.tiles_n, in_px * Ctot * 2.0, (double)p.N * p.K * 2.0, (cfg == 2 || cfg == 4) ? 64 : 32);
    }
    hipStream_t s = (hipStream_t)stream;
    kp.ws = nullptr; kp.splits = 1;
    {
        static const int dbg = getenv("PT_IGEMM_DBG") ? atoi(getenv("PT_IGEMM_DBG")) : 0;
        kp.dbg = dbg;
    }
    pt_prof_begin(0, s, 2.0 * (double)p.M * (double)p.N * (double)p.K);
    static const int pipe8 = getenv("PT_IGEMM_PIPE8") ? atoi(getenv("PT_IGEMM_PIPE8")) : 1;   // 0: the plain 256x256 loop
    if (splits > 1) {
        KParams k1 = kp;                                     // pass 1: bare products into the fp32 slabs
        k1.ws = (float*)p.splitk_ws; k1.splits = splits;
        k1.p.bias = nullptr; k1.p.res = nullptr; k1.p.res_lo = nullptr; k1.p.out_lo = nullptr; k1.p.vec = nullptr; k1.p.blend = nullptr; k1.p.vec_mode = 0; k1.p.act = 0;
        launch10(k1, s);
        KParams k2 = kp;                                     // pass 2: ordered sum + the whole epilogue
        k2.ws = (float*)p.splitk_ws; k2.splits = splits;
        const long long work = (long long)p.M * (p.N / 8);
        long long blocks = (work + 255) / 256;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, k2);
    } else if (cfg == 3) launch10(kp, s);
    else if (cfg == 0 && fast && pipe8) launch8(kp, s);
    else if (cfg == 0) launch<CfgBig>(kp, fast, s);
    else if (cfg == 1) launch<CfgW320>(kp, fast, s);
    else if (cfg == 4) launch<CfgN160>(kp, fast, s);
    else if (cfg == 5) launch<CfgN32>(kp, fast, s);
    else launch<CfgSmall>(kp, fast, s);
    pt_prof_end(0, s);
    PT_LAUNCH_CHECK("pt_igemm_f16");
    return 0;
}
