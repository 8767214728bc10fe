// The parts of the implicit-GEMM kernels that the fused feed-forward kernel (ffn.hip) shares: launch parameters, tile / tail
// geometry, bias-in-the-accumulators, the row-wise fused tail and the epilogue that picks its variant.  Included by igemm.hip and
// ffn.hip; everything lives in an anonymous namespace (one copy per translation unit).
#pragma once
#include <type_traits>

#include "pt_common.h"

namespace {

constexpr int BK = 64;
constexpr int TRASH = 8 * 1024;       // LDS landing rows (1 KiB per wave) of the pipelined kernels' past-the-end copies

struct KParams {
    pt_igemm_params p;
    const f16* zeros;
    int tiles_m, tiles_n;
    int npad;       // rows of the packed weight image
    int vec_ok;     // 16-byte epilogue path allowed
    int gm;         // M tiles per rasterisation group (see the kernel's tile-order comment)
    unsigned long long* stamps;   // tuning: s_memtime stamps (pt_igemm_set_stamps), usually null
    long long stamps_cap;
    float* ws;                    // split-K: fp32 partial sums [splits][M][N] (igemm10_kernel only), else null
    int splits;                   // K tiles are dealt to `splits` workgroups per output tile (1 = off)
    int dbg;                      // tuning ablations (PT_IGEMM_DBG; results are wrong): 1 = no global stores, 2 = no epilogue
    int foldx;                    // one-column kernels (KW = 1, no x padding / stride / upsampling): the output column is folded
                                  // into the pixel base and the packed x coordinate stays 0, so the image may be wider than 16
                                  // bits (the VAE's (3,1,1) convolutions see the image (F, H*W): 589 824 columns at 576 x 1024)
};

// Row passes of the tail.  A store instruction costs the CU's store path 64 lane-clocks whether its lanes are live or
// not, so no lane should idle.  ROWS form (LPR | 64): a pass takes 64 / LPR whole rows, pass g moves DR rows down.
// COLS form (otherwise, when RH | 64): every lane owns ONE row of the chunk, 64 / RH lanes share a row and pass g moves
// them DC columns to the right - N = 320 k tiles (LPR 20, 16-row chunks): 5 passes of 64 lanes instead of 6 of 60;
// row validity, the row's side-input addresses and the row-vector index are then per lane, not per pass, and every
// pass is an immediate offset from one address.
constexpr int PT_TAIL_MIN_LP = 4;
template <int RH, int LPR>
struct PassGeom {
    static constexpr bool COLS = (64 % LPR != 0) && (64 % RH == 0) && (LPR % (64 / RH) == 0) && (64 / RH >= PT_TAIL_MIN_LP);
    static constexpr int LP = COLS ? 64 / RH : LPR;           // lanes side by side in a row
    static constexpr int RPP = 64 / LP;                       // rows per pass
    static constexpr int P = COLS ? LPR / LP : (RH + RPP - 1) / RPP;
    static constexpr int DR = COLS ? 0 : RPP, DC = COLS ? LP * 8 : 0;
};

// Geometry of the epilogue's LDS staging for a variant that is NTL accumulator blocks wide (igemm_tail): padded fp32
// rows, and the tallest chunk (TM*16 / TM*8 / TM*4 rows per wave) whose staging fits CAP bytes per workgroup.
template <int WAVES, int TM, int NTL, int CAP>
struct TailGeom {
    static constexpr int ELD = NTL * 16 + 4;                 // floats per staged row: the variant's width + 4 pad
    static constexpr int RH = (WAVES * TM * 16 * ELD * 4 <= CAP) ? TM * 16 : ((WAVES * TM * 8 * ELD * 4 <= CAP) ? TM * 8 : TM * 4);
    static constexpr int WAVE_BYTES = RH * ELD * 4;
    static_assert(RH % 16 == 0, "chunks are whole accumulator blocks");
};

template <int WM_, int WN_, int TM_, int TN_, int EPI_CAP_ = 144 * 1024>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = WM * WN * 64;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_SLOTS = BM * 8 / NT, B_SLOTS = BN * 8 / NT;
    static constexpr int EPI_CAP = EPI_CAP_;                  // bytes of LDS the epilogue's staging rows may take
    using TailFull = TailGeom<WM * WN, TM, TN, EPI_CAP_>;                           // plain variants
    using TailHalf = TailGeom<WM * WN, TM, (TN % 2 == 0 ? TN / 2 : TN), EPI_CAP_>;  // GEGLU: half as wide, twice as tall
    static constexpr int EPI_BYTES = WM * WN * (TailFull::WAVE_BYTES > TailHalf::WAVE_BYTES ? TailFull::WAVE_BYTES : TailHalf::WAVE_BYTES);
    static constexpr int SMEM = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile rows must divide over the threads");
    static_assert((NT / 16) % 8 == 0, "row swizzle must be slot-group independent");
};

__device__ __forceinline__ int vec_index(const pt_igemm_params& p, int m) {
    if (p.vec_mode == 1) return m / p.vG;
    return ((m / p.vFS) * p.vS + m % p.vS) % p.vB;
}

__device__ __forceinline__ void ig_stamp(const KParams& kp, int wave, int lane, int which) {
    if (kp.stamps && lane == 0) {
        const long long i = ((long long)blockIdx.x * 8 + wave) * 16 + which;
        if (i < kp.stamps_cap) kp.stamps[i] = __builtin_amdgcn_s_memtime();
        if (which == 0 && i + 15 < kp.stamps_cap)            // slot 15: where the wave ran (XCC_ID << 32 | HW_ID), for per-CU timelines
            kp.stamps[i + 15] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
}

// Bias folded into the accumulators' initial value: the loads are issued at the top of the kernel (bias_issue) and
// consumed after the prologue's LDS-DMA copies have been issued (bias_init: counted vmcnt), which removes TM*TN*4 adds
// per lane from the epilogue.
template <class CF>
__device__ __forceinline__ void bias_issue(const KParams& kp, int n0, int wave, int lane, f16x4 (&b4)[CF::TN]) {
    const f16* bias = (const f16*)kp.p.bias;
    const int wc = wave % CF::WN, fq = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < CF::TN; ++ni) {
        int nb = n0 + (wc * CF::TN + ni) * 16 + 4 * fq;
        if (nb > kp.npad - 4) nb = kp.npad - 4;
        b4[ni] = bias ? *(const f16x4*)(bias + nb) : (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    }
}
template <class CF, int NEWER>
__device__ __forceinline__ void bias_init(const f16x4 (&b4)[CF::TN], f32x4 (&acc)[CF::TN][CF::TM]) {
    // the bias loads are older than the NEWER LDS-DMA copies of the prologue issued since: a counted wait retires the
    // loads and leaves the copies in flight (their latency no longer queues behind the bias round trip)
    __builtin_amdgcn_s_waitcnt(((NEWER & 15) | ((NEWER >> 4) << 14)) | 0x0F70);
#pragma unroll
    for (int ni = 0; ni < CF::TN; ++ni) {
        const f32x4 b = {(float)b4[ni][0], (float)b4[ni][1], (float)b4[ni][2], (float)b4[ni][3]};
#pragma unroll
        for (int mi = 0; mi < CF::TM; ++mi) acc[ni][mi] = b;
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Row-wise fused tail of the epilogue for a wave whose first NTL 16-column accumulator blocks are valid: RH rows at a
// time go through LDS (fp32, padded rows), then each lane finishes 8 consecutive channels of one pixel: + residual,
// + broadcast row vector, AlphaBlender lerp, scale, one 16-byte store.  Every row segment written is >= 128
// contiguous bytes (64 for the GEGLU half-width of the 128-wide tiles).
// Store path.  A CU retires stores at ONE LANE PER CLOCK whatever their width (tools/micro/store_bw.hip: 16.8 B/clk/CU
// with 16-byte lanes, 7.8 with 8-byte lanes, idle lanes cost the same): a 256 x 320 fp16 tile is >= 10.2k cycles of
// store path, as much as four K tiles of MFMA work, and nothing in the workgroup overlaps it.  So every store
// instruction should carry 64 live 16-byte lanes (PassGeom above: N = 320 k tiles take 5 passes per 16-row chunk instead of
// 6 of 60 lanes), and the chunk is as tall as LDS allows for the variant's width (GEGLU rows are half as wide: 32 rows
// per chunk on the 256 x 320 tile, 5 store instructions per chunk instead of 2 x 3).
// Ordering of the side loads.  On gfx9 stores count in vmcnt like loads, so a load issued after a store cannot be
// consumed before that store has been acknowledged by memory: with "load - add - store" per row group every group
// paid a full store round trip (stamps: 10-11.6k cycles per 16-row chunk with a residual against 4.4k without).  The
// side inputs of chunk c+1 are therefore loaded during chunk c, pass by pass: the load of (chunk c+1, pass g) right
// after pass g of chunk c has consumed its registers and BEFORE that pass's store is issued; the wait that consumes it,
// a whole chunk later, is counted past every younger store.
// NS = number of side inputs the variant is compiled for (0, 1, 2; 3 = the element-wise path): the side registers are
// then sized exactly.
// Wide stream (WIDE / res_lo).  The tensors of the residual stream (resblock and transformer outputs, the shortcut) can
// be kept as an fp16 PAIR: out = fp16(v), out_lo = fp16(v - out); consumers that use the tensor as a GEMM operand or
// normalise it read `out` alone (exactly the fp16 tensor), the epilogue that adds it as a residual reads res + res_lo
// (side-input kind 4).  The one-rounding-per-block random walk of the stream - 0.98e-3 of the U-Net's 1.08e-3 rel-L2
// (profiles/r02/parity_ladder*.txt) - drops to 2^-22 per store.
template <class CF, int NTL, bool GEGLU, int NS, bool WIDE>
__device__ __forceinline__ void igemm_tail(const KParams& kp, f32x4 (&acc)[CF::TN][CF::TM], char* smem,
                                           int mrow0, int wcol0, int Nout, int wave, int lane) {
    // (GEGLU chunks half as tall - the stores of chunk c under the GELU arithmetic of chunk c + 1 - measured +-0.5 % on both
    // pipelined kernels: profiles/r03/igemm_geglu_chunk_height_ab.txt)
    using TG = TailGeom<CF::WM * CF::WN, CF::TM, NTL, CF::EPI_CAP>;
    constexpr int TM = CF::TM, RH = TG::RH, ELD = TG::ELD;
    static_assert(CF::WM * CF::WN * TG::WAVE_BYTES <= CF::SMEM, "tail staging must fit the kernel's LDS");
    constexpr int LPR = NTL * 2;                             // lanes per row, 8 columns each
    using PG = PassGeom<RH, LPR>;
    constexpr int NPASS = PG::P;
    constexpr int NCHUNK = TM * 16 / RH;
    constexpr int NSA = NS == 0 ? 1 : (NS > 2 ? 1 : NS);     // side register sets
    constexpr bool PREFETCH = NS >= 1 && NS <= 2;            // side inputs of chunk c+1 are loaded during chunk c
    const pt_igemm_params& p = kp.p;
    const int frow = lane & 15, fq = lane >> 4;
    float* E = (float*)(smem + wave * TG::WAVE_BYTES);
    const float alpha = p.alpha;
    const bool res_post = p.res_post != 0;                   // out = res + out_scale * t  (accumulate into `res`)
    f16* out = (f16*)p.out;
    f16* out_lo = (f16*)p.out_lo;
    // this lane's (row, first column) in pass g (PassGeom)
    const int r0 = lane / PG::LP, c0 = (lane % PG::LP) * 8;
    const bool lane_live = r0 < PG::RPP;                     // ROWS form with LPR not dividing 64: the last lanes idle
    auto slot = [&](int g, int& r, int& c8) { r = r0 + g * PG::DR; c8 = c0 + g * PG::DC; };
    // out_scale x the column scale of 8 columns (cs_cols % 8 == 0); recomputed where it is used: a live VGPR
    // for it made the 160-accumulator kernel spill, and the spill's reload waits vmcnt(0) = for every store in flight
    auto col_scale = [&](int col) { return p.out_scale * (col < p.cs_cols ? p.cs_scale : 1.0f); };
    // up to two side inputs in application order (residual, its low half, row vector, blend); kind 1 = add, 2 = add a
    // row vector, 3 = lerp, 4 = add (low half of the residual).  More (never used by the networks) takes the element-wise path.
    const f16* sp[2] = {nullptr, nullptr}; int sld[2] = {0, 0}, skind[2] = {0, 0}, ns = 0;
    if (NS <= 2) {
        if (p.res) { sp[ns] = (const f16*)p.res; sld[ns] = p.ldr; skind[ns++] = 1; }
        if (p.res_lo && ns < 2) { sp[ns] = (const f16*)p.res_lo; sld[ns] = p.ldr; skind[ns++] = 4; }
        if (p.vec && ns < 2) { sp[ns] = (const f16*)p.vec; sld[ns] = p.ldv; skind[ns++] = 2; }
        if (p.blend && ns < 2) { sp[ns] = (const f16*)p.blend; sld[ns] = p.ldb; skind[ns++] = 3; }
    }
    // wave-uniform: a wave whose width is not whole (ragged last N tile) takes the element-wise path - choose_cfg steers
    // clear of configurations whose wave width does not divide N
    const bool fastpath = kp.vec_ok && wcol0 + NTL * 16 <= Nout && NS <= 2;
    f16x8 side[NSA][NPASS];
    auto load_side = [&](int rc, int g) {                    // side inputs of pass g of chunk rc -> side[.][g]
#pragma unroll
        for (int a = 0; a < NSA; ++a) {
            if (a < NS) {
                int r, c8;
                slot(g, r, c8);
                const int m = min(mrow0 + rc * RH + r, p.M - 1);
                const size_t row = skind[a] == 2 ? (size_t)vec_index(p, m) : (size_t)m;
                side[a][g] = *(const f16x8*)(sp[a] + row * sld[a] + wcol0 + c8);
            }
        }
    };
    auto read_row = [&](int g, f32x4& v0, f32x4& v1) {       // this lane's 8 staged values of row pass g
        int r, c8;
        slot(g, r, c8);
        const float* e = E + min(r, RH - 1) * ELD + c8;
        v0 = *(const f32x4*)e; v1 = *(const f32x4*)(e + 4);
    };
    auto finish_row = [&](int gabs, int gside, const f32x4& v0, const f32x4& v1, f16x8& lo8) -> f16x8 {   // -> 8 fp16 outputs
        int r, c8;
        slot(gabs, r, c8);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int a = 0; a < NSA; ++a) {
            if (a < NS) {
                if (skind[a] == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = alpha * (float)side[a][gside][j] + (1.0f - alpha) * v[j];
                } else if (!((skind[a] == 1 || skind[a] == 4) && res_post)) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)side[a][gside][j];
                }
            }
        }
        const float oscale = col_scale(wcol0 + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= oscale;
        if (NS >= 1 && res_post && skind[0] == 1) {          // the residual (and its low half) lead the side inputs
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)side[0][gside][j];
            if (NS >= 2 && skind[NSA - 1] == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)side[NSA - 1][gside][j];
            }
        }
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)v[j];
        if constexpr (WIDE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) lo8[j] = (f16)(v[j] - (float)o[j]);
        }
        return o;
    };
    auto store_row = [&](int rc, int g, const f16x8& o, const f16x8& lo8) {
        int r, c8;
        slot(g, r, c8);
        const int m = mrow0 + rc * RH + r;
        if (lane_live && r < RH && m < p.M && !(kp.dbg & 1)) {
            const size_t off = (size_t)m * p.ldo + wcol0 + c8;
            *(f16x8*)(out + off) = o;
            if constexpr (WIDE) *(f16x8*)(out_lo + off) = lo8;
        }
    };
    if (PREFETCH && fastpath) {
#pragma unroll
        for (int g = 0; g < NPASS; ++g) load_side(0, g);
    }
#pragma unroll
    for (int rc = 0; rc < NCHUNK; ++rc) {
        // activation on the way into LDS (never in place: a three-way branch that rewrites 128-160 live accumulators
        // made the compiler shuffle and spill all of them at the merge point)
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
            for (int mi = 0; mi < RH / 16; ++mi) {
                const int am = rc * (RH / 16) + mi;
                f32x4 o;
                if constexpr (GEGLU) {                       // value block 2 ni, gate block 2 ni + 1 (packing.py interleave)
                    const f32x4 val = acc[2 * ni][am], gate = acc[2 * ni + 1][am];
                    if (kp.dbg & 4) { *(f32x4*)(E + (mi * 16 + frow) * ELD + ni * 16 + 4 * fq) = val * gate; continue; }
                    const f32x2 g01 = pt_gelu_erf2((f32x2){gate[0], gate[1]}), g23 = pt_gelu_erf2((f32x2){gate[2], gate[3]});
                    o = (f32x4){val[0] * g01[0], val[1] * g01[1], val[2] * g23[0], val[3] * g23[1]};
                } else if (p.act == 2) {                     // SiLU (condition encoder, controlnet_sdv.py:101-106)
                    const f32x4 a = acc[ni][am];
                    o = (f32x4){pt_silu(a[0]), pt_silu(a[1]), pt_silu(a[2]), pt_silu(a[3])};
                } else {
                    o = acc[ni][am];
                }
                *(f32x4*)(E + (mi * 16 + frow) * ELD + ni * 16 + 4 * fq) = o;
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's LDS writes have landed
        if (rc < 4) ig_stamp(kp, wave, lane, 5 + 2 * rc);
        const int mc0 = mrow0 + rc * RH;
        if (fastpath) {
            // per pass: finish the row, refill its side registers with the NEXT chunk's values, store.  Every side load is
            // thereby older than the store of its own pass and is consumed a whole chunk later: the counted wait in front
            // of its use never waits for a younger store, and only one finished row is held in registers at a time.
            if constexpr (NS == 0) {                         // nothing to order against: all rows read, then all stores
                f16x8 o8[NPASS], l8[WIDE ? NPASS : 1];
#pragma unroll
                for (int g = 0; g < NPASS; ++g) {
                    f32x4 ea, eb;
                    read_row(g, ea, eb);
                    o8[g] = finish_row(g, g, ea, eb, l8[WIDE ? g : 0]);
                }
#pragma unroll
                for (int g = 0; g < NPASS; ++g) store_row(rc, g, o8[g], l8[WIDE ? g : 0]);
            } else {
#pragma unroll
                for (int g = 0; g < NPASS; ++g) {
                    f32x4 ea, eb;
                    read_row(g, ea, eb);
                    f16x8 l8 = {};
                    const f16x8 o = finish_row(g, g, ea, eb, l8);
                    if (rc + 1 < NCHUNK) load_side(rc + 1, g);
                    store_row(rc, g, o, l8);
                }
            }
        } else {                                             // ragged / unaligned outputs, or more than two side inputs
            constexpr int RPP = 64 / LPR;                    // row-aligned lanes: rows per pass (lanes >= RPP * LPR idle)
            const int lrow = lane / LPR, col0 = wcol0 + (lane - lrow * LPR) * 8;
            const f16* res = (const f16*)p.res;
            const f16* res_lo = (const f16*)p.res_lo;
            const f16* vec = (const f16*)p.vec;
            const f16* blend = (const f16*)p.blend;
            for (int r = lrow; r < RH && lrow < RPP && col0 < Nout; r += RPP) {
                const int m = mc0 + r;
                if (m >= p.M) break;
                const float* e = E + r * ELD + (col0 - wcol0);
                for (int j = 0; j < 8 && col0 + j < Nout; ++j) {
                    float x = e[j];
                    float rs = res ? (float)res[(size_t)m * p.ldr + col0 + j] : 0.f;
                    if (res_lo) rs += (float)res_lo[(size_t)m * p.ldr + col0 + j];
                    if (res && !res_post) x += rs;
                    if (vec) x += (float)vec[(size_t)vec_index(p, m) * p.ldv + col0 + j];
                    if (blend) x = alpha * (float)blend[(size_t)m * p.ldb + col0 + j] + (1.0f - alpha) * x;
                    x *= col_scale(col0);
                    if (res && res_post) x += rs;
                    if (p.out_f32) ((float*)p.out)[(size_t)m * p.ldo + col0 + j] = x;
                    else {
                        const f16 o = (f16)x;
                        out[(size_t)m * p.ldo + col0 + j] = o;
                        if (out_lo) out_lo[(size_t)m * p.ldo + col0 + j] = (f16)(x - (float)o);
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // reads done before the next chunk overwrites E
        if (rc < 4) ig_stamp(kp, wave, lane, 6 + 2 * rc);
    }
}

// Shared epilogue: the wave's full width through LDS in row chunks (GEGLU / SiLU applied on the way in; the bias is
// already in the accumulators) and the fused row-wise tail.  acc[ni][mi] is the 16 x 16 block at rows wr*TM*16 + mi*16, columns wc*TN*16 + ni*16 of
// the tile, lane (frow, fq) holding channels 4 fq .. 4 fq + 3 of pixel frow.
// Tail variants.  The pipelined kernels are instantiated once per variant (tail_variant() picks it on the host), so each
// main loop is register-allocated next to ONE tail: with all eleven tails inlined behind a run-time switch the 256 x 320
// kernel carried 39 VGPR / 121 SGPR spills (the im2col origins reloaded from scratch inside the K loop behind the
// LDS-DMA queue, and reloads between the tail's stores, each a vmcnt(0) wait).  V_RT = run-time switch (plain-loop kernels).
enum { V_RT = -1, V_P0 = 0, V_P1, V_P2, V_EW, V_W0, V_W1, V_W2, V_G0, V_GEW, V_SPLITK, V_COUNT };

int tail_variant(const pt_igemm_params& p) {
    const int nside = (p.res ? 1 : 0) + (p.res_lo ? 1 : 0) + (p.vec ? 1 : 0) + (p.blend ? 1 : 0);
    if (p.act == 1) return (nside == 0 && !p.out_lo) ? V_G0 : V_GEW;
    if (nside > 2) return V_EW;
    return (p.out_lo ? V_W0 : V_P0) + nside;
}

__device__ __forceinline__ int tail_variant_dev(const pt_igemm_params& p) {
    const int nside = (p.res ? 1 : 0) + (p.res_lo ? 1 : 0) + (p.vec ? 1 : 0) + (p.blend ? 1 : 0);
    if (p.act == 1) return (nside == 0 && !p.out_lo) ? V_G0 : V_GEW;
    if (nside > 2) return V_EW;
    return (p.out_lo ? V_W0 : V_P0) + nside;
}

template <class CF, int VAR>
__device__ __forceinline__ void igemm_epilogue(const KParams& kp, f32x4 (&acc)[CF::TN][CF::TM], char* smem,
                                               int m0, int n0, int wave, int lane) {
    constexpr int TM = CF::TM, TN = CF::TN;
    const pt_igemm_params& p = kp.p;
    const int wr = wave / CF::WN, wc = wave % CF::WN;
    // ---------------- epilogue 2: the wave's full width, RH rows at a time, through LDS; row-wise fused tail.
    // every wave is done with the operand tiles.  Raw barrier: __syncthreads() would also drain the pipelined kernels'
    // past-the-end copies (still in flight towards the trash rows, carrying the side-input prefetch) with a vmcnt(0).
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ig_stamp(kp, wave, lane, 4);
    if (kp.dbg & 2) return;
    constexpr bool G = (VAR == V_G0 || VAR == V_GEW);
    const bool geglu = VAR == V_RT ? p.act == 1 : G;
    const int wcol0 = geglu ? (n0 + wc * TN * 16) / 2 : n0 + wc * TN * 16;
    const int mrow0 = m0 + wr * TM * 16;
    const int var = VAR == V_RT ? tail_variant_dev(p) : VAR;
    if constexpr (TN % 2 == 0 && (VAR == V_RT || G)) {
        if (var == V_G0)  { igemm_tail<CF, TN / 2, true, 0, false>(kp, acc, smem, mrow0, wcol0, p.N / 2, wave, lane); return; }
        if (var == V_GEW) { igemm_tail<CF, TN / 2, true, 3, false>(kp, acc, smem, mrow0, wcol0, p.N / 2, wave, lane); return; }
    }
#define PT_TAIL_CASE(V, NS_, WIDE_)                                                                                \
    if constexpr (VAR == V_RT || VAR == V)                                                                         \
        if (var == V) { igemm_tail<CF, TN, false, NS_, WIDE_>(kp, acc, smem, mrow0, wcol0, p.N, wave, lane); return; }
    PT_TAIL_CASE(V_P0, 0, false)
    PT_TAIL_CASE(V_P1, 1, false)
    PT_TAIL_CASE(V_P2, 2, false)
    PT_TAIL_CASE(V_W0, 0, true)
    PT_TAIL_CASE(V_W1, 1, true)
    PT_TAIL_CASE(V_W2, 2, true)
    PT_TAIL_CASE(V_EW, 3, false)
#undef PT_TAIL_CASE
}

}  // namespace
