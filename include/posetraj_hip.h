/*
 * posetraj_hip.h - C ABI of libposetraj_hip.so (gfx950 / MI355X only).
 *
 * Drop-in boundary of the PoseTraj denoising hot path.  The reference (robingg1/PoseTraj) has no FFI: its path is
 * a stack of torch.nn.Module forwards that dispatch to vendor libraries (cuDNN conv, cuBLAS GEMM, SDPA, native
 * GroupNorm/LayerNorm).  Each entry point below replaces one family of those implicit dispatches; the reference
 * call site it replaces is cited next to it (paths relative to /root/reference).  The Python classes in
 * posetraj_amd/ (same names / signatures as the reference's) call these through ctypes; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless said otherwise; activations are fp16, channels-last:
 *    [N = batch*frames, H, W, C] ("NHWC"), i.e. a token matrix [N*H*W, C] for the linear layers.
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no sync, no allocation),
 *    so a sequence of calls can be captured into a hipGraph.
 *  - return value: 0 on success, non-zero on error; pt_last_error() returns a message (thread local).
 *  - weights are PRE-PACKED by the host (posetraj_amd/packing.py): [Npad, Kpad] fp16, row n = output channel,
 *    K ordered (ky, kx, ci), Npad % 128 == 0, Kpad % 64 == 0, zero padded.
 */
#ifndef POSETRAJ_HIP_H
#define POSETRAJ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 8

int         pt_abi_version(void);
const char* pt_last_error(void);
/* remembers a zero-filled 256-byte device buffer as the CURRENT device's "zero page" (padded conv taps and ragged
 * attention tiles read from it).  Per-device state: call it once on every device the process uses.  The library keeps
 * no other device state except the per-device LDS opt-in of its kernels and the (process-wide, single-threaded)
 * profiling hooks at the end of this header. */
int pt_set_zero_page(const void* dev_zeros_256B);

/* ---------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer on MFMA (v_mfma_f32_16x16x32_f16), fused epilogue.
 *   t[m, n]   = sum_k A[m, k] * W[n, k] + bias[n]          m = (img, oy, ox), k = (ky, kx, ci)
 *   act == 1  : GEGLU - W rows are interleaved in blocks of 16 (value block, gate block); N is the packed
 *               width (2 * outputs); t'[m, j] = t_value * gelu_erf(t_gate), output width N / 2
 *   act == 2  : t = silu(t)
 *   t        += res[m, :] (+ res_lo[m, :])  (optional, unless res_post)  + vec[vidx(m), :]  (optional)
 *   t         = alpha * blend[m, :] + (1 - alpha) * t      (optional, AlphaBlender)
 *   out[m, :] = fp16(out_scale * t)        | fp16(res[m, :] + out_scale * t)  when res_post
 * A is gathered on the fly from up to two channels-last sources (x0: channels [0,C0), x1: [C0, C0+C1)) - the
 * skip concatenation of the up blocks costs no copy - with zero padding, optional stride 2 and optional nearest
 * 2x upsampling of the source.  A plain linear layer is the case KH = KW = 1, Nimg = M, Hin = Win = Hout = Wout = 1
 * (output extents must stay below 32000: pixel coordinates travel as 16-bit values inside the kernel; a short, very wide
 * image - Hout * stride + KH < 60 - may have up to 8 M columns: the temporal (3,1,1) convolutions of the VAE decoder see
 * the image (F, H*W)).  Hout / Wout are taken as given: rows / columns of taps beyond the input read zeros, so
 * Downsample2D(padding=0)'s F.pad(0,1,0,1) + stride-2 convolution is pad_h = pad_w = 0 with
 * Hout = (Hin + 1 - 3) / 2 + 1.
 * vec_mode: 0 none; 1 vidx = m / vG (per frame / per clip row vectors); 2 the batch-interleaved index of the
 *   temporal cross-attention context (models/modified_svd.py:152-159): vidx = ((m / vFS) * vS + m % vS) % vB.
 * Replaces: nn.Conv2d / nn.Conv3d((3,1,1)) / nn.Linear dispatches of diffusers' ResnetBlock2D,
 *   TemporalResnetBlock, Downsample2D, Upsample2D, Attention.to_{q,k,v,out}, FeedForward/GEGLU, proj_in/out
 *   (constructed at models/controlnet_sdv.py:352-391, models/unet_spatio_temporal_condition_controlnet.py:169-245),
 *   the zero-convs + conditioning_scale (models/controlnet_sdv.py:630-643), the condition encoder convs
 *   (models/controlnet_sdv.py:101-108) and cc_projection (models/controlnet_sdv_cam_infer.py:118).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct pt_igemm_params {
    const void* x0;  const void* x1;      /* fp16 sources; x1 may be NULL                               */
    int32_t C0, C1;                       /* channels taken from each source (C1 = 0 if x1 == NULL)     */
    int32_t ld0, ld1;                     /* elements between consecutive pixels of each source         */
    int32_t Nimg, Hin, Win, Hout, Wout;
    int32_t KH, KW, stride, pad_h, pad_w, upsample2x;
    int32_t M, N, K, Kpad;                /* M = Nimg*Hout*Wout; N = real (packed) output channels      */
    const void* w;                        /* fp16 [Npad, Kpad]                                          */
    const void* bias;                     /* fp16 [Npad] or NULL (interleaved like W when act == 1)     */
    void*       out;  int32_t ldo;
    const void* res;  int32_t ldr;
    const void* vec;  int32_t ldv;  int32_t vec_mode, vG, vFS, vS, vB;
    const void* blend; int32_t ldb; float alpha;
    float       out_scale;
    int32_t     act;
    int32_t     res_post;                 /* 1: out = res + out_scale * t  (t without the residual): accumulates
                                           * multiplicity x conditioning_scale x zero-conv INTO a U-Net skip
                                           * (out may alias res) - unet...:451-459,469 + controlnet_sdv.py:630-643 */
    int32_t     out_f32;                  /* 1: `out` is fp32 [M, ldo] (narrow outputs only: conv_out, N = 4)        */
    int32_t     cs_cols;  float cs_scale; /* output columns [0, cs_cols) are additionally multiplied by cs_scale
                                           * (cs_cols % 8 == 0): the Q third of a fused QKV projection leaves
                                           * pre-multiplied by softmax scale * log2(e) for pt_attn_spatial_f16     */
    void*       splitk_ws;                /* optional fp32 workspace of >= pt_igemm_splitk_ws_bytes(p) bytes: lets   */
    int64_t     splitk_ws_bytes;          /* small-M problems run split-K (two launches: slabs, ordered reduce)      */
    const void* res_lo;                   /* optional low half of `res` (same pitch ldr): the residual is res + res_lo */
    void*       out_lo;                   /* optional: also write fp16(v - fp16(v)) here (same pitch ldo) - the output
                                           * as an fp16 PAIR.  For the tensors of the residual stream (resblock /
                                           * transformer outputs, shortcut): GEMM operands and norms read `out` alone,
                                           * the epilogue that adds the tensor back as a residual reads the pair.    */
} pt_igemm_params;

int pt_igemm_f16(const pt_igemm_params* p, void* stream);
/* bytes of fp32 workspace with which pt_igemm_f16 would run this problem split-K (0: it would not).  No allocation
 * happens inside the library: the caller owns the workspace and passes it in splitk_ws / splitk_ws_bytes. */
int64_t pt_igemm_splitk_ws_bytes(const pt_igemm_params* p);
/* test / tuning hook: force the tile configuration (0 = 256x256, 1 = 128x320 (no GEGLU), 2 = 128x128,
 * 3 = 256x320 (channel-aligned layers only; others fall back), 4 = 128x160, 5 = 256x32 (N <= 32: the condition encoder), -1 = automatic) */
int pt_igemm_force_config(int32_t cfg);
/* tuning hook: device buffer of `capacity` uint64 that the next launches of the 256x256 / 256x320 kernels fill with
 * s_memtime stamps, 16 slots per wave ((workgroup * 8 + wave) * 16 + {0: start, 1: first K tile landed, 2: main loop
 * done, 3: stores issued, 4: epilogue barrier passed, 5 + 2c / 6 + 2c: epilogue chunk c staged in LDS / finished});
 * NULL switches the stamps off (the default). */
int pt_igemm_set_stamps(void* buf, int64_t capacity);

/* ---------------------------------------------------------------------------------------------------------
 * Fused GEGLU feed-forward (round 5):  out = tail( W2 . (value * gelu_erf(gate)) + b2 ),  [value | gate] = W1 . x + b1.
 * Replaces diffusers' FeedForward(activation_fn="geglu") = GEGLU(dim, 4 dim) -> Dropout(0) -> Linear(4 dim, dim) together with
 * the residual add that follows it - BasicTransformerBlock.ff (called from forward_TransformerSpatioTemporalModel,
 * /root/reference/models/modified_svd.py:193-196) and TemporalBasicTransformerBlock.ff_in / .ff (:73-76 and :97-103 with their
 * `+ residual` lines :78, :104) - where a workgroup can own whole rows: C == 320, the first level of the SVD U-Net and
 * ControlNet.  One launch instead of two pt_igemm_f16 calls; the [M, inner] intermediate never reaches memory.  Same packed
 * weights as those calls (w1 / b1 GEGLU-interleaved in 16-row blocks, w2 plain), same roundings (h to fp16 before the second
 * product), same fp32 accumulation order: results are bit-identical to the two-launch form.  Side inputs as in
 * pt_igemm_params (at most two of res / vec / blend).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct pt_ffn_params {
    const void* x;   int32_t ldx;         /* fp16 [M, ldx]: the normalised input (LayerNorm output)        */
    int32_t M, C, inner;                  /* C == 320; inner % 64 == 0 (1280)                               */
    const void* w1;  const void* b1;      /* fp16 [2 * inner, kpad1] / [2 * inner] (b1 may be NULL)         */
    int32_t kpad1;                        /* row pitch of w1 (== C)                                         */
    const void* w2;  const void* b2;      /* fp16 [Npad, kpad2] / [Npad] (b2 may be NULL)                   */
    int32_t kpad2;                        /* row pitch of w2 (>= inner)                                     */
    void*       out; int32_t ldo;
    const void* res; int32_t ldr;
    const void* vec; int32_t ldv; int32_t vec_mode, vG, vFS, vS, vB;
    const void* blend; int32_t ldb; float alpha;
    /* optional (pre_w != NULL): start one step earlier in the transformer block.  `x` then holds the ATTENTION OUTPUT rows and the
     * kernel's prologue computes  h = x . pre_w^T + pre_b + pre_res + pre_vec[idx(m)]  (the attention's output projection with its
     * residual and the collapsed cross-attention row vector, indexed like pt_igemm_params.vec),  LayerNorm(h; ln_gamma, ln_beta,
     * ln_eps)  as the feed-forward's input, and uses h itself as the feed-forward's residual (`res` must be NULL):
     *     out = tail( W2 . geglu(W1 . LN(h) + b1) + b2 + h )          - attn1.to_out + residual, norm3, ff + residual of
     * BasicTransformerBlock / TemporalBasicTransformerBlock (modified_svd.py:79-82,97-104) in one launch.  h is kept in fp32. */
    const void* pre_w; const void* pre_b; int32_t pre_kpad;
    const void* pre_res; int32_t pre_ldr;
    const void* pre_vec; int32_t pre_ldv; int32_t pre_vec_mode, pre_vG, pre_vFS, pre_vS, pre_vB;
    const void* ln_gamma; const void* ln_beta; float ln_eps;
} pt_ffn_params;
int pt_ffn_geglu_f16(const pt_ffn_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * LayerNorm + bias-free linear layer at K = 320 in ONE launch (round 6):  out = LN(x; gamma, beta, eps) . W^T, columns < cs_cols
 * multiplied by cs_scale before the rounding to fp16 (the attention's pre-scaled Q, as pt_igemm_params.cs_cols / cs_scale).
 * Replaces norm1 (nn.LayerNorm) + attn1.to_q / to_k / to_v (stacked [3C, C]) of BasicTransformerBlock and
 * TemporalBasicTransformerBlock at the 320-channel level (models/modified_svd.py:79-81; diffusers BasicTransformerBlock.forward):
 * a workgroup owns 128 whole rows, the normalised rows never reach memory.  w: the plain pack of pt_igemm_f16 ([Npad, kpad], Npad =
 * N rounded up to 128, kpad == K == 320).  The result is pt_layernorm_f16 + pt_igemm_f16 up to the summation order of the row
 * statistics (an fp16 ulp on isolated values of LN(x)).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct pt_lnlin_params {
    const void* x;   int32_t ldx;         /* fp16 [M, ldx]                                                  */
    int32_t M, N, K;                      /* K == 320; N % 8 == 0                                           */
    const void* w;   int32_t kpad;        /* fp16 [Npad, kpad]                                              */
    const void* bias;                     /* must be NULL (the projections this serves are bias-free)       */
    const void* ln_gamma; const void* ln_beta; float ln_eps;
    void*       out; int32_t ldo;
    int32_t cs_cols; float cs_scale;      /* cs_cols % 64 == 0                                              */
} pt_lnlin_params;
int pt_ln_linear_f16(const pt_lnlin_params* p, void* stream);
/* tuning hook (like pt_igemm_force_config): ablation instances of the kernel for tools/lnlin_bench.py - 1 no output stores, 2 no weight
 * copies behind the prologue's, 3 both, 4 no MFMAs, 8 return behind the prologue; results are wrong unless 0. */
int pt_ln_linear_set_ablation(int32_t bits);

/* ---------------------------------------------------------------------------------------------------------
 * fp32 path of the VAE encoder (round 5): `force_upcast`.  The reference runs its fp16 VAE in fp32 around encode()
 * (/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:453-462: vae.to(torch.float32) ... _encode_vae_image
 * (:174-195) ... vae.to(torch.float16)); these entry points replace nn.Conv2d / nn.Linear / nn.GroupNorm (+ SiLU) / the
 * mid block's softmax(Q K^T / sqrt(d)) V of diffusers' Encoder with fp32 operands end to end (fp32-input MFMA: exact fmaf chains).
 *   pt_conv2d_f32: channels-last fp32 x [Nimg, Hin, Win, ldx], w fp32 [Co, ldw] with K ordered (ky, kx, ci), taps outside the
 *   input read zeros (Hout / Wout explicit: Downsample2D(padding=0) = F.pad (0,1,0,1) + stride 2 is pad 0 with Hout = Hin / 2);
 *   out[m, co] = scale * (sum + bias[co]) + res[m, co].  A linear layer / plain A . B^T product is Hin = Win = KH = KW = 1.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct pt_conv_f32_params {
    const void* x; const void* w; const void* bias; const void* res; void* out;
    int32_t Nimg, Hin, Win, Hout, Wout, Ci, Co, KH, KW, stride, pad_h, pad_w;
    int32_t ldx, ldw, ldo, ldr;
    float   scale;
} pt_conv_f32_params;
int pt_conv2d_f32(const pt_conv_f32_params* p, void* stream);
/* GroupNorm over channels-last fp32 [n_samples, rows_per_sample, C] with fp64 statistics (stats_scratch: 2 * n_samples * groups
 * doubles), optional SiLU */
int pt_groupnorm_f32(const float* x, int64_t rows_per_sample, int32_t n_samples, int32_t C, int32_t groups, float eps,
                     const float* gamma, const float* beta, int32_t silu, double* stats_scratch, float* y, void* stream);
/* in place: scores[r, :n] = softmax(scale * scores[r, :n]) */
int pt_softmax_rows_f32(float* scores, int64_t rows, int32_t n, int64_t ld, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * GroupNorm (32 groups in the reference; any G dividing C here) over channels-last data.
 * A "sample" is rows_per_sample consecutive pixels: H*W for the 4-D norms, F*H*W for the norms of
 * TemporalResnetBlock, whose statistics run over (C/G, F, H, W).
 *   pt_groupnorm_stats : per-block partial sums of every group into `partials` (a scratch of
 *                        pt_groupnorm_scratch_floats() floats; deterministic, no atomics).
 *   pt_groupnorm_apply : folds the partials of its sample in a fixed order (every block the same bits), then
 *                        y = (x - mean) * rstd * gamma + beta, optionally SiLU; two sources are written out concatenated.
 *                        Same (C0, C1, groups, rows_per_sample, n_samples) as the stats call that filled `partials`.
 * Replaces nn.GroupNorm (+ SiLU) in ResnetBlock2D / TemporalResnetBlock / TransformerSpatioTemporalModel.norm and
 * conv_norm_out + conv_act (models/unet_spatio_temporal_condition_controlnet.py:237-238,494-495).
 * --------------------------------------------------------------------------------------------------------- */
int64_t pt_groupnorm_scratch_floats(int64_t rows_total, int32_t C, int32_t n_samples);
int pt_groupnorm_stats(const void* x0, const void* x1, int32_t C0, int32_t C1, int32_t groups,
                       int64_t rows_per_sample, int32_t n_samples, float* partials, void* stream);
int pt_groupnorm_apply(const void* x0, const void* x1, int32_t C0, int32_t C1, int32_t groups, int64_t rows_per_sample,
                       int32_t n_samples, float eps, const void* gamma, const void* beta, const float* partials,
                       int32_t silu, void* y, void* stream);

/* LayerNorm over the last dim of [M, C] fp16, eps inside sqrt, affine; optional row vector added BEFORE the
 * statistics (x + vec[vidx(m)], same vec_mode rules as the GEMM) - the frame-index embedding of
 * models/modified_svd.py:198-199.  Replaces nn.LayerNorm in Basic/TemporalBasicTransformerBlock. */
int pt_layernorm_f16(const void* x, int64_t M, int32_t C, const void* vec, int32_t ldv, int32_t vec_mode,
                     int32_t vG, const void* gamma, const void* beta, float eps, void* y, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Spatial self-attention, flash style: softmax(Q K^T * scale) V per (image, head), head_dim == 64.
 * qkv is the fused projection output [Nimg*S, ld] with Q at column h*64, K at k_off + h*64, V at v_off + h*64.
 * Replaces F.scaled_dot_product_attention in BasicTransformerBlock.attn1 (AttnProcessor2_0).
 * Temporal self-attention over the frame axis: tokens of one sequence are the F rows (b, f, s), f = 0..F-1, of
 * the same matrix, F <= 32 (two 16-frame blocks above 16: SVD-XT and the in-tree default num_frames = 25).  Replaces SDPA in TemporalBasicTransformerBlock.attn1 (models/modified_svd.py:79-81)
 * together with the two permute/reshape copies around it (:64-66, :110-112).
 * --------------------------------------------------------------------------------------------------------- */
/* q_prescaled = 1: the Q columns already carry scale * log2(e) (cs_cols / cs_scale of the projection's pt_igemm_f16
 * call - one rounding instead of a second one here, and one VALU op less per score in the kernel) */
int pt_attn_spatial_f16(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, void* out, int32_t ldo,
                        int32_t Nimg, int32_t S, int32_t heads, int32_t head_dim, float scale, int32_t q_prescaled,
                        void* stream);
int pt_attn_temporal_f16(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, void* out, int32_t ldo,
                         int32_t B, int32_t F, int32_t S, int32_t heads, int32_t head_dim, float scale, void* stream);
/* test / tuning hook (like pt_igemm_force_config): query blocks of 16 per wave of pt_attn_spatial_f16 - 3 (192 queries per
 * workgroup, the default) or 2 (128, the A/B form of round 5).  Process-wide; not seen by an already captured hipGraph. */
int pt_attn_spatial_set_nqb(int32_t nqb);

/* General attention, flash style, for the head sizes beside the U-Net's 64: softmax(Q K^T * scale) V per (batch, head) with
 * head_dim in {64, 80, 128, 512}; self- or cross-attention.  q [nbatch*Sq, ldq] (head h at column h*head_dim), k / v
 * [nbatch*Sk, ldk / ldv], out [nbatch*Sq, ldo]; q / k / v may be the three column blocks of one fused projection.
 * Replaces F.scaled_dot_product_attention in diffusers' Attention of the VAE mid blocks (one head of 512 channels over the
 * h*w tokens of a frame: pipeline/pipeline_stable_video_diffusion_controlnet.py:124,182,243 -> vae.encode / vae.decode), in
 * transformers' CLIPAttention of the image encoder (head_dim 80, 257 tokens: pipeline...:125,157) and in
 * BasicTransformerBlock.attn1 at head_dim 128 (the in-tree default num_attention_heads = (5,10,10,20),
 * models/controlnet_sdv.py:262). */
int pt_attn_f16(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out, int32_t ldo,
                int32_t nbatch, int32_t Sq, int32_t Sk, int32_t heads, int32_t head_dim, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * VAE-side pieces: the tail of AutoencoderKLTemporalDecoder.decode, decode_latents' layout,
 * tensor2vid's post-processing (pipeline/pipeline_stable_video_diffusion_controlnet.py:70-83,225-251), the posterior.
 * --------------------------------------------------------------------------------------------------------- */
/* time_conv_out = Conv3d(3, 3, (3,1,1), padding (1,0,0)) over the F frames of one vae.decode call, fused with the
 * channels-last -> NCHW transposition: x fp32 channels-last [F, HW, ldx] (first 3 columns), out fp32 [F, 3, HW].
 * w_host [3][3][3] = weight[co][ci][kt] and b_host [3] are HOST pointers (30 floats, passed to the kernel by value). */
int pt_vae_time_conv_out(const float* x, int32_t ldx, const float* w_host, const float* b_host, int32_t F, int64_t HW,
                         float* out, void* stream);
/* tensor2vid + VaeImageProcessor.postprocess of one clip: src fp32 [F, 3, HW]; u = clamp(x / 2 + 0.5, 0, 1);
 * mode 0 "pt": dst fp32 [F, 3, HW] = u; 1 "np": dst fp32 [F, HW, 3] = u; 2 "pil": dst uint8 [F, HW, 3] = rint(255 u) */
int pt_frames_postprocess(const float* src, int32_t F, int64_t HW, int32_t mode, void* dst, void* stream);
/* fp32 channels-last [N, HW, ld] -> fp32 [N, C, HW] (the encoder's moments / narrow fp32 outputs) */
int pt_nhwc_to_nchw_f32(const float* src, int32_t N, int32_t C, int64_t HW, int32_t ld, float* dst, void* stream);
/* DiagonalGaussianDistribution.sample: out = mean + exp(0.5 * clamp(logvar, -30, 20)) * noise with
 * params fp32 [N, 2C, HW] = (mean | logvar), noise / out fp32 [N, C, HW] */
int pt_gaussian_sample(const float* params, const float* noise, int32_t N, int32_t C, int64_t HW, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * CLIP vision tower pieces (transformers.CLIPVisionModelWithProjection, pipeline...:22,125,157); its linear
 * layers are pt_igemm_f16, its norms pt_layernorm_f16, its attention pt_attn_f16 (head_dim 80).
 * --------------------------------------------------------------------------------------------------------- */
/* patch extraction for the patch-embedding convolution (Conv2d(C, width, kernel = stride = P, bias=False) == a GEMM over
 * flattened patches): img [B, C, H, W] fp32 or fp16 -> fp16 [B * (H/P) * (W/P), ld], row = the patch in (c, ky, kx) order,
 * zero padded to ld >= C*P*P */
int pt_patchify_f16(const void* img, int32_t img_is_f32, int32_t B, int32_t C, int32_t H, int32_t W, int32_t P, int32_t ld,
                    void* out, void* stream);
/* y = act(x) on fp16: kind 1 erf-GELU ("gelu"), 2 x * sigmoid(1.702 x) ("quick_gelu") - CLIPMLP.activation_fn */
int pt_act_f16(const void* x, void* y, int64_t n, int32_t kind, void* stream);

/* Trajectory-map rasteriser: the maps scripts/run_inference_vipseg_json_repro.py:435-447 (flip_mode 0) and
 * utils/dataset.py:755-764 (flip_mode 1: cvtColor inside the per-track loop) draw with cv2.line(thickness 3, BGR (0,0,255)) +
 * cv2.circle(radius 3, filled, (0,255,0)), written as the [-1, 1] tensor [n_total, 3, H, W] (fp16 or fp32) that
 * image_processor.preprocess would produce; maps >= n_maps are black (-1).  pts: int32 [n_tracks][n_points][2] scaled (x, y);
 * map m draws segment point (start + m) -> (start + m + 1) and the disc at its end, tracks in order. */
int pt_rasterize_tracks(const int32_t* pts, int32_t n_tracks, int32_t n_points, int32_t start, int32_t n_maps, int32_t n_total,
                        int32_t H, int32_t W, int32_t flip_mode, int32_t out_is_f32, void* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Training objective (scripts/train_svd_traj_VIPSeg_14.py:1282-1407); its reverse pass: the "Training step" section below.
 * --------------------------------------------------------------------------------------------------------- */
/* network input of a training step (:1288-1345): noisy = latents + noise * sigma[b] (fp32, kept for the loss);
 * out[b, f, y, x, 0:4] = noisy / sqrt(sigma^2 + 1), out[..., 4:8] = (latents[b, 0] + noise[b, 0] * aug) * cond_scale[b] with
 * cond_scale[b] = image_mask[b] / vae.scaling_factor (conditioning dropout + the un-scaling of :1291), fp16 channels-last.
 * latents / noise / noisy fp32 [B, F, 4, HW]; sigma / cond_scale fp32 [B] on the device. */
int pt_edm_train_input(const float* latents, const float* noise, const float* sigma, const float* cond_scale, float aug,
                       int32_t B, int32_t F, int64_t HW, float* noisy, void* out, void* stream);
/* the sigma-weighted MSE (:1372-1384): loss[b] = mean_n w (pred c_out + c_skip noisy - target)^2, c_out = -s / sqrt(s^2 + 1),
 * c_skip = 1 / (s^2 + 1), w = (1 + s^2) / s^2.  pred channels-last fp16 / fp32 [B, F, HW, ldp] (first 4 channels), noisy /
 * target fp32 [B, F, 4, HW]; deterministic reduction, one workgroup per sample. */
int pt_edm_loss(const void* pred, int32_t pred_is_f32, int32_t ldp, const float* noisy, const float* target, const float* sigma,
                int32_t B, int32_t F, int64_t HW, float* loss, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Small element-wise pieces of the path.
 * --------------------------------------------------------------------------------------------------------- */
/* out = a + m * r   (ControlNet residual add with its multiplicity, unet...:451-459,469) */
int pt_axpy_f16(const void* a, const void* r, float m, void* out, int64_t n, void* stream);
/* y = silu(x) */
int pt_silu_f16(const void* x, void* y, int64_t n, void* stream);
/* sinusoidal Timesteps(dim, flip_sin_to_cos=True, shift 0): out[i, :] = [cos(t_i f_k) | sin(t_i f_k)] as fp16 */
int pt_timestep_embedding(const float* t, int32_t n, int32_t dim, void* out, void* stream);
/* [N, C, H, W] (fp16 or fp32, src_is_f32) -> channels-last fp16 [N, H, W, Cpad] (zero fill c >= C), and back */
int pt_nchw_to_nhwc_f16(const void* src, int32_t src_is_f32, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t Cpad, void* dst, void* stream);
int pt_nhwc_to_nchw(const void* src, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ld, void* dst,
                    int32_t dst_is_f32, void* stream);
/* per-pixel concat of camera R|T onto the condition features (controlnet_sdv_cam_infer.py:109-116):
 * dst[p, 0:C] = feat[p, 0:C]; dst[p, C:C+12] = cam[img(p), 0:12]; dst[p, C+12:Cpad] = 0 */
int pt_concat_camera(const void* feat, int32_t C, const void* cam, int32_t n_img, int64_t pix_per_img,
                     int32_t Cpad, void* dst, void* stream);
/* Loop prologue (pipeline...:532-537 + scheduling...:264-288): builds the CFG-doubled, sigma-scaled, image-latent
 * concatenated model input in channels-last fp16:  out[c2, f, y, x, 0:4] = latents[c, f, :, y, x] / sqrt(sigma^2+1),
 * out[c2, f, y, x, 4:8] = image_latents[c2, :, y, x]  for c2 in {uncond(0), cond(1)} x clips.
 * latents: fp32 [Bc, F, 4, h, w]; image_latents: fp16 [2*Bc, 4, h, w] (one frame, repeated over F). */
int pt_scale_concat_input(const float* latents, const void* image_latents, float sigma, int32_t Bc, int32_t F,
                          int32_t h, int32_t w, void* out, void* stream);
/* Loop epilogue (pipeline...:567-572 + scheduling...:418-528, gamma == 0): classifier-free guidance with the
 * per-frame scale + Euler step, fp32 state:  pred = u + g_f (c - u) (rounded to fp16 like the reference's fp16
 * model output when noise_pred is fp16; kept in fp32 when np_is_f32);
 * x0 = pred*c_out + x*c_skip (v_prediction) | x - sigma*pred (epsilon) | pred (sample);
 * x += (x - x0)/sigma * (sigma_next - sigma).
 * noise_pred: fp16 or fp32 channels-last [2*Bc, F, h, w, ldn] (first 4 channels used); latents fp32 [Bc, F, 4, h, w]. */
int pt_cfg_euler_step(const void* noise_pred, int32_t np_is_f32, int32_t ldn, const float* guidance, float sigma,
                      float sigma_next, int32_t prediction_type, int32_t Bc, int32_t F, int32_t h, int32_t w,
                      float* latents, void* stream);

/* EulerDiscreteScheduler.scale_model_input (scheduling...:264-288): y = x * k with k = 1/sqrt(sigma^2+1) */
int pt_scale(const void* x, int32_t is_f32, float k, void* y, int64_t n, void* stream);
/* EulerDiscreteScheduler.step, gamma == 0 (scheduling...:418-528) on flat arrays: fp32 sample in, fp32 prev_sample out;
 * prediction_type 0 v_prediction, 1 epsilon, 2 sample */
int pt_euler_step(const void* model_output, int32_t mo_is_f32, const float* sample, float sigma, float sigma_next,
                  int32_t prediction_type, float* prev_sample, int64_t n, void* stream);

/* EulerDiscreteScheduler.add_noise (scheduling...:530-553): y[b, ...] = x[b, ...] + noise[b, ...] * sigma[b] in the
 * tensors' dtype (fp16 or fp32); sigma_per_sample: fp32 [n / per_sample] on the device */
int pt_add_noise(const void* x, const void* noise, int32_t is_f32, const float* sigma_per_sample, int64_t per_sample,
                 void* y, int64_t n, void* stream);

/* _resize_with_antialiasing (pipeline/pipeline_stable_video_diffusion_controlnet.py:604-712), the first stage of
 * _encode_image (:145-157): separable Gaussian blur with reflect padding (x then y; taps_x [kx], taps_y [ky] fp32 on
 * the device, built by the caller with the reference's formula) followed by bicubic interpolation with
 * align_corners=True (A = -0.75, border taps clamped).  src [planes, H, W] fp32 -> dst [planes, oh, ow] fp32;
 * tmp: 2 * planes * H * W floats of caller-owned scratch. */
int pt_resize_antialias_f32(const float* src, int32_t planes, int32_t H, int32_t W, int32_t oh, int32_t ow,
                            const float* taps_x, int32_t kx, const float* taps_y, int32_t ky, float* tmp, float* dst,
                            void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Training step (scripts/train_svd_traj_VIPSeg_14.py:1408-1425 `accelerator.backward(loss)` /
 * `optimizer.step()`): the reverse pass of the ControlNet through the frozen U-Net's up path.  Data gradients of
 * convolutions / linear layers are pt_igemm_f16 launches over a transposed (and tap-flipped) pack; everything else is below.
 * Activation gradients are fp16 (the caller scales the loss, like accelerate's fp16 GradScaler), parameter gradients are
 * ACCUMULATED into fp32 buffers (atomics; the caller zeroes them once per optimizer step).
 *
 * pt_gemm_f16: C[b] (+)= alpha * A[b] * B[b] with A(m, k) at A + m sa_m + k sa_k, B(k, n) at B + k sb_k + n sb_n (one
 * unit stride per operand), C(m, n) at C + m sc_m + n sc_n; batch b = (b0, b1, b2) with per-level element offsets
 * ba* / bb* / bc*.  out_mode 0: fp16 store, 1: fp32 store, 2: fp32 atomic add (required for splits > 1: split-K),
 * 3: fp32 C += (plain read-modify-write: one writer per element, splits == 1).
 * g_H > 0 turns on the convolution gather for B (weight gradients): k runs over the OUTPUT pixels (img, oy, ox) of a
 * [*, g_OH, g_OW] image, the innermost batch level b2 over the g_KH x g_KW taps, and row k of B is the input pixel
 * (oy g_stride + ky - g_pad_h, ox g_stride + kx - g_pad_w) of the channels-last [*, g_H, g_W, g_ld] source (zeros outside).
 * Replaces autograd's conv / linear weight gradients and the backward of F.scaled_dot_product_attention.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct pt_gemm_params {
    const void* A;  const void* B;  void* C;
    int32_t M, N;
    int64_t K;
    int64_t sa_m, sa_k, sb_k, sb_n, sc_m, sc_n;
    int32_t nb0, nb1, nb2, out_mode;
    int64_t ba0, ba1, ba2, bb0, bb1, bb2, bc0, bc1, bc2;
    float   alpha;
    int32_t splits;
    int32_t g_H, g_W, g_OH, g_OW, g_KH, g_KW, g_stride, g_pad_h, g_pad_w, g_reserved;
    int64_t g_ld;
} pt_gemm_params;
int pt_gemm_f16(const pt_gemm_params* p, void* stream);

/* spatial self-attention of the training step.  Forward = pt_attn_f16 (head_dim 64 / 128) that also writes
 * lse[(batch Sq + q) heads + head] = log2 sum_key exp2(score scale log2 e); backward = a flash backward in two passes
 * (query-stationary: dQ; key-stationary: dK, dV) that rebuilds P from lse tile by tile - no score matrix reaches HBM.
 * out: the forward's output (for Dq = sum_d dO O, written to dq_dot [nbatch S heads] fp32 scratch); dq / dk / dv: fp16 with
 * pitch ldd (the column blocks of one [rows, 3 C] gradient).  Replaces the backward of F.scaled_dot_product_attention. */
int pt_attn_fwd_lse_f16(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, void* out,
                        int32_t ldo, int32_t nbatch, int32_t Sq, int32_t Sk, int32_t heads, int32_t head_dim, float scale,
                        float* lse, void* stream);
int pt_attn_bwd_f16(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v, int32_t ldv, const void* out,
                    int32_t ldout, const void* dout, int32_t ldo, const float* lse, float* dq_dot, void* dq, void* dk, void* dv,
                    int32_t ldd, int32_t nbatch, int32_t S, int32_t heads, int32_t head_dim, float scale, void* stream);
/* backward of pt_attn_temporal_f16 (same addressing: token f of (clip b, position s) is row (b F + f) S + s of the fused
 * projection) for F <= 16 frames: dqkv gets dQ | dK | dV at the offsets 0 / k_off / v_off of its rows (pitch ldd). */
int pt_attn_temporal_bwd_f16(const void* qkv, int32_t ld, int32_t k_off, int32_t v_off, const void* dout, int32_t ldo, void* dqkv,
                             int32_t ldd, int32_t B, int32_t F, int32_t S, int32_t heads, int32_t head_dim, float scale, void* stream);
/* backward of pt_groupnorm_stats + pt_groupnorm_apply (same argument meaning; two channels-last sources): dy [rows, C0 + C1]
 * -> dx0 [rows, C0], dx1 [rows, C1]; dgamma / dbeta fp32 [C0 + C1] accumulated, or both NULL (frozen network).
 * stat: 4 * n_samples * groups floats of scratch. */
int pt_groupnorm_bwd(const void* x0, const void* x1, int32_t C0, int32_t C1, int32_t groups, int64_t rows_per_sample,
                     int32_t n_samples, float eps, const void* gamma, const void* beta, int32_t silu, const void* dy,
                     void* dx0, void* dx1, float* dgamma, float* dbeta, float* stat, void* stream);
/* backward of pt_layernorm_f16 (without the pre-add vector); rowstat: 2 M floats of scratch, needed with dgamma / dbeta */
int pt_layernorm_bwd(const void* x, int64_t M, int32_t C, const void* gamma, float eps, const void* dy, void* dx,
                     float* dgamma, float* dbeta, float* rowstat, void* stream);
/* out[seg, c] += sum of dy[row, c] over the rows_per_seg rows of segment seg: bias gradients (one segment) and the gradient
 * of row vectors that were broadcast over the rows of a frame / clip */
int pt_colsum_f16(const void* dy, int64_t rows_per_seg, int32_t nseg, int32_t C, int32_t ld, float* out, void* stream);
/* attention backward around pt_gemm_f16: P = softmax(S) per row (S fp32, already scaled; P fp16);
 * dS = P (dP - sum_j P_j dP_j) */
int pt_softmax_rows(const float* S, int64_t rows, int32_t n, int64_t ld, void* P, int64_t ldp, void* stream);
int pt_softmax_bwd_rows(const void* P, int64_t ldp, const float* dP, int64_t ld, int64_t rows, int32_t n, void* dS, int64_t lds,
                        void* stream);
/* GEGLU un-fused (the training path keeps the projection's output): h [M, 2 I] = [value | gate] -> y = value * gelu(gate);
 * backward dh = [dy gelu(gate) | dy value gelu'(gate)] */
int pt_geglu_f16(const void* h, int64_t M, int32_t I, void* y, void* stream);
int pt_geglu_bwd(const void* h, const void* dy, int64_t M, int32_t I, void* dh, void* stream);
int pt_silu_bwd(const void* x, const void* dy, int64_t n, void* dx, void* stream);
/* AlphaBlender with a clip-wide weight: out = alpha a + (1 - alpha) b; gradient of the weight: out += scale * sum dy (a - b)
 * (b may be NULL: sum dy a) */
int pt_lerp_f16(const void* a, const void* b, float alpha, int64_t n, void* out, void* stream);
int pt_dot_diff(const void* dy, const void* a, const void* b, int64_t n, float scale, float* out, void* stream);
/* The same with the weight in DEVICE memory (round 6: ControlNetTrainer(use_graph=True) replays the step as a hipGraph, and a captured
 * launch cannot carry a host float that changes every step): alpha_dev / k_dev point at one fp32.
 *   pt_lerp_f16_dev: out = alpha a + (1 - alpha) b;  pt_dot_diff_dev: out += alpha (1 - alpha) sum dy (a - b)  (d alpha / d mix_factor folded in);
 *   pt_scale_f16_dev: y = k x, or (1 - k) x with one_minus;  pt_sigmoid_gather_f32: out[i] = sigmoid(flat[idx[i]]) - all AlphaBlender
 *   weights of the trainable network from the fp32 master buffer in one launch. */
int pt_lerp_f16_dev(const void* a, const void* b, const float* alpha_dev, int64_t n, void* out, void* stream);
int pt_dot_diff_dev(const void* dy, const void* a, const void* b, int64_t n, const float* alpha_dev, float* out, void* stream);
int pt_scale_f16_dev(const void* x, const float* k_dev, int32_t one_minus, int64_t n, void* y, void* stream);
int pt_sigmoid_gather_f32(const float* flat, const int64_t* idx, int32_t n, float* out, void* stream);
/* y[r, :] = x[r, :] + vec[r / rows_per_vec, :] */
int pt_add_rowvec_f16(const void* x, const void* vec, int64_t rows, int32_t C, int64_t rows_per_vec, void* y, void* stream);
/* backward of the nearest 2x upsampling in front of Upsample2D's convolution: [N, 2H, 2W, C] -> [N, H, W, C] block sums */
int pt_sumpool2x_f16(const void* du, int32_t N, int32_t H, int32_t W, int32_t C, void* dx, void* stream);
/* zero-interleave of a stride-2 convolution's output gradient: z [N, H, W, C], z[2y, 2x] = dy[y, x] */
int pt_zero_insert2x_f16(const void* dy, int32_t N, int32_t OH, int32_t OW, int32_t H, int32_t W, int32_t C, void* z, void* stream);
/* gradient of pt_edm_loss's batch mean w.r.t. pred, times `scale`: fp16 channels-last [B, F, HW, 8], channels 4..7 zero */
int pt_edm_loss_bwd(const void* pred, int32_t pred_is_f32, int32_t ldp, const float* noisy, const float* target, const float* sigma,
                    int32_t B, int32_t F, int64_t HW, float scale, void* dpred, void* stream);
/* torch.optim.AdamW step (scripts/train_svd_traj_VIPSeg_14.py:1051,1070-1076,1423) over a flat fp32 parameter buffer;
 * g is multiplied by inv_scale first (loss un-scaling); step counts from 1 */
int pt_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int32_t step, float inv_scale, void* stream);
/* pt_adamw_f32 with the two passes the trainer runs behind it folded in (round 6): half_mirror (may be NULL) receives fp16(p) - the fp16
 * copy of the parameters the next forward reads norm weights and biases from - and zero_grad != 0 clears g.  n % 4 == 0, 16-byte aligned. */
int pt_adamw_fused_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int32_t step, float inv_scale, void* half_mirror, int32_t zero_grad, void* stream);
/* the fp16 operand pt_igemm_f16 streams, straight from the fp32 master weight w [T][Co][Ci] (tap-major: T = kh kw taps, 1 for a
 * linear layer - the layout the trainer keeps weights, gradients and Adam moments in, so that weight gradients are written
 * with unit stride):
 * transposed == 0: dst[co][t Cpad + ci] (the forward pack); 1: dst[ci][(T - 1 - t) Cpad + co] (the data gradient's operand:
 * channels swapped, taps flipped).  dst rows have Kpad halfs; its padding is never written (zero-fill the buffer once).
 * bias / dst_bias (forward pack only, both optional): fp16 copy of the bias.  Runs after every optimizer step. */
int pt_pack_weight_f32(const float* w, int32_t Co, int32_t Ci, int32_t T, int32_t transposed, const float* bias, void* dst,
                       int32_t Kpad, int32_t Cpad, void* dst_bias, void* stream);
/* linear layer over M <= 16 rows (time-embedding MLPs, time_emb_proj, the single-key cross-attentions' to_v / to_out, the frame
 * position embedding): out[m, n] = sum_k x[m, k] W[n, k] + bias[n] (+ res[m, n]); W: a pack [*, Kpad] */
int pt_gemv_f16(const void* x, int32_t ldx, int32_t M, const void* W, int32_t Kpad, int32_t K, int32_t N, const void* bias,
                const void* res, int32_t ldr, void* out, int32_t ldo, void* stream);
/* out[0] += sum g^2 in fp64 (gradient norm; non-finite when any gradient overflowed) */
int pt_sumsq_f32(const float* g, int64_t n, double* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Measurement hooks for bench.py: when enabled every pt_igemm_f16 / pt_attn_spatial_f16 launch is bracketed by
 * hipEvents on its stream.  pt_prof_collect() synchronises those events and accumulates per kernel family
 * (0 = igemm, 1 = attn_spatial, 2 = pt_gemm_f16): launches, milliseconds, algorithmic flops.
 * --------------------------------------------------------------------------------------------------------- */
int pt_prof_enable(int32_t on);
int pt_prof_collect(int32_t family, int64_t* launches, double* ms, double* flops);
/* per-launch form: writes up to cap (ms, flops) pairs in launch order, returns how many; clears the records */
int64_t pt_prof_collect_list(int32_t family, double* ms, double* flops, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* POSETRAJ_HIP_H */
