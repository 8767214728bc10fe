"""Host-side execution of the spatio-temporal blocks on the HIP kernels.

Each class mirrors one diffusers==0.24.0 block the reference instantiates (``models/controlnet_sdv.py:352-391``,
``models/unet_spatio_temporal_condition_controlnet.py:169-232``; forwards restated in-tree at
``models/modified_svd.py:50-348``) but executes it on ONE activation layout - channels-last fp16
``[N = B*F, H, W, C]`` - so every permute/reshape copy of the reference disappears:

* spatial ops see the token matrix ``[N*H*W, C]``;
* temporal convolutions see an image ``(H', W') = (F, H*W)`` with a (3 x 1) kernel over the same memory;
* temporal attention reads its F tokens with a row stride of ``H*W`` (``pt_attn_temporal_f16``);
* the skip concatenation of the up blocks is a 2-source gather inside the convolution.

Work that provably does not depend on the activations is hoisted out of the blocks:
``time_emb_proj(silu(emb))`` of all residual blocks is ONE stacked GEMM per forward, and cross-attention over the
single image-embedding token (softmax over one logit == 1) collapses to ``to_out(to_v(ctx))``, stacked likewise
and added as a per-clip row vector in the epilogue of the self-attention output projection - including the
batch-interleaved context index of the temporal blocks (``modified_svd.py:152-159``).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import ops
from .packing import pack_conv2d, pack_conv_t3, pack_linear, vec16

HEAD_DIMS = (64, 128)
# feed-forward intermediates larger than this are produced and consumed in row chunks (TransformerSpatioTemporalModel._feed_forward)
FF_CHUNK_BYTES = int(os.environ.get("PT_FF_CHUNK_MB", "0")) << 20


class RowStack:
    """Collects ``[C_i, K]`` weight slabs that all multiply the same small input, to run them as one GEMM."""

    def __init__(self):
        self.ws: List[torch.Tensor] = []
        self.bs: List[torch.Tensor] = []
        self.n = 0

    def add(self, w: torch.Tensor, b: torch.Tensor) -> int:
        off = self.n
        self.ws.append(w.detach().float())
        self.bs.append(b.detach().float())
        self.n += w.shape[0]
        return off

    def pack(self, device) -> Optional[ops.Packed]:
        if not self.ws:
            return None
        pw = pack_linear(torch.cat(self.ws, 0), torch.cat(self.bs, 0), device)
        self.ws, self.bs = [], []
        return pw


@dataclass
class Ctx:
    """Per-forward state shared by the blocks."""
    B: int                       # clips x CFG halves
    F: int                       # frames
    temb: torch.Tensor           # [B, sum C] : time_emb_proj(silu(emb)) of every residual block
    xattn: Optional[torch.Tensor]  # [B, sum C] : to_out(to_v(image embedding)) of every cross-attention
    cache: Dict = field(default_factory=dict)
    # a forward over ONE CFG half of the batch (pipeline: denoise(split_cfg=True)): the temporal cross-attention's context index
    # ((b * S + s) mod B_total, modified_svd.py:152-159) still addresses the table of BOTH halves
    half: Optional[int] = None                 # which slice of the full batch this forward covers (None: the whole batch)
    xattn_full: Optional[torch.Tensor] = None  # [B_total, sum C]
    xattn_full_swapped: Optional[torch.Tensor] = None   # rows rolled by one: the table a half with an odd row offset sees


class SpatioTemporalResBlock:
    """``SpatioTemporalResBlock`` = ResnetBlock2D -> TemporalResnetBlock -> AlphaBlender."""

    def __init__(self, sd, p, eps, device, temb_stack: Optional[RowStack], eps_t: Optional[float] = None, switch: bool = False):
        """``eps_t``: eps of the temporal block's norms (``temporal_eps``; default = ``eps``).  ``switch``:
        ``switch_spatial_to_temporal_mix`` of the VAE decoder's blocks (``merge_strategy="learned"``): the blend weight of the
        SPATIAL branch is 1 - sigmoid(mix_factor) instead of sigmoid(mix_factor).  Blocks whose state dict holds no
        ``time_emb_proj`` (``temb_channels=None``: the VAE) add no time-embedding row."""
        s, t = p + "spatial_res_block.", p + "temporal_res_block."
        self.eps, self.eps_t = eps, (eps if eps_t is None else eps_t)
        self.n1 = (vec16(sd[s + "norm1.weight"], device), vec16(sd[s + "norm1.bias"], device))
        self.conv1 = pack_conv2d(sd[s + "conv1.weight"], sd[s + "conv1.bias"], device)
        self.cout = self.conv1.N
        self.has_temb = s + "time_emb_proj.weight" in sd
        if self.has_temb:
            self.off_s = temb_stack.add(sd[s + "time_emb_proj.weight"], sd[s + "time_emb_proj.bias"])
        self.n2 = (vec16(sd[s + "norm2.weight"], device), vec16(sd[s + "norm2.bias"], device))
        self.conv2 = pack_conv2d(sd[s + "conv2.weight"], sd[s + "conv2.bias"], device)
        self.shortcut = None
        if s + "conv_shortcut.weight" in sd:
            self.shortcut = pack_conv2d(sd[s + "conv_shortcut.weight"], sd[s + "conv_shortcut.bias"], device, padding=0)
        self.tn1 = (vec16(sd[t + "norm1.weight"], device), vec16(sd[t + "norm1.bias"], device))
        self.tconv1 = pack_conv_t3(sd[t + "conv1.weight"], sd[t + "conv1.bias"], device)
        if self.has_temb:
            self.off_t = temb_stack.add(sd[t + "time_emb_proj.weight"], sd[t + "time_emb_proj.bias"])
        self.tn2 = (vec16(sd[t + "norm2.weight"], device), vec16(sd[t + "norm2.bias"], device))
        self.tconv2 = pack_conv_t3(sd[t + "conv2.weight"], sd[t + "conv2.bias"], device)
        sig = float(torch.sigmoid(sd[p + "time_mixer.mix_factor"].detach().float().cpu())[0])
        self.alpha = 1.0 - sig if switch else sig               # weight of the spatial branch

    def run(self, ctx: Ctx, x0: torch.Tensor, x1: Optional[torch.Tensor] = None) -> torch.Tensor:
        N, H, W, _ = x0.shape
        S, F, B, C = H * W, ctx.F, ctx.B, self.cout
        geom = (N, H, W)
        # -- spatial ResnetBlock2D
        y = ops.groupnorm(x0, *self.n1, rows_per_sample=S, n_samples=N, eps=self.eps, silu=True, x1=x1)
        tv = lambda off: dict(vec=ctx.temb[:, off:off + C], vec_mode=1, vG=F * S) if self.has_temb else {}
        h = ops.igemm(y.view(N, H, W, -1), self.conv1, geom=geom, **tv(self.off_s if self.has_temb else 0))
        y = ops.groupnorm(h.view(N, H, W, C), *self.n2, rows_per_sample=S, n_samples=N, eps=self.eps, silu=True)
        # residual stream: shortcut, spatial and block outputs are fp16 pairs (ops.WIDE_STREAM); they enter norms and
        # GEMMs as their high half and residual adds as the pair
        if self.shortcut is not None:
            sc = ops.igemm(x0, self.shortcut, x1=x1, geom=geom, wide="sc" in ops.WIDE_KINDS)
        else:
            sc = ops.wview(x0, N * S, C)
        xs = ops.igemm(y.view(N, H, W, C), self.conv2, geom=geom, res=sc, wide="xs" in ops.WIDE_KINDS)
        # -- TemporalResnetBlock on the image (F, H*W); GroupNorm statistics over (C/32, F, H, W)
        tgeom = (B, F, S)
        y = ops.groupnorm(xs, *self.tn1, rows_per_sample=F * S, n_samples=B, eps=self.eps_t, silu=True)
        h = ops.igemm(y.view(B, F, S, C), self.tconv1, geom=tgeom, **tv(self.off_t if self.has_temb else 0))
        y = ops.groupnorm(h, *self.tn2, rows_per_sample=F * S, n_samples=B, eps=self.eps_t, silu=True)
        # x_t = conv + bias + xs ; out = a*xs + (1-a)*x_t = xs + (1-a)*(conv + bias)   (AlphaBlender, image_only_indicator
        # == 0): residual and blend input are the same tensor, so ONE side input read after the scale does both
        out = ops.igemm(y.view(B, F, S, C), self.tconv2, geom=tgeom, res=xs, res_post=True, out_scale=1.0 - self.alpha,
                        wide="rb" in ops.WIDE_KINDS)
        return ops.wview(out, N, H, W, C)


class _TLayer:
    pass


class TransformerSpatioTemporalModel:
    """``TransformerSpatioTemporalModel`` (forward restated at ``modified_svd.py:118-223``)."""

    def __init__(self, sd, p, heads, device, xattn_stack: RowStack):
        self.heads = heads
        self.norm = (vec16(sd[p + "norm.weight"], device), vec16(sd[p + "norm.bias"], device))
        self.proj_in = pack_linear(sd[p + "proj_in.weight"], sd[p + "proj_in.bias"], device)
        self.C = self.proj_in.N
        # head_dim 64 (the SVD checkpoints: heads (5,10,20,20)) or 128 (level 2 of the reference's in-tree default
        # num_attention_heads = (5,10,10,20), models/controlnet_sdv.py:262)
        if self.C % heads or self.C // heads not in HEAD_DIMS:
            raise ValueError(f"{p}: {self.C} channels / {heads} heads = head_dim {self.C / heads}; posetraj_amd supports head_dim "
                             f"{HEAD_DIMS}")
        self.head_dim = self.C // heads
        self.layers: List[_TLayer] = []
        i = 0
        while f"{p}transformer_blocks.{i}.norm1.weight" in sd:
            a, b = f"{p}transformer_blocks.{i}.", f"{p}temporal_transformer_blocks.{i}."
            L = _TLayer()
            ln = lambda k: (vec16(sd[k + ".weight"], device), vec16(sd[k + ".bias"], device))
            lin = lambda k, **kw: pack_linear(sd[k + ".weight"], sd.get(k + ".bias"), device, **kw)
            qkv = lambda k: pack_linear(torch.cat([sd[k + "to_q.weight"], sd[k + "to_k.weight"], sd[k + "to_v.weight"]], 0),
                                        None, device)
            # single-token cross-attention: softmax over one logit is exactly 1 -> out = to_out(to_v(ctx))
            # (pre-multiplied on the HOST in fp32: pack time must not pull a vendor GEMM onto the device)
            xat = lambda k: xattn_stack.add(sd[k + "to_out.0.weight"].detach().float().cpu() @ sd[k + "to_v.weight"].detach().float().cpu(),
                                            sd[k + "to_out.0.bias"].detach().cpu())
            L.ln1, L.qkv, L.o = ln(a + "norm1"), qkv(a + "attn1."), lin(a + "attn1.to_out.0")
            L.x_off = xat(a + "attn2.")
            L.ln3, L.ff1, L.ff2 = ln(a + "norm3"), lin(a + "ff.net.0.proj", geglu=True), lin(a + "ff.net.2")
            L.ln_in, L.fi1, L.fi2 = ln(b + "norm_in"), lin(b + "ff_in.net.0.proj", geglu=True), lin(b + "ff_in.net.2")
            L.tln1, L.tqkv, L.to = ln(b + "norm1"), qkv(b + "attn1."), lin(b + "attn1.to_out.0")
            L.tx_off = xat(b + "attn2.")
            L.tln3, L.tf1, L.tf2 = ln(b + "norm3"), lin(b + "ff.net.0.proj", geglu=True), lin(b + "ff.net.2")
            self.layers.append(L)
            i += 1
        self.tpe1 = pack_linear(sd[p + "time_pos_embed.linear_1.weight"], sd[p + "time_pos_embed.linear_1.bias"], device)
        self.tpe2 = pack_linear(sd[p + "time_pos_embed.linear_2.weight"], sd[p + "time_pos_embed.linear_2.bias"], device)
        self.alpha = float(torch.sigmoid(sd[p + "time_mixer.mix_factor"].detach().float().cpu())[0])
        self.proj_out = pack_linear(sd[p + "proj_out.weight"], sd[p + "proj_out.bias"], device)
        self._femb: Dict = {}

    def frame_embedding(self, B: int, F: int, device) -> torch.Tensor:
        """time_pos_embed(Timesteps(C)(arange(F))) tiled over the batch -> [B*F, C]; weights-only, cached."""
        key = (B, F)
        if key not in self._femb:
            t = torch.arange(F, dtype=torch.float32, device=device)
            e = ops.igemm(ops.silu(ops.igemm(ops.timestep_embedding(t, self.C), self.tpe1)), self.tpe2)
            self._femb[key] = e.repeat(B, 1).contiguous()
        return self._femb[key]

    @staticmethod
    def _feed_forward(y, w1, w2, N, S, *, res, vec=None, blend=None, alpha=0.0):
        """GEGLU feed-forward ``w2(geglu(w1(y))) + res`` (+ per-frame row vector, AlphaBlender) in FF_CHUNKS row chunks of
        whole frames: the ``[rows, 4C]`` intermediate of one chunk (660 MB for all 28 frames at level 0) is written by the
        first GEMM and read back by the second while it still sits in the 256 MiB Infinity Cache instead of making a round
        trip through HBM.  Same kernels, same results bit for bit (rows are independent)."""
        M, C = res.shape
        if ops.ffn_fusable(w1, w2) and getattr(res, "lo", None) is None:   # C = 320: one launch, the [M, 4C] intermediate stays on the CU
            # (a wide-stream residual - no caller passes one today - takes the two launches below: pt_ffn_geglu_f16 has no wide tail)
            kw = dict(vec=vec, vec_mode=1, vG=S) if vec is not None else {}
            return ops.ffn_geglu(y, w1, w2, res=res, blend=blend, alpha=alpha, **kw)
        inter = M * w1.n_out * 2
        chunks = 1
        if FF_CHUNK_BYTES > 0:
            while chunks < N and inter // chunks > FF_CHUNK_BYTES and N % (chunks * 2) == 0 and (M // (chunks * 2)) % 256 == 0 \
                    and (M // (chunks * 2)) // 256 * ((w2.N + 319) // 320) >= 224:
                chunks *= 2
        if chunks == 1:
            g = ops.igemm(y, w1)
            kw = dict(vec=vec, vec_mode=1, vG=S) if vec is not None else {}
            return ops.igemm(g, w2, res=res, blend=blend, alpha=alpha, **kw)
        out = torch.empty((M, C), dtype=torch.float16, device=y.device)
        rows, fpc = M // chunks, N // chunks
        g = torch.empty((rows, w1.n_out), dtype=torch.float16, device=y.device)
        for c in range(chunks):
            sl = slice(c * rows, (c + 1) * rows)
            ops.igemm(y[sl], w1, out=g)
            kw = dict(vec=vec[c * fpc:(c + 1) * fpc], vec_mode=1, vG=S) if vec is not None else {}
            ops.igemm(g, w2, res=res[sl], blend=None if blend is None else blend[sl], alpha=alpha, out=out[sl], **kw)
        return out

    def run(self, ctx: Ctx, x: torch.Tensor) -> torch.Tensor:
        N, H, W, C = x.shape
        S, F, B, heads = H * W, ctx.F, ctx.B, self.heads
        xt = ops.wview(x, N * S, C)
        h = ops.igemm(ops.groupnorm(x, *self.norm, rows_per_sample=S, n_samples=N, eps=1e-6, silu=False), self.proj_in)
        emb = self.frame_embedding(B, F, x.device)
        ldx = ctx.xattn
        for L in self.layers:
            # ---- BasicTransformerBlock: self-attn (+ collapsed cross-attn) ; GEGLU feed-forward
            # the Q third leaves the projection pre-multiplied by softmax scale * log2(e) (one rounding, no per-score
            # multiply in the attention kernel)
            hd = self.head_dim
            if hd == 64:
                if ops.ln_linear_fusable(h, L.qkv):          # C = 320: norm1 + Q | K | V in one launch, the normalised rows stay in registers
                    qkv = ops.ln_linear(h, *L.ln1, L.qkv, cs_cols=C, cs_scale=ops.attn_q_prescale(hd))
                else:
                    qkv = ops.igemm(ops.layernorm(h, *L.ln1), L.qkv, cs_cols=C, cs_scale=ops.attn_q_prescale(hd))
                a = ops.attn_spatial(qkv, N, S, heads, hd, q_prescaled=True)
            else:                                            # the general kernel (pt_attn_f16) on the fused projection's column blocks
                qkv = ops.igemm(ops.layernorm(h, *L.ln1), L.qkv)
                a = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], N, S, S, heads, hd)
            pre_ok = ops.FUSED_PRE and ops.ffn_fusable(L.ff1, L.ff2) and getattr(h, "lo", None) is None
            if pre_ok:                                       # out-projection + residual + cross-attention row, norm3 and ff in ONE launch
                hs = ops.ffn_geglu(a, L.ff1, L.ff2, pre=dict(w=L.o, res=h, vec=ldx[:, L.x_off:L.x_off + C], vec_mode=1, vG=F * S, ln=L.ln3))
            else:
                h = ops.igemm(a, L.o, res=h, vec=ldx[:, L.x_off:L.x_off + C], vec_mode=1, vG=F * S)
                hs = self._feed_forward(ops.layernorm(h, *L.ln3), L.ff1, L.ff2, N, S, res=h)
            # ---- TemporalBasicTransformerBlock on (hs + frame embedding)
            u = self._feed_forward(ops.layernorm(hs, *L.ln_in, vec=emb, vG=S), L.fi1, L.fi2, N, S, res=hs, vec=emb)
            qkv = ops.ln_linear(u, *L.tln1, L.tqkv) if ops.ln_linear_fusable(u, L.tqkv) else ops.igemm(ops.layernorm(u, *L.tln1), L.tqkv)
            a = ops.attn_temporal(qkv, B, F, S, heads, hd)
            if ctx.half is None:
                tbl = ldx
            else:                                            # global index ((b + half * B) * S + s) mod B_total with B_total = 2 B, B = 1
                tbl = ctx.xattn_full if (ctx.half * B * S) % ctx.xattn_full.shape[0] == 0 else ctx.xattn_full_swapped
            # ff(norm3(u)) + u, then AlphaBlender(hs, .)
            if ops.FUSED_PRE and ops.ffn_fusable(L.tf1, L.tf2) and getattr(u, "lo", None) is None:
                h = ops.ffn_geglu(a, L.tf1, L.tf2, blend=hs, alpha=self.alpha,
                                  pre=dict(w=L.to, res=u, vec=tbl[:, L.tx_off:L.tx_off + C], vec_mode=2, vFS=F * S, vS=S, vB=tbl.shape[0], ln=L.tln3))
            else:
                u = ops.igemm(a, L.to, res=u, vec=tbl[:, L.tx_off:L.tx_off + C], vec_mode=2, vFS=F * S, vS=S, vB=tbl.shape[0])
                h = self._feed_forward(ops.layernorm(u, *L.tln3), L.tf1, L.tf2, N, S, res=u, blend=hs, alpha=self.alpha)
        # (the block's own output is a plain fp16 tensor: widening it buys 1 % of the error for a fifth of the cost)
        y = ops.igemm(h, self.proj_out, res=xt)
        return y.view(N, H, W, C)


class DownBlock:
    """``CrossAttnDownBlockSpatioTemporal`` (eps 1e-6) / ``DownBlockSpatioTemporal`` (eps 1e-5)."""

    def __init__(self, sd, p, cross: bool, heads, device, temb_stack, xattn_stack):
        self.has_cross_attention = cross
        eps = 1e-6 if cross else 1e-5
        self.resnets, self.attentions = [], []
        j = 0
        while f"{p}resnets.{j}.spatial_res_block.norm1.weight" in sd:
            self.resnets.append(SpatioTemporalResBlock(sd, f"{p}resnets.{j}.", eps, device, temb_stack))
            if cross:
                self.attentions.append(TransformerSpatioTemporalModel(sd, f"{p}attentions.{j}.", heads, device, xattn_stack))
            j += 1
        self.down = None
        if p + "downsamplers.0.conv.weight" in sd:
            self.down = pack_conv2d(sd[p + "downsamplers.0.conv.weight"], sd[p + "downsamplers.0.conv.bias"], device,
                                    stride=2, padding=1)

    def run(self, ctx, x):
        taps = []
        for i, res in enumerate(self.resnets):
            x = res.run(ctx, x)
            if self.has_cross_attention:
                x = self.attentions[i].run(ctx, x)
            taps.append(x)
        if self.down is not None:
            N, H, W, C = x.shape
            y = ops.igemm(x, self.down, geom=(N, H, W))
            x = y.view(N, (H + 1) // 2, (W + 1) // 2, C)
            taps.append(x)
        return x, taps


class MidBlock:
    """``UNetMidBlockSpatioTemporal``: ResBlock, Transformer, ResBlock (eps 1e-5)."""
    has_cross_attention = True

    def __init__(self, sd, p, heads, device, temb_stack, xattn_stack):
        self.r0 = SpatioTemporalResBlock(sd, p + "resnets.0.", 1e-5, device, temb_stack)
        self.attn = TransformerSpatioTemporalModel(sd, p + "attentions.0.", heads, device, xattn_stack)
        self.r1 = SpatioTemporalResBlock(sd, p + "resnets.1.", 1e-5, device, temb_stack)

    def run(self, ctx, x):
        return self.r1.run(ctx, self.attn.run(ctx, self.r0.run(ctx, x)))


class UpBlock:
    """``UpBlockSpatioTemporal`` / ``CrossAttnUpBlockSpatioTemporal`` (resnet_eps = 1e-5, ``unet...:227``)."""

    def __init__(self, sd, p, cross: bool, heads, device, temb_stack, xattn_stack):
        self.has_cross_attention = cross
        self.resnets, self.attentions = [], []
        j = 0
        while f"{p}resnets.{j}.spatial_res_block.norm1.weight" in sd:
            self.resnets.append(SpatioTemporalResBlock(sd, f"{p}resnets.{j}.", 1e-5, device, temb_stack))
            if cross:
                self.attentions.append(TransformerSpatioTemporalModel(sd, f"{p}attentions.{j}.", heads, device, xattn_stack))
            j += 1
        self.up = None
        if p + "upsamplers.0.conv.weight" in sd:
            self.up = pack_conv2d(sd[p + "upsamplers.0.conv.weight"], sd[p + "upsamplers.0.conv.bias"], device)

    def run(self, ctx, x, skips: List[torch.Tensor]):
        skips = list(skips)
        for i, res in enumerate(self.resnets):
            skip = skips.pop()
            if tuple(skip.shape[:3]) != tuple(x.shape[:3]):  # torch.cat([hidden, skip], dim=1) of the reference raises here too
                raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {x.shape[1]}x{x.shape[2]} but got "
                                   f"size {skip.shape[1]}x{skip.shape[2]} for the skip connection (latent height / width must be "
                                   f"multiples of {2 ** 3})")
            x = res.run(ctx, x, x1=skip)                   # cat([hidden, skip], dim=1) folded into the gathers
            if self.has_cross_attention:
                x = self.attentions[i].run(ctx, x)
        if self.up is not None:
            N, H, W, C = x.shape
            x = ops.igemm(x, self.up, geom=(N, H, W), upsample2x=True).view(N, 2 * H, 2 * W, C)
        return x


class TimeEmbedding:
    """Timesteps + TimestepEmbedding flow of ``controlnet_sdv.py:551-590`` == ``unet...:387-426``."""

    def __init__(self, sd, ch0: int, add_dim: int, device):
        lin = lambda k: pack_linear(sd[k + ".weight"], sd[k + ".bias"], device)
        self.ch0, self.add_dim = ch0, add_dim
        self.t1, self.t2 = lin("time_embedding.linear_1"), lin("time_embedding.linear_2")
        self.a1, self.a2 = lin("add_embedding.linear_1"), lin("add_embedding.linear_2")

    def run(self, timesteps: torch.Tensor, added_time_ids: torch.Tensor, B: int) -> torch.Tensor:
        """-> silu(time_embedding(t) + add_embedding(ids)) as fp16 [B, 4*ch0]  (every consumer applies SiLU first)."""
        dev = added_time_ids.device
        t = timesteps.to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        emb = ops.igemm(ops.silu(ops.igemm(ops.timestep_embedding(t, self.ch0), self.t1)), self.t2)
        ids = added_time_ids.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        te = ops.timestep_embedding(ids, self.add_dim).view(B, -1)
        emb = ops.igemm(ops.silu(ops.igemm(te, self.a1)), self.a2, res=emb)
        return ops.silu(emb)
