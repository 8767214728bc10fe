"""Oracle restatement of the diffusers==0.24.0 blocks used by the hot path.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Pinning status: the COMPOSITE
forwards (``TemporalBasicTransformerBlock``, ``TransformerSpatioTemporalModel``,
``CrossAttnDown/UpBlockSpatioTemporal``) are pinned by a reference run -
``tests/golden/blocks.npz`` holds the outputs of the reference's own
``models/modified_svd.py`` forwards executed over this file's leaf modules, and
``tests/test_oracle_golden.py`` requires these forwards to reproduce them bit for
bit.  PARITY UNPINNED for the LEAVES (resnets, Attention, FeedForward/GEGLU,
AlphaBlender, Timesteps, eps values): ``diffusers`` (``/root/reference/requirements.txt:4``,
guard at ``scripts/train_svd_traj_VIPSeg_14.py:68``) is a third-party dependency that
is absent from ``/root/reference`` and from this image.  Sources followed, in
priority order:

1. ``/root/reference/models/modified_svd.py`` - in-tree copies of four
   diffusers forwards (temporal transformer block ``:50-114``, spatio-temporal
   transformer ``:118-223``, cross-attn up block ``:225-285``, cross-attn down
   block ``:287-348``).
2. the constructor call sites in ``models/controlnet_sdv.py:352-391`` and
   ``models/unet_spatio_temporal_condition_controlnet.py:169-232`` for every
   hyper-parameter.
3. the published behaviour of diffusers 0.24.0 (``models/resnet.py``,
   ``models/attention.py``, ``models/embeddings.py``,
   ``models/transformer_temporal.py``, ``models/unet_3d_blocks.py``) for
   internal defaults (eps values, GEGLU/erf-GELU, AlphaBlender).

Module attribute names reproduce the diffusers state-dict key families
 so a real checkpoint's keys map one-to-one.
All tensors are NCHW / ``[B*F, C, H, W]`` like the reference.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from .quant import q


# --------------------------------------------------------------------------- embeddings
def sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``.

    Call sites: ``controlnet_sdv.py:309,314,568,577``; ``unet...:137,142,404,413``;
    ``modified_svd.py:168-171``.  f_k = exp(-ln(1e4) * k / (dim/2)); out = [cos | sin], fp32.
    """
    half = dim // 2
    k = torch.arange(half, dtype=torch.float32, device=t.device)
    freq = torch.exp(-math.log(10000.0) * k / half)
    ang = t.reshape(-1, 1).to(torch.float32) * freq.reshape(1, -1)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class Timesteps(nn.Module):
    def __init__(self, num_channels: int, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0):
        super().__init__()
        assert flip_sin_to_cos and downscale_freq_shift == 0
        self.num_channels = num_channels

    def forward(self, t):
        return q(sinusoid(t, self.num_channels), True)


class TimestepEmbedding(nn.Module):
    """Linear -> SiLU -> Linear (``controlnet_sdv.py:312,315``; ``modified_svd.py:178``)."""

    def __init__(self, in_channels: int, time_embed_dim: int, out_dim: int | None = None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, x):
        return q(self.linear_2(q(F.silu(q(self.linear_1(x), True)), True)), True)


class AlphaBlender(nn.Module):
    """``merge_strategy="learned_with_images"``; the path always passes
    ``image_only_indicator == 0`` (``controlnet_sdv.py:602``, ``unet...:431``) so the
    blend weight is the scalar sigmoid(mix_factor):  a*x_spatial + (1-a)*x_temporal."""

    def __init__(self, alpha: float = 0.5):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([alpha], dtype=torch.float32))

    def forward(self, x_spatial, x_temporal, image_only_indicator):
        a = torch.where(
            image_only_indicator.bool(),
            torch.ones(1, 1, device=x_spatial.device, dtype=self.mix_factor.dtype),
            torch.sigmoid(self.mix_factor)[..., None],
        )
        if x_spatial.ndim == 5:          # [B, C, F, H, W]
            a = a[:, None, :, None, None]
        elif x_spatial.ndim == 3:        # [B*F, S, C]
            a = a.reshape(-1)[:, None, None]
        a = a.to(x_spatial.dtype)
        return q(a * x_spatial + (1.0 - a) * x_temporal, True, wide="rb" if x_spatial.dim() == 5 else None)    # resblock output: stream (pair)


# --------------------------------------------------------------------------- resnets
class ResnetBlock2D(nn.Module):
    """GN-SiLU-conv3x3 ; + Linear(SiLU(temb)) ; GN-SiLU-conv3x3 ; + shortcut (1x1 conv iff Cin != Cout)."""

    def __init__(self, in_channels, out_channels, temb_channels, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        # temb_channels=None: the VAE's blocks (oracle/vae.py) carry no time embedding
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        xb = q(x, True)               # a stream tensor enters a branch (norm / GEMM operand) as its fp16 high half
        h = q(self.conv1(q(F.silu(q(self.norm1(xb))), True)), self.time_emb_proj is None)
        if self.time_emb_proj is not None:
            t = q(self.time_emb_proj(q(F.silu(temb), True)), True)
            h = q(h + t[:, :, None, None], True)
        h = q(self.conv2(q(F.silu(q(self.norm2(h))), True)))
        if self.conv_shortcut is not None:
            x = q(self.conv_shortcut(xb), True, wide="sc")
        return q(x + h, True, wide="xs")


class TemporalResnetBlock(nn.Module):
    """Same shape of block on ``[B, C, F, H, W]`` with (3,1,1) convolutions; GroupNorm
    statistics therefore run over (C/32, F, H, W)."""

    def __init__(self, channels, temb_channels, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, channels, eps=eps, affine=True)
        self.conv1 = nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(32, channels, eps=eps, affine=True)
        self.conv2 = nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, x, temb):                       # temb [B, F, D]
        h = q(self.conv1(q(F.silu(q(self.norm1(q(x, True)))), True)), self.time_emb_proj is None)
        if self.time_emb_proj is not None:
            t = q(self.time_emb_proj(q(F.silu(temb), True)), True)          # [B, F, C]
            h = q(h + t.permute(0, 2, 1)[:, :, :, None, None], True)
        h = q(self.conv2(q(F.silu(q(self.norm2(h))), True)))
        return q(x + h)               # the MI355X path folds this add into the AlphaBlender epilogue


class SpatioTemporalResBlock(nn.Module):
    """spatial ResnetBlock2D -> [B,C,F,H,W] -> TemporalResnetBlock -> AlphaBlender.
    Invoked as ``resnet(hidden_states, temb, image_only_indicator=)`` (``modified_svd.py:268-272``)."""

    def __init__(self, in_channels, out_channels, temb_channels, eps):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps)
        self.temporal_res_block = TemporalResnetBlock(out_channels, temb_channels, eps)
        self.time_mixer = AlphaBlender(0.5)

    def forward(self, x, temb, image_only_indicator):
        nf = image_only_indicator.shape[-1]
        x = self.spatial_res_block(x, temb)
        bf, c, hh, ww = x.shape
        b = bf // nf
        xs = x.reshape(b, nf, c, hh, ww).permute(0, 2, 1, 3, 4)
        xt = self.temporal_res_block(xs, temb.reshape(b, nf, -1))
        y = self.time_mixer(xs, xt, image_only_indicator)
        return y.permute(0, 2, 1, 3, 4).reshape(bf, c, hh, ww)


# --------------------------------------------------------------------------- attention
class Attention(nn.Module):
    """softmax(Q K^T / sqrt(d)) V, no mask; bias-free q/k/v, biased out projection."""

    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv_dim, inner, bias=False)
        self.to_v = nn.Linear(kv_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def forward(self, x, encoder_hidden_states=None):
        ctx = x if encoder_hidden_states is None else encoder_hidden_states
        b, s, _ = x.shape
        cross = encoder_hidden_states is not None      # single-token cross-attention is hoisted on the MI355X path
        qq, k, v = q(self.to_q(x), not cross), q(self.to_k(ctx), not cross), q(self.to_v(ctx), not cross)
        d = qq.shape[-1] // self.heads
        qq = qq.view(b, -1, self.heads, d).transpose(1, 2)
        k = k.view(b, -1, self.heads, d).transpose(1, 2)
        v = v.view(b, -1, self.heads, d).transpose(1, 2)
        # same arithmetic sample by sample when the [b, heads, s, s] score tensor would not fit in memory
        # (b = 28 frames x 9216 tokens at the 576 x 1024 geometry: 9.5 GB per head in fp32)
        step = max(1, min(b, (1 << 27) // max(1, self.heads * s * k.shape[2])))
        # ... and query block by query block once one sample's score matrix outgrows the caches (rows of a softmax are
        # independent: the same arithmetic, 20 x faster on the host at S = 9216; below 2^20 scores per head - every golden
        # fixture and tiny test - the single-block path is taken and results stay bit-identical to earlier versions)
        klen = k.shape[2]
        qrows = s if s * klen <= (1 << 20) else max(64, (1 << 21) // (self.heads * klen))
        outs = []
        for i in range(0, b, step):
            ki, vi = k[i:i + step].transpose(-1, -2), v[i:i + step]
            parts = []
            for j in range(0, s, qrows):
                w = q(torch.softmax(q((qq[i:i + step, :, j:j + qrows] @ ki) / math.sqrt(d)), dim=-1))
                parts.append(w @ vi)
            outs.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=2))
        o = outs[0] if len(outs) == 1 else torch.cat(outs)
        o = q(o.transpose(1, 2).reshape(b, s, self.heads * d), not cross)
        return q(self.to_out[0](o), cross)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, g = q(self.proj(x)).chunk(2, dim=-1)
        return q(h * q(F.gelu(g)), True)            # erf GELU


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4):
        super().__init__()
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim_out or dim)])

    def forward(self, x):
        return q(self.net[2](self.net[0](x)))


class BasicTransformerBlock(nn.Module):
    """LN->self-attn->+ ; LN->cross-attn(ehs)->+ ; LN->GEGLU-FF->+   (called at ``modified_svd.py:193-196``)."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_attention_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states=None):
        x = q(self.attn1(q(self.norm1(x), True)) + x)        # one epilogue with the next add on the MI355X path
        x = q(self.attn2(q(self.norm2(x)), encoder_hidden_states) + x, True)
        x = q(self.ff(q(self.norm3(x), True)) + x, True)
        return x


class TemporalBasicTransformerBlock(nn.Module):
    """Follows ``modified_svd.py:57-114`` (camera branch ``:83-89`` is dead on this path)."""

    def __init__(self, dim, time_mix_inner_dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.is_res = dim == time_mix_inner_dim
        self.norm_in = nn.LayerNorm(dim, eps=1e-5)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim, eps=1e-5)
        self.attn1 = Attention(time_mix_inner_dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(time_mix_inner_dim, eps=1e-5)
        self.attn2 = Attention(time_mix_inner_dim, heads, dim_head, cross_attention_dim)
        self.norm3 = nn.LayerNorm(time_mix_inner_dim, eps=1e-5)
        self.ff = FeedForward(time_mix_inner_dim)

    def forward(self, x, num_frames, encoder_hidden_states=None):
        bf, s, c = x.shape
        b = bf // num_frames
        x = x.reshape(b, num_frames, s, c).permute(0, 2, 1, 3).reshape(b * s, num_frames, c)   # :64-66
        res = x
        x = self.ff_in(q(self.norm_in(q(x, True)), True))     # the MI355X LayerNorm rounds (h + emb) like the fp16 reference
        if self.is_res:
            x = q(x + res, True)
        x = q(self.attn1(q(self.norm1(x), True)) + x)
        x = q(self.attn2(q(self.norm2(x)), encoder_hidden_states) + x, True)                   # :92-95
        y = self.ff(q(self.norm3(x), True))
        x = q(y + x) if self.is_res else y                    # folded into the AlphaBlender epilogue on the MI355X path
        return x.reshape(b, s, num_frames, c).permute(0, 2, 1, 3).reshape(bf, s, c)            # :110-112


class TransformerSpatioTemporalModel(nn.Module):
    """Follows ``modified_svd.py:147-223`` including the batch-interleaved ``time_context``
    (``:152-159``) and the frame-index embedding added before the temporal block only."""

    def __init__(self, heads, dim_head, in_channels, num_layers=1, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner, inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_proj = Timesteps(in_channels, True, 0)
        self.time_mixer = AlphaBlender(0.5)
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states, image_only_indicator):
        bf, _, hh, ww = x.shape
        nf = image_only_indicator.shape[-1]
        b = bf // nf
        ctx = encoder_hidden_states
        first = ctx.reshape(b, nf, -1, ctx.shape[-1])[:, 0]                                    # [B, 1, D]
        tctx = first[None].broadcast_to(hh * ww, b, 1, ctx.shape[-1]).reshape(hh * ww * b, 1, ctx.shape[-1])
        res = x
        h = q(self.norm(q(x, True)), True)
        c = h.shape[1]
        h = q(self.proj_in(h.permute(0, 2, 3, 1).reshape(bf, hh * ww, c)), True)
        frame_idx = torch.arange(nf, device=x.device).repeat(b, 1).reshape(-1)
        emb = self.time_pos_embed(self.time_proj(frame_idx).to(h.dtype))[:, None, :]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            h = blk(h, encoder_hidden_states=encoder_hidden_states)
            hm = tblk(q(h + emb), num_frames=nf, encoder_hidden_states=tctx)
            h = self.time_mixer(h, hm, image_only_indicator)
        h = q(self.proj_out(h))
        return q(h.reshape(bf, hh, ww, c).permute(0, 3, 1, 2).contiguous() + res, True, wide="tr")      # :216 (NCHW memory, like the reference)


# --------------------------------------------------------------------------- samplers
class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return q(self.conv(q(x, True)), True, wide="ds")


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return q(self.conv(F.interpolate(q(x, True), scale_factor=2.0, mode="nearest")), True, wide="us")


# --------------------------------------------------------------------------- U-Net blocks
class CrossAttnDownBlockSpatioTemporal(nn.Module):
    """[(ResBlock eps=1e-6, Transformer)] x L, tap after each pair, Downsample2D, tap
    (forward cited at ``modified_svd.py:295-348``)."""
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, transformer_layers,
                 num_attention_heads, cross_attention_dim, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList()
        for i in range(num_layers):
            self.resnets.append(SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels,
                                                       temb_channels, eps=1e-6))
            self.attentions.append(TransformerSpatioTemporalModel(
                num_attention_heads, out_channels // num_attention_heads, out_channels,
                num_layers=transformer_layers, cross_attention_dim=cross_attention_dim))
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb, encoder_hidden_states, image_only_indicator):
        taps = ()
        for res, attn in zip(self.resnets, self.attentions):
            hidden_states = res(hidden_states, temb, image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)
            taps += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            taps += (hidden_states,)
        return hidden_states, taps


class DownBlockSpatioTemporal(nn.Module):
    """ResBlock(eps=1e-5) x L with taps; the path's last level has no downsampler
    (``unet...:167,176``)."""
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps=1e-5)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb, image_only_indicator):
        taps = ()
        for res in self.resnets:
            hidden_states = res(hidden_states, temb, image_only_indicator)
            taps += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            taps += (hidden_states,)
        return hidden_states, taps


class UNetMidBlockSpatioTemporal(nn.Module):
    """ResBlock, Transformer, ResBlock (eps 1e-5); constructed at ``unet...:185-191``."""
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, transformer_layers_per_block, num_attention_heads,
                 cross_attention_dim):
        super().__init__()
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5),
            SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5)])
        self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(
            num_attention_heads, in_channels // num_attention_heads, in_channels,
            num_layers=transformer_layers_per_block, cross_attention_dim=cross_attention_dim)])

    def forward(self, hidden_states, temb, encoder_hidden_states, image_only_indicator):
        hidden_states = self.resnets[0](hidden_states, temb, image_only_indicator)
        hidden_states = self.attentions[0](hidden_states, encoder_hidden_states, image_only_indicator)
        return self.resnets[1](hidden_states, temb, image_only_indicator)


class _UpBase(nn.Module):
    def _make_resnets(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, eps):
        rs = nn.ModuleList()
        for i in range(num_layers):
            skip_c = in_channels if i == num_layers - 1 else out_channels
            in_c = prev_output_channel if i == 0 else out_channels
            rs.append(SpatioTemporalResBlock(in_c + skip_c, out_channels, temb_channels, eps=eps))
        return rs


class UpBlockSpatioTemporal(_UpBase):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, resnet_eps,
                 add_upsample):
        super().__init__()
        self.resnets = self._make_resnets(in_channels, prev_output_channel, out_channels, temb_channels,
                                          num_layers, resnet_eps)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, image_only_indicator):
        for res in self.resnets:
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = res(torch.cat([hidden_states, skip], dim=1), temb, image_only_indicator)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


class CrossAttnUpBlockSpatioTemporal(_UpBase):
    """pop skip, cat(dim=1), ResBlock, Transformer (x L); Upsample2D (``modified_svd.py:234-285``)."""
    has_cross_attention = True

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                 transformer_layers, resnet_eps, num_attention_heads, cross_attention_dim, add_upsample):
        super().__init__()
        self.resnets = self._make_resnets(in_channels, prev_output_channel, out_channels, temb_channels,
                                          num_layers, resnet_eps)
        self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(
            num_attention_heads, out_channels // num_attention_heads, out_channels,
            num_layers=transformer_layers, cross_attention_dim=cross_attention_dim) for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, image_only_indicator):
        for res, attn in zip(self.resnets, self.attentions):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = res(torch.cat([hidden_states, skip], dim=1), temb, image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


def get_down_block(down_block_type, num_layers, transformer_layers_per_block, in_channels, out_channels,
                   temb_channels, add_downsample, resnet_eps, cross_attention_dim, num_attention_heads,
                   resnet_act_fn="silu"):
    """Signature of the call at ``controlnet_sdv.py:352-364`` / ``unet...:169-181``.  ``resnet_eps`` and
    ``resnet_act_fn`` are accepted and ignored by the spatio-temporal down blocks, as in diffusers 0.24.0."""
    if down_block_type == "DownBlockSpatioTemporal":
        return DownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers, add_downsample)
    if down_block_type == "CrossAttnDownBlockSpatioTemporal":
        return CrossAttnDownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers,
                                                transformer_layers_per_block, num_attention_heads,
                                                cross_attention_dim, add_downsample)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type, num_layers, transformer_layers_per_block, in_channels, out_channels,
                 prev_output_channel, temb_channels, add_upsample, resnet_eps, resolution_idx=None,
                 cross_attention_dim=None, num_attention_heads=None, resnet_act_fn="silu"):
    """Signature of the call at ``unet...:218-232``."""
    if up_block_type == "UpBlockSpatioTemporal":
        return UpBlockSpatioTemporal(in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                                     resnet_eps, add_upsample)
    if up_block_type == "CrossAttnUpBlockSpatioTemporal":
        return CrossAttnUpBlockSpatioTemporal(in_channels, prev_output_channel, out_channels, temb_channels,
                                              num_layers, transformer_layers_per_block, resnet_eps,
                                              num_attention_heads, cross_attention_dim, add_upsample)
    raise ValueError(f"{up_block_type} does not exist.")
