"""Oracle restatement of ``diffusers==0.24.0``'s ``AutoencoderKLTemporalDecoder`` - the ``vae`` of the reference pipeline
(``/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:26,124,174-195,225-251``) - and of the two in-tree
functions around it, ``decode_latents`` (``:225-251``) and ``tensor2vid`` (``:70-83``).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Pinning status
  PINNED (reference run, ``tests/golden/vae_io.npz``): ``decode_latents`` and ``tensor2vid`` - the reference's own functions
  are executed in ``tests/golden/make_golden.py`` over THIS file's decoder and compared with ``decode_latents`` /
  ``tensor2vid`` below: ``1 / scaling_factor``, the ``decode_chunk_size`` slicing with ``num_frames`` = frames in the chunk,
  the ``[B*F, C, H, W] -> [B, C, F, H, W]`` reshape / permute, ``.float()``, the per-clip ``postprocess`` call.
  PARITY UNPINNED: the networks themselves.  ``diffusers`` (``/root/reference/requirements.txt:4``) is a third-party
  dependency that is absent from ``/root/reference`` and from this image; ``Encoder`` / ``TemporalDecoder`` / ``Attention``
  / ``VaeImageProcessor.postprocess`` below restate its published 0.24.0 behaviour
  (``models/autoencoder_kl_temporal_decoder.py``, ``models/vae.py``, ``models/unet_2d_blocks.py``,
  ``models/unet_3d_blocks.py``, ``models/resnet.py``, ``models/attention_processor.py``, ``image_processor.py``):

  * ``Encoder``: conv_in 3x3; four ``DownEncoderBlock2D`` (2 x ResnetBlock2D(eps 1e-6, no temb); ``Downsample2D`` with
    ``padding=0``: F.pad (0,1,0,1) then conv 3x3 stride 2) - none on the last; ``UNetMidBlock2D`` (resnet, single-head
    attention of head_dim = channels, resnet); GroupNorm(32, eps 1e-6) + SiLU + conv_out 3x3 -> 2 * latent channels;
    ``quant_conv`` 1x1; ``DiagonalGaussianDistribution`` (mean | logvar, ``mode()`` = mean).
  * ``TemporalDecoder``: conv_in 3x3; ``MidBlockTemporalDecoder`` (SpatioTemporalResBlock, attention, SpatioTemporalResBlock);
    four ``UpBlockTemporalDecoder`` (3 x SpatioTemporalResBlock, ``Upsample2D`` = nearest 2x + conv 3x3 except on the last);
    GroupNorm(32, eps 1e-6) + SiLU + conv_out 3x3; ``time_conv_out`` Conv3d (3,1,1) over the frames of one decode call.
    Its ``SpatioTemporalResBlock``s have no time embedding, eps 1e-6 (spatial) / 1e-5 (temporal),
    ``merge_strategy="learned"`` with ``switch_spatial_to_temporal_mix=True``: out = (1 - s) x_spatial + s x_temporal,
    s = sigmoid(mix_factor).
  * ``Attention`` (``_from_deprecated_attn_block``): GroupNorm(32, eps 1e-6) on the tokens, biased q / k / v / out
    projections, one head, ``+ residual``, ``rescale_output_factor = 1``.

Module attribute names reproduce the diffusers state-dict keys ([UNVERIFIED-MEMORY], like SURVEY Appendix C).
"""
from __future__ import annotations

import inspect
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import blocks as B
from .quant import q


# --------------------------------------------------------------------------------------- leaves
class VaeAttention(nn.Module):
    """Single-head self-attention over the h*w tokens of one frame, with its own GroupNorm and residual."""

    def __init__(self, channels, head_dim, groups=32, eps=1e-6):
        super().__init__()
        self.heads = channels // head_dim
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps, affine=True)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def forward(self, x):
        n, c, hh, ww = x.shape
        res = x
        t = q(x, True).view(n, c, hh * ww)
        t = q(self.group_norm(t), True).transpose(1, 2)                      # [n, S, c]
        qq, k, v = q(self.to_q(t), True), q(self.to_k(t), True), q(self.to_v(t), True)
        d = c // self.heads
        split = lambda u: u.view(n, hh * ww, self.heads, d).transpose(1, 2)
        outs = []
        for i in range(n):                                                    # frame by frame: S x S scores (S = 9216 at 576 x 1024)
            s = hh * ww
            rows = s if s * s <= (1 << 22) else max(64, (1 << 22) // s)
            parts = []
            kk, vv = split(k)[i:i + 1].transpose(-1, -2), split(v)[i:i + 1]
            for j in range(0, s, rows):
                w = q(torch.softmax(q((split(qq)[i:i + 1, :, j:j + rows] @ kk) / d ** 0.5), dim=-1))
                parts.append(w @ vv)
            outs.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=2))
        o = q(torch.cat(outs).transpose(1, 2).reshape(n, hh * ww, c), True)
        o = self.to_out[0](o)
        o = o.transpose(1, 2).reshape(n, c, hh, ww)
        return q(o + res, True, wide="at")


class Downsample2D(nn.Module):
    """``Downsample2D(use_conv=True, padding=0)``: zero row / column appended at the bottom / right, conv 3x3 stride 2."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=0)

    def forward(self, x):
        return q(self.conv(F.pad(q(x, True), (0, 1, 0, 1), mode="constant", value=0.0)), True, wide="ds")


class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([B.ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, None, eps=1e-6)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class UNetMidBlock2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.resnets = nn.ModuleList([B.ResnetBlock2D(channels, channels, None, eps=1e-6) for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(channels, channels)])

    def forward(self, x):
        x = self.resnets[0](x, None)
        x = self.attentions[0](x)
        return self.resnets[1](x, None)


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, double_z=True):
        super().__init__()
        ch = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out_c = ch[0]
        for i, c in enumerate(ch):
            in_c, out_c = out_c, c
            self.down_blocks.append(DownEncoderBlock2D(in_c, out_c, layers_per_block, add_downsample=i != len(ch) - 1))
        self.mid_block = UNetMidBlock2D(ch[-1])
        self.conv_norm_out = nn.GroupNorm(32, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * out_channels if double_z else out_channels, 3, padding=1)

    def forward(self, x):
        x = q(self.conv_in(q(x, True)), True, wide="ci")
        for blk in self.down_blocks:
            x = blk(x)
        x = self.mid_block(x)
        x = q(F.silu(q(self.conv_norm_out(q(x, True)))), True)
        return q(self.conv_out(x), True)


class VaeSpatioTemporalResBlock(nn.Module):
    """``SpatioTemporalResBlock(temb_channels=None, eps=1e-6, temporal_eps=1e-5, merge_factor=0.0, merge_strategy="learned",
    switch_spatial_to_temporal_mix=True)``."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.spatial_res_block = B.ResnetBlock2D(in_channels, out_channels, None, eps=1e-6)
        self.temporal_res_block = B.TemporalResnetBlock(out_channels, None, eps=1e-5)
        self.time_mixer = nn.Module()
        self.time_mixer.mix_factor = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x, image_only_indicator):
        nf = image_only_indicator.shape[-1]
        x = self.spatial_res_block(x, None)
        bf, c, hh, ww = x.shape
        b = bf // nf
        xs = x.reshape(b, nf, c, hh, ww).permute(0, 2, 1, 3, 4)
        xt = self.temporal_res_block(xs, None)
        a = 1.0 - torch.sigmoid(self.time_mixer.mix_factor).to(xs.dtype)      # "learned", switched
        y = q(a * xs + (1.0 - a) * xt, True, wide="rb")
        return y.permute(0, 2, 1, 3, 4).reshape(bf, c, hh, ww)


class MidBlockTemporalDecoder(nn.Module):
    def __init__(self, channels, num_layers):
        super().__init__()
        self.resnets = nn.ModuleList([VaeSpatioTemporalResBlock(channels, channels) for _ in range(num_layers)])
        self.attentions = nn.ModuleList([VaeAttention(channels, channels)])

    def forward(self, x, image_only_indicator):
        x = self.resnets[0](x, image_only_indicator)
        for r, a in zip(self.resnets[1:], self.attentions):
            x = a(x)
            x = r(x, image_only_indicator)
        return x


class UpBlockTemporalDecoder(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([VaeSpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels)
                                      for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([B.Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, x, image_only_indicator):
        for r in self.resnets:
            x = r(x, image_only_indicator)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class TemporalDecoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block):
        super().__init__()
        ch = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, ch[-1], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(ch[-1], layers_per_block)
        self.up_blocks = nn.ModuleList()
        rch = ch[::-1]
        out_c = rch[0]
        for i, c in enumerate(rch):
            prev, out_c = out_c, c
            self.up_blocks.append(UpBlockTemporalDecoder(prev, out_c, layers_per_block + 1, add_upsample=i != len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(32, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, sample, image_only_indicator, num_frames=1):
        x = q(self.conv_in(q(sample, True)), True, wide="ci")
        x = self.mid_block(x, image_only_indicator)
        for blk in self.up_blocks:
            x = blk(x, image_only_indicator)
        x = q(F.silu(q(self.conv_norm_out(q(x, True)))), True)
        x = q(self.conv_out(x))                                  # the MI355X path keeps this 3-channel image in fp32
        bf, c, hh, ww = x.shape
        b = bf // num_frames
        x = x.reshape(b, num_frames, c, hh, ww).permute(0, 2, 1, 3, 4)
        x = self.time_conv_out(x)
        return x.permute(0, 2, 1, 3, 4).reshape(bf, c, hh, ww)


class DiagonalGaussianDistribution:
    def __init__(self, parameters):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)


def svd_vae_config():
    """``vae/config.json`` of stabilityai/stable-video-diffusion-img2vid [UNVERIFIED-MEMORY: not in the reference tree]."""
    return dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, sample_size=768,
                scaling_factor=0.18215, force_upcast=True)


def tiny_vae_config():
    return dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=4, sample_size=64,
                scaling_factor=0.18215, force_upcast=True)


class AutoencoderKLTemporalDecoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",), block_out_channels=(64,),
                 layers_per_block=1, latent_channels=4, sample_size=32, scaling_factor=0.18215, force_upcast=True):
        super().__init__()
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, double_z=True)
        self.decoder = TemporalDecoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def encode(self, x, return_dict=True):
        moments = q(self.quant_conv(self.encoder(x)))
        post = DiagonalGaussianDistribution(moments)
        return SimpleNamespace(latent_dist=post) if return_dict else (post,)

    def decode(self, z, num_frames, return_dict=True):
        batch = z.shape[0] // num_frames
        ind = torch.zeros(batch, num_frames, dtype=z.dtype, device=z.device)
        dec = self.decoder(z, num_frames=num_frames, image_only_indicator=ind)
        return SimpleNamespace(sample=dec) if return_dict else (dec,)

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None, num_frames=1):
        post = self.encode(sample).latent_dist
        z = post.sample(generator) if sample_posterior else post.mode()
        dec = self.decode(z, num_frames=num_frames).sample
        return SimpleNamespace(sample=dec) if return_dict else (dec,)


# --------------------------------------------------------------------------------------- in-tree functions around the VAE
def decode_latents(vae, latents, num_frames, decode_chunk_size=14):
    """``pipeline...:225-251``.  ``latents`` ``[B, F, 4, h, w]`` -> ``[B, 3, F, 8h, 8w]`` fp32.  Every chunk is decoded as ONE
    clip of ``len(chunk)`` frames (``num_frames`` = frames in the chunk, ``:238-244``): the temporal layers never see across
    a chunk boundary, and a chunk may span two clips when ``B > 1``."""
    latents = latents.flatten(0, 1)
    latents = 1 / vae.config.scaling_factor * latents
    accepts_num_frames = "num_frames" in set(inspect.signature(vae.forward).parameters.keys())
    frames = []
    for i in range(0, latents.shape[0], decode_chunk_size):
        n_in = latents[i:i + decode_chunk_size].shape[0]
        kw = dict(num_frames=n_in) if accepts_num_frames else {}
        frames.append(vae.decode(latents[i:i + decode_chunk_size], **kw).sample)
    frames = torch.cat(frames, dim=0)
    frames = frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4)
    return frames.float()


def postprocess(image: torch.Tensor, output_type: str = "pil"):
    """``VaeImageProcessor.postprocess`` (diffusers 0.24.0, ``do_normalize=True``) [UNVERIFIED-MEMORY]: ``[F, 3, H, W]`` in
    [-1, 1] -> ``(x / 2 + 0.5).clamp(0, 1)``; "pt": that tensor; "np": ``[F, H, W, 3]`` float32; "pil": list of
    ``PIL.Image`` from ``(x * 255).round().astype(uint8)``; "latent": the input."""
    if output_type == "latent":
        return image
    image = (image / 2 + 0.5).clamp(0, 1)
    if output_type == "pt":
        return image
    arr = image.cpu().permute(0, 2, 3, 1).float().numpy()
    if output_type == "np":
        return arr
    if output_type == "pil":
        import PIL.Image
        u8 = (arr * 255).round().astype("uint8")
        return [PIL.Image.fromarray(a) for a in u8]
    raise ValueError(f"output_type {output_type!r}")


def tensor2vid(video: torch.Tensor, processor=None, output_type="np"):
    """``pipeline...:70-83``: per clip ``[3, F, H, W] -> [F, 3, H, W] -> postprocess``; returns a list (one entry per clip)."""
    post = processor.postprocess if processor is not None else postprocess
    outs = []
    for b in range(video.shape[0]):
        outs.append(post(video[b].permute(1, 0, 2, 3), output_type))
    return outs
