"""MI355X-native ``AutoencoderKLTemporalDecoder`` - the ``vae`` of the reference pipeline
(``/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:26,124``: ``vae.encode(image).latent_dist.mode()``
at ``:182``, ``vae.decode(latents, num_frames=).sample`` at ``:243``).  The class is diffusers 0.24.0's (not in the reference
tree); constructor arguments, ``encode`` / ``decode`` / ``forward`` signatures, config fields (``scaling_factor``,
``force_upcast``, ``block_out_channels``) and state-dict keys follow it (sources: DESIGN.md section 2).

Executed like the U-Net (``blocks.py``): channels-last fp16 activations, fp32 accumulation; every 3x3 / (3,1,1) / 1x1
convolution and projection is ``pt_igemm_f16`` (the decoder's temporal convolutions see the image ``(F, H*W)`` - 589 824
columns at 576 x 1024), GroupNorm(+SiLU) ``pt_groupnorm_*``, the mid blocks' single 512-wide attention head ``pt_attn_f16``;
``time_conv_out`` runs in fp32 fused with the layout change ``decode_latents`` needs (``pt_vae_time_conv_out``).
``dtype`` is reported as fp16 and ``force_upcast`` is honoured as a no-op: the kernels accumulate in fp32 whatever the
storage type, so the reference's fp32 detour around ``encode`` (``pipeline...:454-463``) has nothing to switch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import blocks as B
from . import ops, spec
from .modeling import BaseOutput, HipModel
from .packing import pack_conv2d, pack_linear, vec16


class AutoencoderKLOutput(BaseOutput):
    """``latent_dist``."""


class DecoderOutput(BaseOutput):
    """``sample``."""


class DiagonalGaussianDistribution:
    """``parameters`` fp32 ``[N, 2C, h, w]`` = (mean | logvar) on the device."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        c = parameters.shape[1] // 2
        self.mean = parameters[:, :c]
        self.logvar = parameters[:, c:]

    def mode(self) -> torch.Tensor:
        return self.mean

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        from .pipeline_stable_video_diffusion_controlnet import randn_tensor
        noise = randn_tensor(tuple(self.mean.shape), generator=generator, device=self.parameters.device, dtype=torch.float32)
        return ops.gaussian_sample(self.parameters, noise)


class _ResnetBlock2D:
    """``ResnetBlock2D(temb_channels=None, eps=1e-6)`` of the encoder: GN-SiLU-conv3x3, GN-SiLU-conv3x3, + shortcut."""

    def __init__(self, sd, p, device, eps=1e-6):
        self.eps = eps
        self.n1 = (vec16(sd[p + "norm1.weight"], device), vec16(sd[p + "norm1.bias"], device))
        self.conv1 = pack_conv2d(sd[p + "conv1.weight"], sd[p + "conv1.bias"], device)
        self.n2 = (vec16(sd[p + "norm2.weight"], device), vec16(sd[p + "norm2.bias"], device))
        self.conv2 = pack_conv2d(sd[p + "conv2.weight"], sd[p + "conv2.bias"], device)
        self.shortcut = None
        if p + "conv_shortcut.weight" in sd:
            self.shortcut = pack_conv2d(sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"], device, padding=0)

    def run(self, x: torch.Tensor) -> torch.Tensor:
        N, H, W, _ = x.shape
        S, Cc, geom = H * W, self.conv1.N, (N, H, W)
        y = ops.groupnorm(x, *self.n1, rows_per_sample=S, n_samples=N, eps=self.eps, silu=True)
        h = ops.igemm(y.view(N, H, W, -1), self.conv1, geom=geom)
        y = ops.groupnorm(h.view(N, H, W, Cc), *self.n2, rows_per_sample=S, n_samples=N, eps=self.eps, silu=True)
        sc = ops.igemm(x, self.shortcut, geom=geom, wide="sc" in ops.WIDE_KINDS) if self.shortcut is not None else ops.wview(x, N * S, Cc)
        out = ops.igemm(y.view(N, H, W, Cc), self.conv2, geom=geom, res=sc, wide="xs" in ops.WIDE_KINDS)
        return ops.wview(out, N, H, W, Cc)


class _Attention:
    """diffusers ``Attention`` of the VAE mid blocks: GroupNorm(32, eps 1e-6), biased q / k / v / out projections, ONE head
    of ``channels`` dims over the h*w tokens of a frame, + residual."""

    def __init__(self, sd, p, device):
        self.gn = (vec16(sd[p + "group_norm.weight"], device), vec16(sd[p + "group_norm.bias"], device))
        self.qkv = pack_linear(torch.cat([sd[p + "to_q.weight"], sd[p + "to_k.weight"], sd[p + "to_v.weight"]], 0),
                               torch.cat([sd[p + "to_q.bias"], sd[p + "to_k.bias"], sd[p + "to_v.bias"]], 0), device)
        self.o = pack_linear(sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"], device)
        self.C = self.o.N

    def run(self, x: torch.Tensor) -> torch.Tensor:
        N, H, W, Cc = x.shape
        S = H * W
        y = ops.groupnorm(x, *self.gn, rows_per_sample=S, n_samples=N, eps=1e-6, silu=False)
        qkv = ops.igemm(y, self.qkv)
        a = ops.attention(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], N, S, S, 1, Cc)
        return ops.igemm(a, self.o, res=ops.wview(x, N * S, Cc)).view(N, H, W, Cc)


class AutoencoderKLTemporalDecoder(HipModel):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, down_block_types: Tuple[str] = ("DownEncoderBlock2D",),
                 block_out_channels: Tuple[int] = (64,), layers_per_block: int = 1, latent_channels: int = 4,
                 sample_size: int = 32, scaling_factor: float = 0.18215, force_upcast: float = True):
        if any(t != "DownEncoderBlock2D" for t in down_block_types) or len(down_block_types) != len(block_out_channels):
            raise ValueError(f"down_block_types must be one 'DownEncoderBlock2D' per entry of block_out_channels; got {down_block_types}")
        if any(c % 32 for c in block_out_channels):
            raise ValueError(f"block_out_channels must be multiples of 32 (GroupNorm groups); got {block_out_channels}")
        if block_out_channels[-1] not in (64, 128, 512):
            raise ValueError(f"the mid blocks attend with ONE head of {block_out_channels[-1]} channels; pt_attn_f16 handles 64, 128, 512")
        super().__init__(in_channels=in_channels, out_channels=out_channels, down_block_types=tuple(down_block_types),
                         block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                         latent_channels=latent_channels, sample_size=sample_size, scaling_factor=scaling_factor,
                         force_upcast=force_upcast)

    def param_spec(self):
        return spec.vae_spec(self.config)

    # ------------------------------------------------------------------------------------------------ packing
    def _pack(self, sd, device):
        cfg = self.config
        n, L = len(cfg.block_out_channels), cfg.layers_per_block
        conv = lambda k, **kw: pack_conv2d(sd[k + ".weight"], sd[k + ".bias"], device, **kw)
        norm = lambda k: (vec16(sd[k + ".weight"], device), vec16(sd[k + ".bias"], device))
        # encoder
        self.e_conv_in = conv("encoder.conv_in")
        self.e_down = []
        for i in range(n):
            res = [_ResnetBlock2D(sd, f"encoder.down_blocks.{i}.resnets.{j}.", device) for j in range(L)]
            down = conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0) if i != n - 1 else None
            self.e_down.append((res, down))
        self.e_mid = (_ResnetBlock2D(sd, "encoder.mid_block.resnets.0.", device),
                      _Attention(sd, "encoder.mid_block.attentions.0.", device),
                      _ResnetBlock2D(sd, "encoder.mid_block.resnets.1.", device))
        self.e_norm_out = norm("encoder.conv_norm_out")
        self.e_conv_out = conv("encoder.conv_out")
        self.quant_conv = conv("quant_conv", padding=0)
        # decoder
        rb = lambda p: B.SpatioTemporalResBlock(sd, p, 1e-6, device, None, eps_t=1e-5, switch=True)
        self.d_conv_in = conv("decoder.conv_in")
        self.d_mid_res = [rb(f"decoder.mid_block.resnets.{j}.") for j in range(L)]
        self.d_mid_attn = _Attention(sd, "decoder.mid_block.attentions.0.", device)
        self.d_up = []
        for i in range(n):
            res = [rb(f"decoder.up_blocks.{i}.resnets.{j}.") for j in range(L + 1)]
            up = conv(f"decoder.up_blocks.{i}.upsamplers.0.conv") if i != n - 1 else None
            self.d_up.append((res, up))
        self.d_norm_out = norm("decoder.conv_norm_out")
        self.d_conv_out = conv("decoder.conv_out")
        oc = cfg.out_channels
        if oc != 3:
            raise ValueError("time_conv_out is built for 3 output channels (RGB frames)")
        w = sd["decoder.time_conv_out.weight"].detach().float().cpu().reshape(oc, oc, 3).contiguous()
        b = sd["decoder.time_conv_out.bias"].detach().float().cpu().contiguous()
        self._tw = (C.c_float * 27)(*w.flatten().tolist())                   # host side: passed to the kernel by value
        self._tb = (C.c_float * 3)(*b.tolist())

    # ------------------------------------------------------------------------------------------------ API
    def _check(self, x, what):
        if not self._loaded:
            raise RuntimeError("AutoencoderKLTemporalDecoder: no weights loaded (load_state_dict / from_pretrained / init_random_)")
        if not torch.is_tensor(x) or x.dim() != 4:
            raise ValueError(f"{what} must be a [batch, channels, height, width] tensor")
        if not x.is_cuda:
            raise RuntimeError("posetraj_amd: inputs must be on the ROCm device (no CPU path exists)")

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """``[N, 3, H, W]`` in [-1, 1] -> ``AutoencoderKLOutput(latent_dist=DiagonalGaussianDistribution)`` with fp32
        parameters ``[N, 8, H/8, W/8]``."""
        self._check(x, "x")
        N, Cin, H, W = x.shape
        cfg = self.config
        if Cin != cfg.in_channels or H % 2 ** (len(cfg.block_out_channels) - 1) or W % 2 ** (len(cfg.block_out_channels) - 1):
            raise ValueError(f"encode: {Cin} channels / {H} x {W} pixels do not fit the encoder ({cfg.in_channels} channels, "
                             f"{len(cfg.block_out_channels) - 1} halvings)")
        h = ops.to_channels_last(x, cpad=self.e_conv_in.cin)
        h = ops.igemm(h, self.e_conv_in, geom=(N, H, W)).view(N, H, W, -1)
        for res, down in self.e_down:
            for r in res:
                h = r.run(h)
            if down is not None:                               # Downsample2D(padding=0): F.pad (0,1,0,1) + conv stride 2
                n_, hh, ww, c = h.shape
                h = ops.igemm(h, down, geom=(n_, hh, ww), out_hw=(hh // 2, ww // 2)).view(n_, hh // 2, ww // 2, c)
        h = self.e_mid[2].run(self.e_mid[1].run(self.e_mid[0].run(h)))
        n_, hh, ww, c = h.shape
        y = ops.groupnorm(h, *self.e_norm_out, rows_per_sample=hh * ww, n_samples=n_, eps=1e-6, silu=True)
        m = ops.igemm(y.view(n_, hh, ww, c), self.e_conv_out, geom=(n_, hh, ww))
        z2 = self.e_conv_out.N
        mom = ops.igemm(m.view(n_, hh, ww, z2), self.quant_conv, geom=(n_, hh, ww), out_f32=True)        # fp32 [M, 2z]
        params = ops.to_nchw_f32(mom, n_, hh * ww, z2).view(n_, z2, hh, ww)
        post = DiagonalGaussianDistribution(params)
        return AutoencoderKLOutput(latent_dist=post) if return_dict else (post,)

    def decode(self, z: torch.Tensor, num_frames: int, return_dict: bool = True, out: Optional[torch.Tensor] = None):
        """``z`` ``[batch*num_frames, 4, h, w]`` (already divided by ``scaling_factor``) -> ``DecoderOutput(sample)`` with
        ``sample`` fp32 ``[batch*num_frames, 3, 8h, 8w]``.  ``out``: a contiguous fp32 buffer of that shape to write into
        (``decode_latents`` hands in slices of its frame buffer, so no concatenation pass exists)."""
        self._check(z, "z")
        NF, Cz, h, w = z.shape
        if Cz != self.config.latent_channels:
            raise ValueError(f"decode: {Cz} latent channels, the decoder takes {self.config.latent_channels}")
        if num_frames < 1 or NF % num_frames:
            raise ValueError(f"decode: {NF} frames do not split into clips of num_frames={num_frames}")
        ctx = B.Ctx(B=NF // num_frames, F=num_frames, temb=None, xattn=None)
        x = ops.to_channels_last(z, cpad=self.d_conv_in.cin)
        x = ops.igemm(x, self.d_conv_in, geom=(NF, h, w)).view(NF, h, w, -1)
        x = self.d_mid_res[0].run(ctx, x)
        for r in self.d_mid_res[1:2]:                          # MidBlockTemporalDecoder: zip(resnets[1:], attentions) with ONE
            x = self.d_mid_attn.run(x)                         # attention - the zip stops after resnets[1] whatever layers_per_block
            x = r.run(ctx, x)
        for res, up in self.d_up:
            for r in res:
                x = r.run(ctx, x)
            if up is not None:
                n_, hh, ww, c = x.shape
                x = ops.igemm(x, up, geom=(n_, hh, ww), upsample2x=True).view(n_, 2 * hh, 2 * ww, c)
        n_, hh, ww, c = x.shape
        y = ops.groupnorm(x, *self.d_norm_out, rows_per_sample=hh * ww, n_samples=n_, eps=1e-6, silu=True)
        img = torch.empty((n_ * hh * ww, 4), dtype=torch.float32, device=x.device)          # 3 channels, row pitch 4
        ops.igemm(y.view(n_, hh, ww, c), self.d_conv_out, geom=(n_, hh, ww), out_f32=True, out=img)
        if out is None:
            out = torch.empty((NF, 3, hh, ww), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (NF, 3, hh, ww) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError(f"decode: `out` must be a contiguous fp32 {(NF, 3, hh, ww)} tensor")
        HW = hh * ww
        for b in range(NF // num_frames):                      # the (3,1,1) convolution runs inside each clip of the call
            ops.vae_time_conv_out(img[b * num_frames * HW:(b + 1) * num_frames * HW], self._tw, self._tb, num_frames, HW,
                                  out[b * num_frames:(b + 1) * num_frames])
        return DecoderOutput(sample=out) if return_dict else (out,)

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True,
                generator: Optional[torch.Generator] = None, num_frames: int = 1):
        post = self.encode(sample).latent_dist
        z = post.sample(generator) if sample_posterior else post.mode()
        dec = self.decode(z, num_frames=num_frames).sample
        return DecoderOutput(sample=dec) if return_dict else (dec,)
