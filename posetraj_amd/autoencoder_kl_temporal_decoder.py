"""MI355X-native ``AutoencoderKLTemporalDecoder`` - the ``vae`` of the reference pipeline
(``/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:26,124``: ``vae.encode(image).latent_dist.mode()``
at ``:182``, ``vae.decode(latents, num_frames=).sample`` at ``:243``).  The class is diffusers 0.24.0's (not in the reference
tree); constructor arguments, ``encode`` / ``decode`` / ``forward`` signatures, config fields (``scaling_factor``,
``force_upcast``, ``block_out_channels``) and state-dict keys follow it (sources: DESIGN.md section 2).

Executed like the U-Net (``blocks.py``): channels-last fp16 activations, fp32 accumulation; every 3x3 / (3,1,1) / 1x1
convolution and projection is ``pt_igemm_f16`` (the decoder's temporal convolutions see the image ``(F, H*W)`` - 589 824
columns at 576 x 1024), GroupNorm(+SiLU) ``pt_groupnorm_*``, the mid blocks' single 512-wide attention head ``pt_attn_f16``;
``time_conv_out`` runs in fp32 fused with the layout change ``decode_latents`` needs (``pt_vae_time_conv_out``).
``dtype`` is fp16; ``.to(dtype=torch.float32)`` - what the reference's pipeline does around ``encode`` when the config says
``force_upcast`` (``pipeline...:453-462``) - switches ``encode`` to an fp32 path: fp32 activations and weights end to end on
``pt_conv2d_f32`` / ``pt_groupnorm_f32`` / ``pt_softmax_rows_f32`` (``csrc/vae_f32.hip``: fp32-input MFMA, exact fmaf chains;
fp64 GroupNorm statistics), 2e-6 ... 1e-5 from the fp32 result where the fp16 kernels sit at 1e-3.  ``decode`` stays fp16
(the reference decodes in fp16 too: it casts the VAE back first, ``:587-588``).
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Optional, Tuple

import torch

from . import blocks as B
from . import hip, ops, spec
from .modeling import BaseOutput, HipModel
from .packing import pack_conv2d, pack_linear, vec16

NORM_GROUPS = 32          # diffusers' Encoder / TemporalDecoder build every GroupNorm with 32 groups; AutoencoderKLTemporalDecoder's
                          # config has no field for it (the constructor demands block_out_channels % 32 == 0)


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
        # the same encoder in fp32 (force_upcast, 34 M parameters, 137 MB): built on the first fp32 encode() from the tensors kept here
        # (a VAE that only decodes, or one whose config says force_upcast = false, never pays for it)
        self._e32 = None
        self._e32_src = ({k: v for k, v in sd.items() if k.startswith("encoder.") or k.startswith("quant_conv.")}, device)
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

    def to(self, *a, **k):
        """``vae.to(dtype=torch.float32)`` / ``.to(dtype=torch.float16)`` (``pipeline...:455,461,588``): selects the arithmetic of
        ``encode`` - the fp32 path of ``csrc/vae_f32.hip`` or the fp16 MFMA kernels.  Device moves are not offered (``HipModel``)."""
        dt = k.get("dtype", next((x for x in a if isinstance(x, torch.dtype)), None))
        if dt is not None:
            if dt in (torch.float16, torch.float32):
                self.dtype = dt
            else:                                              # (HipModel.to ignores what it cannot do as well: weights stay packed fp16)
                warnings.warn(f"AutoencoderKLTemporalDecoder.to: dtype {dt} is not offered (fp16 kernels, fp32 encoder); left at {self.dtype}")
        return self

    def _build_e32(self):
        """The encoder's weights in fp32: convolution weights [Co, kh * kw * Ci] in (ky, kx, ci) order."""
        sd, device = self._e32_src
        cfg = self.config
        n, L = len(cfg.block_out_channels), cfg.layers_per_block
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        c32 = lambda k: (f32(sd[k + ".weight"].detach().float().permute(0, 2, 3, 1).reshape(sd[k + ".weight"].shape[0], -1)), f32(sd[k + ".bias"]),
                         tuple(sd[k + ".weight"].shape[2:]))
        l32 = lambda k: (f32(sd[k + ".weight"]), f32(sd[k + ".bias"]), (1, 1))
        n32 = lambda k: (f32(sd[k + ".weight"]), f32(sd[k + ".bias"]))

        def res32(p):
            d = dict(n1=n32(p + "norm1"), c1=c32(p + "conv1"), n2=n32(p + "norm2"), c2=c32(p + "conv2"), sc=None)
            if p + "conv_shortcut.weight" in sd:
                d["sc"] = c32(p + "conv_shortcut")
            return d
        return dict(conv_in=c32("encoder.conv_in"),
                    down=[([res32(f"encoder.down_blocks.{i}.resnets.{j}.") for j in range(L)],
                           c32(f"encoder.down_blocks.{i}.downsamplers.0.conv") if i != n - 1 else None) for i in range(n)],
                    mid=(res32("encoder.mid_block.resnets.0."), res32("encoder.mid_block.resnets.1.")),
                    attn=dict(gn=n32("encoder.mid_block.attentions.0.group_norm"), q=l32("encoder.mid_block.attentions.0.to_q"),
                              k=l32("encoder.mid_block.attentions.0.to_k"), v=l32("encoder.mid_block.attentions.0.to_v"),
                              o=l32("encoder.mid_block.attentions.0.to_out.0")),
                    norm_out=n32("encoder.conv_norm_out"), conv_out=c32("encoder.conv_out"), quant=c32("quant_conv"))

    # ---- fp32 encoder (force_upcast): channels-last fp32 tensors [N, H, W, C]
    def _conv32(self, x, wb, *, stride=1, pad=None, out_hw=None, res=None, scale=1.0):
        w, b, (kh, kw) = wb
        N, H, W, Ci = x.shape
        pad = (kh // 2) if pad is None else pad
        Ho, Wo = ((H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1) if out_hw is None else out_hw
        Co = w.shape[0]
        out = torch.empty((N, Ho, Wo, Co), dtype=torch.float32, device=x.device)
        p = hip.ConvF32Params()
        p.x, p.w, p.bias, p.res, p.out = x.data_ptr(), w.data_ptr(), (None if b is None else b.data_ptr()), (None if res is None else res.data_ptr()), out.data_ptr()
        p.Nimg, p.Hin, p.Win, p.Hout, p.Wout, p.Ci, p.Co = N, H, W, Ho, Wo, Ci, Co
        p.KH, p.KW, p.stride, p.pad_h, p.pad_w = kh, kw, stride, pad, pad
        p.ldx, p.ldw, p.ldo, p.ldr, p.scale = x.stride(2), w.stride(0), Co, (Co if res is not None else 0), float(scale)
        hip.check(hip.lib().pt_conv2d_f32(C.byref(p), ops._stream()), "pt_conv2d_f32")
        return out

    def _gn32(self, x, gb, silu, eps=1e-6):
        N, H, W, Cc = x.shape
        y = torch.empty_like(x)
        st = torch.empty(2 * N * NORM_GROUPS, dtype=torch.float64, device=x.device)
        hip.check(hip.lib().pt_groupnorm_f32(x.data_ptr(), H * W, N, Cc, NORM_GROUPS, eps, gb[0].data_ptr(), gb[1].data_ptr(), 1 if silu else 0,
                                             st.data_ptr(), y.data_ptr(), ops._stream()), "pt_groupnorm_f32")
        return y

    def _res32(self, x, r):
        h = self._conv32(self._gn32(x, r["n1"], True), r["c1"])
        sc = x if r["sc"] is None else self._conv32(x, r["sc"], pad=0)
        return self._conv32(self._gn32(h, r["n2"], True), r["c2"], res=sc)

    def _attn32(self, x, a):
        N, H, W, Cc = x.shape
        S = H * W
        y = self._gn32(x, a["gn"], False).view(N * S, 1, 1, Cc)
        q, k, v = (self._conv32(y, a[n]).view(N, S, Cc) for n in ("q", "k", "v"))
        outs = []
        for i in range(N):                                     # one frame at a time: S x S fp32 scores (340 MB at 576 x 1024)
            sc = self._conv32(q[i].view(S, 1, 1, Cc), (k[i], None, (1, 1))).view(S, S)
            hip.check(hip.lib().pt_softmax_rows_f32(sc.data_ptr(), S, S, S, Cc ** -0.5, ops._stream()), "pt_softmax_rows_f32")
            outs.append(self._conv32(sc.view(S, 1, 1, S), (v[i].t().contiguous(), None, (1, 1))).view(S, Cc))
        o = torch.stack(outs).view(N * S, 1, 1, Cc)
        return self._conv32(o, a["o"], res=x.reshape(N * S, 1, 1, Cc)).view(N, H, W, Cc)

    def _encode_f32(self, x: torch.Tensor) -> torch.Tensor:
        if self._e32 is None:
            self._e32 = self._build_e32()
        e = self._e32
        h = x.to(torch.float32).permute(0, 2, 3, 1).contiguous()                      # layout change only
        h = self._conv32(h, e["conv_in"])
        for res, down in e["down"]:
            for r in res:
                h = self._res32(h, r)
            if down is not None:                               # Downsample2D(padding=0): F.pad (0,1,0,1) + conv stride 2
                h = self._conv32(h, down, stride=2, pad=0, out_hw=(h.shape[1] // 2, h.shape[2] // 2))
        h = self._res32(self._attn32(self._res32(h, e["mid"][0]), e["attn"]), e["mid"][1])
        m = self._conv32(self._gn32(h, e["norm_out"], True), e["conv_out"])
        mom = self._conv32(m, e["quant"], pad=0)                                      # [N, h, w, 2z]
        return mom.permute(0, 3, 1, 2).contiguous()

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """``[N, 3, H, W]`` in [-1, 1] -> ``AutoencoderKLOutput(latent_dist=DiagonalGaussianDistribution)`` with fp32
        parameters ``[N, 8, H/8, W/8]``.  After ``.to(dtype=torch.float32)`` (``force_upcast``) every operation runs in fp32."""
        self._check(x, "x")
        N, Cin, H, W = x.shape
        cfg = self.config
        if Cin != cfg.in_channels or H % 2 ** (len(cfg.block_out_channels) - 1) or W % 2 ** (len(cfg.block_out_channels) - 1):
            raise ValueError(f"encode: {Cin} channels / {H} x {W} pixels do not fit the encoder ({cfg.in_channels} channels, "
                             f"{len(cfg.block_out_channels) - 1} halvings)")
        if self.dtype == torch.float32:
            post = DiagonalGaussianDistribution(self._encode_f32(x))
            return AutoencoderKLOutput(latent_dist=post) if return_dict else (post,)
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
