"""MI355X-native ``ControlNetSDVModel`` - drop-in for ``/root/reference/models/controlnet_sdv.py:201-650`` and, with
``camera=True`` (or via ``posetraj_amd.controlnet_sdv_cam_infer``), for ``models/controlnet_sdv_cam_infer.py``.

Same constructor arguments, ``forward`` signature and return convention as the reference; the arithmetic runs in
``libposetraj_hip.so``.  Outputs are ``[B*F, C, h, w]``-shaped fp16 tensors that are channels-last in memory (zero-copy
views of the kernels' layout); ``UNetSpatioTemporalConditionControlNetModel`` consumes them without a copy.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from . import ops, spec
from .modeling import DOWN_TYPES, UP_TYPES, BaseOutput, HipModel, _tup, check_block_args
from .packing import pack_conv2d, pack_linear


class ControlNetOutput(BaseOutput):
    """``down_block_res_samples`` (12 tensors) and ``mid_block_res_sample`` (``controlnet_sdv.py:41-58``)."""


class ControlNetConditioningEmbeddingSVD:
    """Condition encoder (``controlnet_sdv.py:61-116``; camera twin ``controlnet_sdv_cam_infer.py:61-130``):
    8 convolutions with SiLU fused into each producer's epilogue; the last (zero-initialised) one has none."""

    def __init__(self, sd, p, device, camera: bool):
        conv = lambda k, **kw: pack_conv2d(sd[p + k + ".weight"], sd[p + k + ".bias"], device, **kw)
        self.conv_in = conv("conv_in")
        self.conv_in.silu = True
        self.blocks = []
        i = 0
        while f"{p}blocks.{i}.weight" in sd:
            b = conv(f"blocks.{i}", stride=2 if i % 2 else 1)
            b.silu = True
            self.blocks.append(b)
            i += 1
        self.cc = None
        if camera:
            w, b = sd[p + "cc_projection.weight"], sd[p + "cc_projection.bias"]
            kpad = (w.shape[1] + 7) // 8 * 8
            wp = torch.zeros((w.shape[0], kpad), dtype=torch.float32, device=w.device)
            wp[:, :w.shape[1]] = w.float()
            self.cc = pack_linear(wp, b, device)
        self.conv_out = conv("conv_out")

    def run(self, cond: torch.Tensor, camera_RT: Optional[torch.Tensor], res: Optional[torch.Tensor],
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """cond ``[B, F, 3, H, W]`` -> ``[N, H/8, W/8, C]`` (+ ``res`` if given, fused into the last epilogue)."""
        Bc, F, Cc, H, W = cond.shape
        N = Bc * F
        x = ops.to_channels_last(cond.reshape(N, Cc, H, W), cpad=self.conv_in.cin)
        x = ops.igemm(x, self.conv_in, geom=(N, H, W)).view(N, H, W, -1)
        for b in self.blocks:
            n, hh, ww, _ = x.shape
            y = ops.igemm(x, b, geom=(n, hh, ww))
            if b.stride == 2:
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
            x = y.view(n, hh, ww, -1)
        n, hh, ww, c = x.shape
        if self.cc is not None and camera_RT is not None:
            cam = camera_RT.to(device=x.device, dtype=torch.float16).reshape(N, -1).contiguous()
            if cam.shape[1] != 12:
                raise ValueError(f"camera_cond must have 12 values per frame (R|T); got {cam.shape[1]}")
            xc = ops.concat_camera(x, cam, self.cc.cin)
            x = ops.igemm(xc.view(n * hh * ww, -1), self.cc).view(n, hh, ww, c)
        if out is not None and tuple(out.shape) != (n, hh, ww, self.conv_out.N):
            out = None
        o2 = None if out is None else out.view(n * hh * ww, self.conv_out.N)
        return ops.igemm(x, self.conv_out, geom=(n, hh, ww), res=res, out=o2).view(n, hh, ww, -1)


class ControlNetSDVModel(HipModel):
    _supports_gradient_checkpointing = False

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 8, out_channels: int = 4,
                 down_block_types: Tuple[str] = DOWN_TYPES, up_block_types: Tuple[str] = UP_TYPES,
                 block_out_channels: Tuple[int] = (320, 640, 1280, 1280), addition_time_embed_dim: int = 256,
                 projection_class_embeddings_input_dim: int = 768, layers_per_block: Union[int, Tuple[int]] = 2,
                 cross_attention_dim: Union[int, Tuple[int]] = 1024,
                 transformer_layers_per_block: Union[int, Tuple[int], Tuple[Tuple]] = 1,
                 num_attention_heads: Union[int, Tuple[int]] = (5, 10, 10, 20), num_frames: int = 25,
                 conditioning_channels: int = 3, conditioning_embedding_out_channels: Optional[Tuple[int, ...]] = (16, 32, 96, 256),
                 camera: bool = False):
        check_block_args(down_block_types, up_block_types, block_out_channels, num_attention_heads, cross_attention_dim,
                         layers_per_block)
        super().__init__(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                         down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                         block_out_channels=tuple(block_out_channels), addition_time_embed_dim=addition_time_embed_dim,
                         projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
                         layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
                         transformer_layers_per_block=transformer_layers_per_block,
                         num_attention_heads=num_attention_heads, num_frames=num_frames,
                         conditioning_channels=conditioning_channels,
                         conditioning_embedding_out_channels=tuple(conditioning_embedding_out_channels), camera=camera)
        self.sample_size = sample_size
        self._cond_cache = None

    def param_spec(self):
        return spec.controlnet_spec(self.config, camera=self.config.camera)

    def _pack(self, sd, device):
        self._pack_encoder(sd, device)
        self.controlnet_cond_embedding = ControlNetConditioningEmbeddingSVD(sd, "controlnet_cond_embedding.", device,
                                                                            self.config.camera)
        self.controlnet_down_blocks = []
        k = 0
        while f"controlnet_down_blocks.{k}.weight" in sd:
            self.controlnet_down_blocks.append(pack_conv2d(sd[f"controlnet_down_blocks.{k}.weight"],
                                                           sd[f"controlnet_down_blocks.{k}.bias"], device, padding=0))
            k += 1
        self.controlnet_mid_block = pack_conv2d(sd["controlnet_mid_block.weight"], sd["controlnet_mid_block.bias"], device,
                                                padding=0)
        self._finish_pack(device)
        self._cond_cache = None

    # the condition encoder depends on neither the timestep nor the latents: its output is reused while
    # the caller keeps passing the same, unmodified tensors
    def _cond_embedding(self, controlnet_cond, camera_cond, out=None):
        """``out``: a buffer the caller owns and wants the embedding in (the pipeline's hipGraph state: the captured
        graph reads that address, so it must not depend on which buffer this cache happens to hold)."""
        key = tuple((t.data_ptr(), tuple(t.shape), t._version, t.dtype) if t is not None else None
                    for t in (controlnet_cond, camera_cond))
        hit = self._cond_cache is not None and self._cond_cache[0] == key
        if hit and out is not None and self._cond_cache[1].data_ptr() != out.data_ptr():
            hit = False
        if not hit:
            e = self.controlnet_cond_embedding.run(controlnet_cond, camera_cond, None, out=out)
            self._cond_cache = (key, e, controlnet_cond, camera_cond)        # keep inputs alive: data_ptr stays unique
        return self._cond_cache[1]

    def forward(self, sample: torch.FloatTensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor, added_time_ids: torch.Tensor,
                controlnet_cond: torch.FloatTensor = None, image_only_indicator: Optional[torch.Tensor] = None,
                return_dict: bool = True, guess_mode: bool = False, conditioning_scale: float = 1.0,
                camera_cond: Optional[torch.Tensor] = None) -> Union[ControlNetOutput, Tuple]:
        """``controlnet_sdv.py:516-650``.  ``image_only_indicator`` is accepted and ignored exactly like the reference
        (it is overwritten with zeros at ``:602``); ``guess_mode`` is unused there too."""
        taps, x = self._features(sample, timestep, encoder_hidden_states, added_time_ids, controlnet_cond, camera_cond)
        # zero-convs, then * conditioning_scale (:630-643) - one epilogue each
        outs = []
        for t, conv in zip(taps, self.controlnet_down_blocks):
            n, hh, ww, c = t.shape
            outs.append(ops.igemm(t, conv, geom=(n, hh, ww), out_scale=conditioning_scale).view(n, hh, ww, c).permute(0, 3, 1, 2))
        n, hh, ww, c = x.shape
        mid = ops.igemm(x, self.controlnet_mid_block, geom=(n, hh, ww), out_scale=conditioning_scale)
        mid = mid.view(n, hh, ww, c).permute(0, 3, 1, 2)
        if not return_dict:
            return (outs, mid)
        return ControlNetOutput(down_block_res_samples=outs, mid_block_res_sample=mid)

    def _features(self, sample, timestep, encoder_hidden_states, added_time_ids, controlnet_cond=None, camera_cond=None,
                  half=None):
        """conv_in (+ condition embedding), down blocks, mid block (``:551-628``): the 12 taps and the mid feature,
        channels-last, before the zero-convs."""
        ctx, x, (Bc, F, h, w) = self._prologue(sample, timestep, encoder_hidden_states, added_time_ids, half=half)
        N = Bc * F
        res = None
        if controlnet_cond is not None:                                           # :596-599
            nb = Bc if half is None else 2 * Bc                                   # (a half forward is handed the whole batch's maps)
            if tuple(controlnet_cond.shape[:2]) != (nb, F):
                raise ValueError(f"controlnet_cond must be [batch, frames, C, H, W] matching sample; got {tuple(controlnet_cond.shape)}")
            res = self._cond_embedding(controlnet_cond, camera_cond if self.config.camera else None).reshape(nb * F * h * w, -1)
            if half is not None:
                res = res[half * N * h * w:(half + 1) * N * h * w]
        x = ops.igemm(x, self.conv_in, geom=(N, h, w), res=res).view(N, h, w, -1)
        taps = [x]
        for blk in self.down_blocks:
            x, t = blk.run(ctx, x)
            taps += t
        x = self.mid_block.run(ctx, x)
        return taps, x

    def _accumulate_into(self, taps, x_mid, conditioning_scale, skips, mult, unet_mid):
        """The zero-convs with the U-Net's residual adds as their epilogue (``controlnet_sdv.py:630-643`` +
        ``unet...:451-459,469``):  skip_j += mult_j * scale * zero_conv_j(tap_j)  and  mid += scale * zero_conv(mid),
        IN PLACE on the U-Net's own skip / mid tensors, one rounding per element.  Used by the pipeline only - the
        public ``forward`` still returns the residuals like the reference."""
        if len(skips) != len(taps) or len(mult) != len(taps):
            raise ValueError(f"{len(taps)} ControlNet taps against {len(skips)} U-Net skips")
        for t, conv, s, m in zip(taps, self.controlnet_down_blocks, skips, mult):
            n, hh, ww, c = t.shape
            if tuple(s.shape) != (n, hh, ww, c) or not s.is_contiguous():
                raise ValueError(f"skip {tuple(s.shape)} does not match the ControlNet tap {tuple(t.shape)}")
            if m:
                s2 = ops.wview(s, n * hh * ww, c)           # a wide skip is summed as its pair, the result is plain fp16
                ops.igemm(t, conv, geom=(n, hh, ww), out_scale=float(m) * conditioning_scale, res=s2, res_post=True, out=s2)
                ops.drop_lo(s)
        n, hh, ww, c = x_mid.shape
        m2 = ops.wview(unet_mid, n * hh * ww, c)
        ops.igemm(x_mid, self.controlnet_mid_block, geom=(n, hh, ww), out_scale=conditioning_scale, res=m2, res_post=True, out=m2)
        ops.drop_lo(unet_mid)

    @classmethod
    def from_unet(cls, unet, controlnet_conditioning_channel_order: str = "rgb",
                  conditioning_embedding_out_channels: Optional[Tuple[int, ...]] = (16, 32, 96, 256),
                  load_weights_from_unet: bool = True, conditioning_channels: int = 3, camera: bool = False, seed: int = 0):
        """``controlnet_sdv.py:653-709``: same config as the U-Net; copies conv_in / time_embedding / down_blocks /
        mid_block (NOT add_embedding, ``:698-707``).  Needs the U-Net loaded with ``keep_source=True``.  The remaining
        parameters get their reference initialisation: zeros for the ControlNet output convs, PyTorch defaults else."""
        c = unet.config
        net = cls(in_channels=c.in_channels, down_block_types=c.down_block_types, block_out_channels=c.block_out_channels,
                  addition_time_embed_dim=c.addition_time_embed_dim, transformer_layers_per_block=c.transformer_layers_per_block,
                  cross_attention_dim=c.cross_attention_dim, num_attention_heads=c.num_attention_heads,
                  num_frames=c.num_frames, sample_size=c.sample_size, layers_per_block=c.layers_per_block,
                  projection_class_embeddings_input_dim=c.projection_class_embeddings_input_dim,
                  conditioning_channels=conditioning_channels,
                  conditioning_embedding_out_channels=conditioning_embedding_out_channels, camera=camera)
        src = unet.state_dict() if load_weights_from_unet else {}
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shape in net.param_spec().items():
            copied = k.startswith(("conv_in.", "time_embedding.", "down_blocks.", "mid_block."))
            if copied and k in src:
                sd[k] = src[k]
            elif k.startswith(("controlnet_down_blocks", "controlnet_mid_block", "controlnet_cond_embedding.conv_out")):
                sd[k] = torch.zeros(shape)
            elif "norm" in k.split(".")[-2]:
                sd[k] = torch.ones(shape) if k.endswith("weight") else torch.zeros(shape)
            elif k.endswith("mix_factor"):
                sd[k] = torch.full(shape, 0.5)
            else:
                fan_in = 1
                for v in (shape[1:] if len(shape) > 1 else shape):
                    fan_in *= v
                sd[k] = (torch.rand(shape, generator=g) * 2 - 1) / fan_in ** 0.5
        return net.load_state_dict(sd, unet.device, keep_source=True)
