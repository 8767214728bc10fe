"""Oracle restatement of the two top-level networks of the hot path.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Wiring PINNED against the reference forwards run with
recording stand-in blocks (``tests/golden/wiring_*.npz``); block internals live in ``oracle/blocks.py``.

* ``models/controlnet_sdv.py:201-650``  (camera variant ``models/controlnet_sdv_cam_infer.py``: extra
  ``camera_cond`` forwarded to the condition encoder, ``:537,612``)
* ``models/unet_spatio_temporal_condition_controlnet.py:32-504``
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn

from . import blocks as B
from .quant import q
from .cond_embed import (ControlNetConditioningEmbeddingSVD, ControlNetConditioningEmbeddingSVD_CAM, zero_module)

DOWN_TYPES = ("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",)
UP_TYPES = ("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3


def _tuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _time_embed(model, sample, timestep, added_time_ids):
    """Shared prologue ``controlnet_sdv.py:551-590`` == ``unet...:387-426``."""
    t = timestep
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.float64 if isinstance(timestep, float) else torch.int64, device=sample.device)
    elif t.ndim == 0:
        t = t[None].to(sample.device)
    bsz, nf = sample.shape[:2]
    t = t.expand(bsz)
    emb = model.time_embedding(model.time_proj(t).to(sample.dtype))
    aug = model.add_time_proj(added_time_ids.flatten()).reshape(bsz, -1).to(emb.dtype)
    emb = q(emb + model.add_embedding(aug), True)
    return emb.repeat_interleave(nf, dim=0)


class _Encoder(nn.Module):
    """conv_in + time embeddings + 4 down blocks + mid: the part ControlNet shares with the U-Net."""

    def _build_encoder(self, in_channels, block_out_channels, addition_time_embed_dim,
                       projection_class_embeddings_input_dim, layers_per_block, cross_attention_dim,
                       transformer_layers_per_block, num_attention_heads, down_block_types):
        n = len(down_block_types)
        ch = tuple(block_out_channels)
        heads = _tuple(num_attention_heads, n)
        xdim = _tuple(cross_attention_dim, n)
        layers = _tuple(layers_per_block, n)
        tlayers = _tuple(transformer_layers_per_block, n)
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.time_proj = B.Timesteps(ch[0], True, 0)
        self.time_embedding = B.TimestepEmbedding(ch[0], temb)
        self.add_time_proj = B.Timesteps(addition_time_embed_dim, True, 0)
        self.add_embedding = B.TimestepEmbedding(projection_class_embeddings_input_dim, temb)
        self.down_blocks = nn.ModuleList()
        out_c = ch[0]
        for i, typ in enumerate(down_block_types):
            in_c, out_c = out_c, ch[i]
            self.down_blocks.append(B.get_down_block(
                typ, num_layers=layers[i], transformer_layers_per_block=tlayers[i], in_channels=in_c,
                out_channels=out_c, temb_channels=temb, add_downsample=i != n - 1, resnet_eps=1e-5,
                cross_attention_dim=xdim[i], num_attention_heads=heads[i], resnet_act_fn="silu"))
        self.mid_block = B.UNetMidBlockSpatioTemporal(ch[-1], temb, tlayers[-1], heads[-1], xdim[-1])
        return ch, heads, xdim, layers, tlayers, temb

    def _run_down(self, blk, sample, emb, ehs, ind):
        if getattr(blk, "has_cross_attention", False):
            return blk(hidden_states=sample, temb=emb, encoder_hidden_states=ehs, image_only_indicator=ind)
        return blk(hidden_states=sample, temb=emb, image_only_indicator=ind)


class ControlNetSDVModel(_Encoder):
    def __init__(self, sample_size=None, in_channels=8, out_channels=4, down_block_types=DOWN_TYPES,
                 up_block_types=UP_TYPES, block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
                 transformer_layers_per_block=1, num_attention_heads=(5, 10, 10, 20), num_frames=25,
                 conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256), camera=False):
        super().__init__()
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        ch, _, _, layers, _, _ = self._build_encoder(
            in_channels, block_out_channels, addition_time_embed_dim, projection_class_embeddings_input_dim,
            layers_per_block, cross_attention_dim, transformer_layers_per_block, num_attention_heads,
            down_block_types)
        cls = ControlNetConditioningEmbeddingSVD_CAM if camera else ControlNetConditioningEmbeddingSVD
        self.controlnet_cond_embedding = cls(ch[0], conditioning_channels, conditioning_embedding_out_channels)
        # 1 + sum(layers) + (n-1) zero-initialised 1x1 taps (:339-375) and one for mid (:378-382)
        self.controlnet_down_blocks = nn.ModuleList([zero_module(nn.Conv2d(ch[0], ch[0], 1))])
        for i, c in enumerate(ch):
            for _ in range(layers[i]):
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(c, c, 1)))
            if i != len(ch) - 1:
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(c, c, 1)))
        self.controlnet_mid_block = zero_module(nn.Conv2d(ch[-1], ch[-1], 1))

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, controlnet_cond=None,
                image_only_indicator=None, return_dict=True, guess_mode=False, conditioning_scale=1.0,
                camera_cond=None):
        bsz, nf = sample.shape[:2]
        emb = _time_embed(self, sample, timestep, added_time_ids)
        sample = q(sample.flatten(0, 1), True)
        ehs = q(encoder_hidden_states, True).repeat_interleave(nf, dim=0)
        sample = q(self.conv_in(sample), controlnet_cond is None)
        if controlnet_cond is not None:                                                        # :596-599
            if camera_cond is not None or isinstance(self.controlnet_cond_embedding,
                                                     ControlNetConditioningEmbeddingSVD_CAM):
                sample = q(sample + self.controlnet_cond_embedding(controlnet_cond, camera_cond), True)
            else:
                sample = q(sample + self.controlnet_cond_embedding(controlnet_cond), True)
        ind = torch.zeros(bsz, nf, dtype=sample.dtype, device=sample.device)                   # :602 (Q6)
        taps = (sample,)
        for blk in self.down_blocks:
            sample, res = self._run_down(blk, sample, emb, ehs, ind)
            taps += res
        sample = self.mid_block(hidden_states=sample, temb=emb, encoder_hidden_states=ehs, image_only_indicator=ind)
        down = [q(q(conv(q(t, True))) * conditioning_scale, True) for t, conv in zip(taps, self.controlnet_down_blocks)]   # :630-642
        mid = q(q(self.controlnet_mid_block(q(sample, True))) * conditioning_scale, True)
        if not return_dict:
            return (down, mid)
        return SimpleNamespace(down_block_res_samples=down, mid_block_res_sample=mid)

    @classmethod
    def from_unet(cls, unet, conditioning_embedding_out_channels=(16, 32, 96, 256), load_weights_from_unet=True,
                  conditioning_channels=3, camera=False):
        """``controlnet_sdv.py:653-709``: copies conv_in / time_embedding / down / mid, NOT add_embedding."""
        c = unet.config
        net = cls(in_channels=c.in_channels, down_block_types=c.down_block_types,
                  block_out_channels=c.block_out_channels, addition_time_embed_dim=c.addition_time_embed_dim,
                  transformer_layers_per_block=c.transformer_layers_per_block,
                  cross_attention_dim=c.cross_attention_dim, num_attention_heads=c.num_attention_heads,
                  num_frames=c.num_frames, sample_size=c.sample_size, layers_per_block=c.layers_per_block,
                  projection_class_embeddings_input_dim=c.projection_class_embeddings_input_dim,
                  conditioning_channels=conditioning_channels,
                  conditioning_embedding_out_channels=conditioning_embedding_out_channels, camera=camera)
        if load_weights_from_unet:
            net.conv_in.load_state_dict(unet.conv_in.state_dict())
            net.time_embedding.load_state_dict(unet.time_embedding.state_dict())
            net.down_blocks.load_state_dict(unet.down_blocks.state_dict())
            net.mid_block.load_state_dict(unet.mid_block.state_dict())
        return net


class UNetSpatioTemporalConditionControlNetModel(_Encoder):
    def __init__(self, sample_size=None, in_channels=8, out_channels=4, down_block_types=DOWN_TYPES,
                 up_block_types=UP_TYPES, block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
                 transformer_layers_per_block=1, num_attention_heads=(5, 10, 10, 20), num_frames=25):
        super().__init__()
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("self", "__class__")})
        ch, heads, xdim, layers, tlayers, temb = self._build_encoder(
            in_channels, block_out_channels, addition_time_embed_dim, projection_class_embeddings_input_dim,
            layers_per_block, cross_attention_dim, transformer_layers_per_block, num_attention_heads,
            down_block_types)
        n = len(ch)
        rch, rheads, rlayers = ch[::-1], heads[::-1], layers[::-1]
        rxdim, rtl = xdim[::-1], tlayers[::-1]
        self.up_blocks = nn.ModuleList()
        out_c = rch[0]
        for i, typ in enumerate(up_block_types):                                               # :203-234
            prev, out_c = out_c, rch[i]
            in_c = rch[min(i + 1, n - 1)]
            self.up_blocks.append(B.get_up_block(
                typ, num_layers=rlayers[i] + 1, transformer_layers_per_block=rtl[i], in_channels=in_c,
                out_channels=out_c, prev_output_channel=prev, temb_channels=temb, add_upsample=i != n - 1,
                resnet_eps=1e-5, resolution_idx=i, cross_attention_dim=rxdim[i], num_attention_heads=rheads[i],
                resnet_act_fn="silu"))
        self.conv_norm_out = nn.GroupNorm(num_channels=ch[0], num_groups=32, eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict=True, added_time_ids=None):
        bsz, nf = sample.shape[:2]
        emb = _time_embed(self, sample, timestep, added_time_ids)
        sample = q(sample.flatten(0, 1), True)
        ehs = q(encoder_hidden_states, True).repeat_interleave(nf, dim=0)
        sample = q(self.conv_in(sample), True)
        ind = torch.zeros(bsz, nf, dtype=sample.dtype, device=sample.device)
        skips = (sample,)
        for blk in self.down_blocks:
            sample, res = self._run_down(blk, sample, emb, ehs, ind)
            skips += res
            # :451-459 - the add sits INSIDE the block loop, and zip() truncates to the skips collected so
            # far, so earlier skips receive their residual again after every later block;
            # zip(..., None) raises TypeError when no residuals are given (Q2).
            # (each repeated add rounds in the fp16 reference; the MI355X path adds multiplicity x residual once)
            skips = tuple(q(s + r) for s, r in zip(skips, down_block_additional_residuals))
        skips = tuple(q(s, True) for s in skips)
        sample = self.mid_block(hidden_states=sample, temb=emb, encoder_hidden_states=ehs, image_only_indicator=ind)
        sample = q(sample + mid_block_additional_residual, True)
        for blk in self.up_blocks:                                                             # :473-491
            k = len(blk.resnets)
            res, skips = skips[-k:], skips[:-k]
            if getattr(blk, "has_cross_attention", False):
                sample = blk(hidden_states=sample, temb=emb, res_hidden_states_tuple=res,
                             encoder_hidden_states=ehs, image_only_indicator=ind)
            else:
                sample = blk(hidden_states=sample, temb=emb, res_hidden_states_tuple=res, image_only_indicator=ind)
        sample = q(self.conv_out(q(self.conv_act(q(self.conv_norm_out(q(sample, True)))), True)))   # fp32 store on the MI355X path
        sample = sample.reshape(bsz, nf, *sample.shape[1:])
        if not return_dict:
            return (sample,)
        return SimpleNamespace(sample=sample)


# ---------------------------------------------------------------------------------------------- configs
def tiny_config(**over):
    """BASELINE configs[0]: tiny random-init nets.  head_dim is 64 at every level (like SVD)."""
    cfg = dict(block_out_channels=(64, 128, 256, 256), num_attention_heads=(1, 2, 4, 4), cross_attention_dim=64,
               addition_time_embed_dim=32, projection_class_embeddings_input_dim=96, layers_per_block=2,
               num_frames=14)
    cfg.update(over)
    return cfg


def svd_config(**over):
    """SVD-img2vid ``unet/config.json`` values."""
    cfg = dict(block_out_channels=(320, 640, 1280, 1280), num_attention_heads=(5, 10, 20, 20),
               cross_attention_dim=1024, addition_time_embed_dim=256, projection_class_embeddings_input_dim=768,
               layers_per_block=2, num_frames=14)
    cfg.update(over)
    return cfg


def randomize_zero_convs(controlnet: ControlNetSDVModel, std: float = 0.02, seed: int = 7):
    """Zero-initialised output convs make a fresh ControlNet a no-op; parity tests re-randomise them."""
    g = torch.Generator().manual_seed(seed)
    mods = list(controlnet.controlnet_down_blocks) + [controlnet.controlnet_mid_block,
                                                      controlnet.controlnet_cond_embedding.conv_out]
    with torch.no_grad():
        for m in mods:
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * std)
