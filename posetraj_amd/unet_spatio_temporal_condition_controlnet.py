"""MI355X-native ``UNetSpatioTemporalConditionControlNetModel`` - drop-in for
``/root/reference/models/unet_spatio_temporal_condition_controlnet.py:32-504``.

Same constructor arguments, ``forward`` signature and error behaviour; the arithmetic runs in ``libposetraj_hip.so``.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from . import blocks as B
from . import ops, spec
from .modeling import DOWN_TYPES, UP_TYPES, BaseOutput, HipModel, _tup, check_block_args
from .packing import pack_conv2d, vec16


class UNetSpatioTemporalConditionOutput(BaseOutput):
    """``sample``: ``[batch, frames, channels, height, width]`` (``unet...:19-29``)."""


def _as_channels_last(r: torch.Tensor) -> torch.Tensor:
    """[N, C, h, w] residual -> contiguous channels-last fp16 (zero-copy for ControlNetSDVModel's outputs)."""
    if r.dim() != 4:
        raise ValueError(f"additional residuals must be [batch*frames, C, h, w]; got {tuple(r.shape)}")
    return ops.to_channels_last(r)


class UNetSpatioTemporalConditionControlNetModel(HipModel):
    _supports_gradient_checkpointing = False

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 8, out_channels: int = 4,
                 down_block_types: Tuple[str] = DOWN_TYPES, up_block_types: Tuple[str] = UP_TYPES,
                 block_out_channels: Tuple[int] = (320, 640, 1280, 1280), addition_time_embed_dim: int = 256,
                 projection_class_embeddings_input_dim: int = 768, layers_per_block: Union[int, Tuple[int]] = 2,
                 cross_attention_dim: Union[int, Tuple[int]] = 1024,
                 transformer_layers_per_block: Union[int, Tuple[int], Tuple[Tuple]] = 1,
                 num_attention_heads: Union[int, Tuple[int]] = (5, 10, 10, 20), num_frames: int = 25):
        check_block_args(down_block_types, up_block_types, block_out_channels, num_attention_heads, cross_attention_dim,
                         layers_per_block)
        super().__init__(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                         down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                         block_out_channels=tuple(block_out_channels), addition_time_embed_dim=addition_time_embed_dim,
                         projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
                         layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
                         transformer_layers_per_block=transformer_layers_per_block,
                         num_attention_heads=num_attention_heads, num_frames=num_frames)
        self.sample_size = sample_size
        # `_get_add_time_ids` of the pipeline reads unet.add_embedding.linear_1.in_features (pipeline...:48,210)
        lin1 = type("Linear1", (), {"in_features": projection_class_embeddings_input_dim})()
        self.add_embedding = type("AddEmbedding", (), {"linear_1": lin1})()

    def param_spec(self):
        return spec.unet_spec(self.config)

    def _pack(self, sd, device):
        cfg = self.config
        self._pack_encoder(sd, device)
        n = len(cfg.block_out_channels)
        rheads = _tup(cfg.num_attention_heads, n)[::-1]
        self.up_blocks = [B.UpBlock(sd, f"up_blocks.{i}.", typ == "CrossAttnUpBlockSpatioTemporal", rheads[i], device,
                                    self._temb_stack, self._xattn_stack)
                          for i, typ in enumerate(cfg.up_block_types)]
        self.conv_norm_out = (vec16(sd["conv_norm_out.weight"], device), vec16(sd["conv_norm_out.bias"], device))
        self.conv_out = pack_conv2d(sd["conv_out.weight"], sd["conv_out.bias"], device)
        self._finish_pack(device)

    def forward(self, sample: torch.FloatTensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor,
                down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
                mid_block_additional_residual: Optional[torch.Tensor] = None, return_dict: bool = True,
                added_time_ids: torch.Tensor = None) -> Union[UNetSpatioTemporalConditionOutput, Tuple]:
        """``unet...:356-504``."""
        if down_block_additional_residuals is None:
            # the reference zips the skips with None inside the down loop (:453-455) -> TypeError
            raise TypeError("zip argument #2 must support iteration (down_block_additional_residuals is mandatory)")
        if mid_block_additional_residual is None:
            raise TypeError("unsupported operand type(s) for +: 'Tensor' and 'NoneType' (mid_block_additional_residual is mandatory)")
        state = self._encode(sample, timestep, encoder_hidden_states, added_time_ids)
        return self._decode(state, down_block_additional_residuals, mid_block_additional_residual, return_dict)

    # The encoder half (conv_in, down blocks, mid block) never sees the ControlNet outputs - the reference adds them to
    # the collected skips and to the mid block's output (:451-469) - so it is independent of the ControlNet forward:
    # the pipeline runs the two concurrently on two HIP streams (pipeline...: networks()).
    def _encode(self, sample, timestep, encoder_hidden_states, added_time_ids, half=None):
        ctx, x, (Bc, F, h, w) = self._prologue(sample, timestep, encoder_hidden_states, added_time_ids, half=half)
        N = Bc * F
        x = ops.igemm(x, self.conv_in, geom=(N, h, w)).view(N, h, w, -1)
        skips = [x]
        counts = []                                          # number of skips collected after each down block
        for blk in self.down_blocks:
            x, t = blk.run(ctx, x)
            skips += t
            counts.append(len(skips))
        x = self.mid_block.run(ctx, x)
        return dict(ctx=ctx, x=x, skips=skips, counts=counts, dims=(Bc, F))

    @staticmethod
    def _multiplicity(state, n_res):
        """The add-loop sits inside the block loop and zip() stops at the shorter sequence (:451-459): every skip
        collected so far receives its residual again -> multiplicities (4,4,4,4,3,3,3,2,2,2,1,1)."""
        mult = [0] * len(state["skips"])
        for n_so_far in state["counts"]:
            for j in range(min(n_so_far, n_res)):
                mult[j] += 1
        return mult

    def _decode(self, state, down_block_additional_residuals, mid_block_additional_residual, return_dict=True,
                residuals_added: bool = False, out_f32: bool = False, out: Optional[torch.Tensor] = None):
        """Up path.  ``residuals_added``: the ControlNet residuals are already in ``state`` (accumulated there by
        ``ControlNetSDVModel._accumulate_into``); ``out_f32``: conv_out writes fp32 (the pipeline's guidance + Euler
        kernel reads it without an fp16 round trip).  Both are the pipeline's private fast path."""
        ctx, x, skips, (Bc, F) = state["ctx"], state["x"], state["skips"], state["dims"]
        if not residuals_added:
            residuals = list(down_block_additional_residuals)
            mult = self._multiplicity(state, len(residuals))
            # zip() truncation also drops skips beyond len(residuals) from the tuple the up path pops from
            skips = skips[:max(len(residuals), 0)] if len(residuals) < len(skips) else skips
            skips = [ops.axpy(s, _as_channels_last(r), float(m)).view(s.shape) if m else s
                     for s, r, m in zip(skips, residuals, mult)]
            x = ops.axpy(x, _as_channels_last(mid_block_additional_residual), 1.0).view(x.shape)      # :469
        for blk in self.up_blocks:                                                                    # :473-491
            k = len(blk.resnets)
            res, skips = skips[-k:], skips[:-k]
            x = blk.run(ctx, x, res)
        n, hh, ww, c = x.shape
        y = ops.groupnorm(x, *self.conv_norm_out, rows_per_sample=hh * ww, n_samples=n, eps=1e-5, silu=True)
        out = ops.igemm(y.view(n, hh, ww, c), self.conv_out, geom=(n, hh, ww), out_f32=out_f32, out=out)       # [M, out_channels]
        oc = self.config.out_channels
        sample_out = out.view(n, hh, ww, oc).permute(0, 3, 1, 2).reshape(Bc, F, oc, hh, ww)          # channels-last view
        if not return_dict:
            return (sample_out,)
        return UNetSpatioTemporalConditionOutput(sample=sample_out)
