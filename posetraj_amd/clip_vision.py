"""MI355X-native ``CLIPVisionModelWithProjection`` - the ``image_encoder`` of the reference pipeline
(``/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:22,125``; called at ``:157`` as
``self.image_encoder(image).image_embeds`` on the 224 x 224 resize of the conditioning image).  The class is
``transformers``'; constructor arguments are the fields of its ``CLIPVisionConfig`` (unknown fields of a ``config.json`` are
ignored), weights are its state dict (``<dir>/image_encoder/model[.fp16].safetensors``).

Execution: tokens ``[B * 257, C]`` fp16; the patch embedding (Conv2d with kernel = stride = patch) is ``pt_patchify_f16`` + one
``pt_igemm_f16`` whose epilogue adds the position embedding; every projection / MLP layer is ``pt_igemm_f16`` (fused QKV with
bias; residual adds in the epilogues, the residual stream kept as fp16 pairs like the U-Net's); LayerNorm ``pt_layernorm_f16``;
attention ``pt_attn_f16`` at head_dim 80 (ViT-H/14) over 257 tokens; the MLP activation ``pt_act_f16``.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops, spec
from .modeling import BaseOutput, HipModel
from .packing import pack_linear, vec16


class CLIPVisionModelOutput(BaseOutput):
    """``image_embeds`` ``[B, projection_dim]``, ``last_hidden_state`` ``[B, 257, C]``."""


class _Layer:
    pass


class CLIPVisionModelWithProjection(HipModel):
    weights_name = "model.safetensors"

    def __init__(self, hidden_size: int = 768, intermediate_size: int = 3072, projection_dim: int = 512,
                 num_hidden_layers: int = 12, num_attention_heads: int = 12, num_channels: int = 3, image_size: int = 224,
                 patch_size: int = 32, hidden_act: str = "quick_gelu", layer_norm_eps: float = 1e-5, **ignored):
        if hidden_size % num_attention_heads or hidden_size // num_attention_heads not in (64, 80, 128):
            raise ValueError(f"hidden_size {hidden_size} / {num_attention_heads} heads: pt_attn_f16 handles head_dim 64, 80 (ViT-H/14), 128")
        if hidden_act not in ("gelu", "quick_gelu"):
            raise ValueError(f"hidden_act {hidden_act!r} unsupported (gelu, quick_gelu)")
        if image_size % patch_size or hidden_size % 8 or intermediate_size % 8:
            raise ValueError("image_size must be a multiple of patch_size; widths multiples of 8")
        super().__init__(hidden_size=hidden_size, intermediate_size=intermediate_size, projection_dim=projection_dim,
                         num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads, num_channels=num_channels,
                         image_size=image_size, patch_size=patch_size, hidden_act=hidden_act, layer_norm_eps=layer_norm_eps)

    def param_spec(self):
        return spec.clip_vision_spec(self.config)

    def parameters(self):
        """``next(image_encoder.parameters()).dtype`` is how the reference asks for the dtype (``pipeline...:146``)."""
        yield torch.empty(0, dtype=self.dtype)

    def _pack(self, sd, device):
        cfg = self.config
        v = "vision_model."
        ln = lambda k: (vec16(sd[k + ".weight"], device), vec16(sd[k + ".bias"], device))
        lin = lambda k: pack_linear(sd[k + ".weight"], sd.get(k + ".bias"), device)
        w = sd[v + "embeddings.patch_embedding.weight"].detach().float()
        self.kpatch = (w[0].numel() + 7) // 8 * 8
        wp = torch.zeros((w.shape[0], self.kpatch), dtype=torch.float32, device=w.device)
        wp[:, :w[0].numel()] = w.reshape(w.shape[0], -1)
        self.patch = pack_linear(wp, None, device)
        pos = sd[v + "embeddings.position_embedding.weight"].detach().float()
        self.pos_patches = vec16(pos[1:], device)                                          # added in the patch GEMM's epilogue
        self.cls_row = vec16(sd[v + "embeddings.class_embedding"].detach().float() + pos[0], device)
        self.pre_ln, self.post_ln = ln(v + "pre_layrnorm"), ln(v + "post_layernorm")
        self.layers = []
        for i in range(cfg.num_hidden_layers):
            b = f"{v}encoder.layers.{i}."
            L = _Layer()
            a = b + "self_attn."
            L.qkv = pack_linear(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0),
                                torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0), device)
            L.o = lin(a + "out_proj")
            L.ln1, L.ln2 = ln(b + "layer_norm1"), ln(b + "layer_norm2")
            L.fc1, L.fc2 = lin(b + "mlp.fc1"), lin(b + "mlp.fc2")
            self.layers.append(L)
        self.proj = lin("visual_projection")

    def forward(self, pixel_values: Optional[torch.Tensor] = None, interpolate_pos_encoding: bool = False, **kw):
        if not self._loaded:
            raise RuntimeError("CLIPVisionModelWithProjection: no weights loaded (load_state_dict / from_pretrained / init_random_)")
        if interpolate_pos_encoding:
            raise NotImplementedError("interpolate_pos_encoding is not supported (the reference never passes it)")
        cfg = self.config
        x = pixel_values
        if not torch.is_tensor(x) or x.dim() != 4 or not x.is_cuda:
            raise RuntimeError("posetraj_amd: pixel_values must be a [B, 3, H, W] tensor on the ROCm device (no CPU path exists)")
        B, Cin, H, W = x.shape
        if H != cfg.image_size or W != cfg.image_size:
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        C, eps, heads = cfg.hidden_size, cfg.layer_norm_eps, cfg.num_attention_heads
        n_p = (cfg.image_size // cfg.patch_size) ** 2
        S = n_p + 1
        emb = torch.empty((B, S, C), dtype=torch.float16, device=x.device)
        patches = ops.patchify(x, cfg.patch_size, self.kpatch)                              # [B * n_p, kpatch]
        for b in range(B):                                                                  # rows 1 .. n_p of each image; row 0 = class token
            ops.igemm(patches[b * n_p:(b + 1) * n_p], self.patch, vec=self.pos_patches, vec_mode=1, vG=1, out=emb[b, 1:])
        emb[:, 0] = self.cls_row
        h = ops.layernorm(emb.view(B * S, C), *self.pre_ln, eps=eps)
        for L in self.layers:
            qkv = ops.igemm(ops.layernorm(h, *L.ln1, eps=eps), L.qkv)
            a = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, S, S, heads, C // heads)
            h = ops.igemm(a, L.o, res=h, wide=True)
            m = ops.activation(ops.igemm(ops.layernorm(h, *L.ln2, eps=eps), L.fc1), cfg.hidden_act)
            h = ops.igemm(m, L.fc2, res=h, wide=True)
        pooled = ops.layernorm(h.view(B, S, C)[:, 0].contiguous(), *self.post_ln, eps=eps)
        embeds = ops.igemm(pooled, self.proj)
        return CLIPVisionModelOutput(image_embeds=embeds, last_hidden_state=h.view(B, S, C))
