"""Parameter inventory (state-dict key -> shape) of the two networks of the hot path, derived from the same
constructor arguments the reference classes take (``models/controlnet_sdv.py:238-391``,
``models/unet_spatio_temporal_condition_controlnet.py:69-245``; diffusers 0.24.0 key names).
Used to validate checkpoints on load and to create random-init weights of the right architecture for benchmarks."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

Shape = Tuple[int, ...]


def _lin(d, name, cin, cout, bias=True):
    d[name + ".weight"] = (cout, cin)
    if bias:
        d[name + ".bias"] = (cout,)


def _norm(d, name, c):
    d[name + ".weight"] = (c,)
    d[name + ".bias"] = (c,)


def _conv(d, name, cin, cout, k):
    d[name + ".weight"] = (cout, cin, k, k)
    d[name + ".bias"] = (cout,)


def _resblock(d, p, cin, cout, temb):
    s, t = p + "spatial_res_block.", p + "temporal_res_block."
    _norm(d, s + "norm1", cin); _conv(d, s + "conv1", cin, cout, 3); _lin(d, s + "time_emb_proj", temb, cout)
    _norm(d, s + "norm2", cout); _conv(d, s + "conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, s + "conv_shortcut", cin, cout, 1)
    _norm(d, t + "norm1", cout)
    d[t + "conv1.weight"] = (cout, cout, 3, 1, 1); d[t + "conv1.bias"] = (cout,)
    _lin(d, t + "time_emb_proj", temb, cout)
    _norm(d, t + "norm2", cout)
    d[t + "conv2.weight"] = (cout, cout, 3, 1, 1); d[t + "conv2.bias"] = (cout,)
    d[p + "time_mixer.mix_factor"] = (1,)


def _attention(d, p, c, kv):
    _lin(d, p + "to_q", c, c, False); _lin(d, p + "to_k", kv, c, False); _lin(d, p + "to_v", kv, c, False)
    _lin(d, p + "to_out.0", c, c)


def _ff(d, p, c):
    _lin(d, p + "net.0.proj", c, 8 * c); _lin(d, p + "net.2", 4 * c, c)


def _transformer(d, p, c, xdim, layers):
    _norm(d, p + "norm", c); _lin(d, p + "proj_in", c, c)
    for i in range(layers):
        b = f"{p}transformer_blocks.{i}."
        _norm(d, b + "norm1", c); _attention(d, b + "attn1.", c, c)
        _norm(d, b + "norm2", c); _attention(d, b + "attn2.", c, xdim)
        _norm(d, b + "norm3", c); _ff(d, b + "ff.", c)
    for i in range(layers):
        b = f"{p}temporal_transformer_blocks.{i}."
        _norm(d, b + "norm_in", c); _ff(d, b + "ff_in.", c)
        _norm(d, b + "norm1", c); _attention(d, b + "attn1.", c, c)
        _norm(d, b + "norm2", c); _attention(d, b + "attn2.", c, xdim)
        _norm(d, b + "norm3", c); _ff(d, b + "ff.", c)
    _lin(d, p + "time_pos_embed.linear_1", c, 4 * c); _lin(d, p + "time_pos_embed.linear_2", 4 * c, c)
    d[p + "time_mixer.mix_factor"] = (1,)
    _lin(d, p + "proj_out", c, c)


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def encoder_spec(cfg) -> "OrderedDict[str, Shape]":
    d: "OrderedDict[str, Shape]" = OrderedDict()
    ch = tuple(cfg["block_out_channels"]); n = len(ch)
    temb = ch[0] * 4
    layers = _tup(cfg["layers_per_block"], n); tl = _tup(cfg["transformer_layers_per_block"], n)
    xdim = _tup(cfg["cross_attention_dim"], n)
    _conv(d, "conv_in", cfg["in_channels"], ch[0], 3)
    _lin(d, "time_embedding.linear_1", ch[0], temb); _lin(d, "time_embedding.linear_2", temb, temb)
    _lin(d, "add_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], temb)
    _lin(d, "add_embedding.linear_2", temb, temb)
    out_c = ch[0]
    for i, typ in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, ch[i]
        for j in range(layers[i]):
            _resblock(d, f"down_blocks.{i}.resnets.{j}.", in_c if j == 0 else out_c, out_c, temb)
            if typ == "CrossAttnDownBlockSpatioTemporal":
                _transformer(d, f"down_blocks.{i}.attentions.{j}.", out_c, xdim[i], tl[i])
        if i != n - 1:
            _conv(d, f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    _resblock(d, "mid_block.resnets.0.", ch[-1], ch[-1], temb)
    _transformer(d, "mid_block.attentions.0.", ch[-1], xdim[-1], tl[-1])
    _resblock(d, "mid_block.resnets.1.", ch[-1], ch[-1], temb)
    return d


def unet_spec(cfg) -> "OrderedDict[str, Shape]":
    d = encoder_spec(cfg)
    ch = tuple(cfg["block_out_channels"]); n = len(ch)
    temb = ch[0] * 4
    rch = ch[::-1]
    layers = _tup(cfg["layers_per_block"], n)[::-1]; tl = _tup(cfg["transformer_layers_per_block"], n)[::-1]
    xdim = _tup(cfg["cross_attention_dim"], n)[::-1]
    out_c = rch[0]
    for i, typ in enumerate(cfg["up_block_types"]):
        prev, out_c = out_c, rch[i]
        in_c = rch[min(i + 1, n - 1)]
        nl = layers[i] + 1
        for j in range(nl):
            skip_c = in_c if j == nl - 1 else out_c
            res_in = prev if j == 0 else out_c
            _resblock(d, f"up_blocks.{i}.resnets.{j}.", res_in + skip_c, out_c, temb)
            if typ == "CrossAttnUpBlockSpatioTemporal":
                _transformer(d, f"up_blocks.{i}.attentions.{j}.", out_c, xdim[i], tl[i])
        if i != n - 1:
            _conv(d, f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    _norm(d, "conv_norm_out", ch[0])
    _conv(d, "conv_out", ch[0], cfg["out_channels"], 3)
    return d


def controlnet_spec(cfg, camera: bool = False) -> "OrderedDict[str, Shape]":
    d = encoder_spec(cfg)
    ch = tuple(cfg["block_out_channels"]); n = len(ch)
    layers = _tup(cfg["layers_per_block"], n)
    ce = tuple(cfg["conditioning_embedding_out_channels"])
    p = "controlnet_cond_embedding."
    _conv(d, p + "conv_in", cfg["conditioning_channels"], ce[0], 3)
    for i in range(len(ce) - 1):
        _conv(d, f"{p}blocks.{2 * i}", ce[i], ce[i], 3)
        _conv(d, f"{p}blocks.{2 * i + 1}", ce[i], ce[i + 1], 3)
    if camera:
        _lin(d, p + "cc_projection", ce[-1] + 12, ce[-1])
    _conv(d, p + "conv_out", ce[-1], ch[0], 3)
    k = 0
    _conv(d, f"controlnet_down_blocks.{k}", ch[0], ch[0], 1); k += 1
    for i, c in enumerate(ch):
        for _ in range(layers[i]):
            _conv(d, f"controlnet_down_blocks.{k}", c, c, 1); k += 1
        if i != n - 1:
            _conv(d, f"controlnet_down_blocks.{k}", c, c, 1); k += 1
    _conv(d, "controlnet_mid_block", ch[-1], ch[-1], 1)
    return d


def _plain_resnet(d, p, cin, cout):
    _norm(d, p + "norm1", cin); _conv(d, p + "conv1", cin, cout, 3)
    _norm(d, p + "norm2", cout); _conv(d, p + "conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, p + "conv_shortcut", cin, cout, 1)


def _vae_resblock(d, p, cin, cout):
    s, t = p + "spatial_res_block.", p + "temporal_res_block."
    _plain_resnet(d, s, cin, cout)
    _norm(d, t + "norm1", cout)
    d[t + "conv1.weight"] = (cout, cout, 3, 1, 1); d[t + "conv1.bias"] = (cout,)
    _norm(d, t + "norm2", cout)
    d[t + "conv2.weight"] = (cout, cout, 3, 1, 1); d[t + "conv2.bias"] = (cout,)
    d[p + "time_mixer.mix_factor"] = (1,)


def _vae_attention(d, p, c):
    _norm(d, p + "group_norm", c)
    for k in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(d, p + k, c, c)


def vae_spec(cfg) -> "OrderedDict[str, Shape]":
    """``AutoencoderKLTemporalDecoder`` of diffusers 0.24.0 (the ``vae`` of ``pipeline/pipeline_stable_video_diffusion_controlnet.py:124``):
    ``Encoder`` (DownEncoderBlock2D x n, UNetMidBlock2D), ``quant_conv``, ``TemporalDecoder`` [UNVERIFIED-MEMORY key names]."""
    d: "OrderedDict[str, Shape]" = OrderedDict()
    ch = tuple(cfg["block_out_channels"]); n = len(ch)
    L, z = cfg["layers_per_block"], cfg["latent_channels"]
    _conv(d, "encoder.conv_in", cfg["in_channels"], ch[0], 3)
    out_c = ch[0]
    for i in range(n):
        in_c, out_c = out_c, ch[i]
        for j in range(L):
            _plain_resnet(d, f"encoder.down_blocks.{i}.resnets.{j}.", in_c if j == 0 else out_c, out_c)
        if i != n - 1:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    _plain_resnet(d, "encoder.mid_block.resnets.0.", ch[-1], ch[-1])
    _vae_attention(d, "encoder.mid_block.attentions.0.", ch[-1])
    _plain_resnet(d, "encoder.mid_block.resnets.1.", ch[-1], ch[-1])
    _norm(d, "encoder.conv_norm_out", ch[-1])
    _conv(d, "encoder.conv_out", ch[-1], 2 * z, 3)
    _conv(d, "decoder.conv_in", z, ch[-1], 3)
    for j in range(L):
        _vae_resblock(d, f"decoder.mid_block.resnets.{j}.", ch[-1], ch[-1])
    _vae_attention(d, "decoder.mid_block.attentions.0.", ch[-1])
    rch = ch[::-1]
    out_c = rch[0]
    for i in range(n):
        prev, out_c = out_c, rch[i]
        for j in range(L + 1):
            _vae_resblock(d, f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else out_c, out_c)
        if i != n - 1:
            _conv(d, f"decoder.up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    _norm(d, "decoder.conv_norm_out", ch[0])
    _conv(d, "decoder.conv_out", ch[0], cfg["out_channels"], 3)
    d["decoder.time_conv_out.weight"] = (cfg["out_channels"], cfg["out_channels"], 3, 1, 1)
    d["decoder.time_conv_out.bias"] = (cfg["out_channels"],)
    _conv(d, "quant_conv", 2 * z, 2 * z, 1)
    return d


def clip_vision_spec(cfg) -> "OrderedDict[str, Shape]":
    """``transformers.CLIPVisionModelWithProjection`` state dict (``pipeline/pipeline_stable_video_diffusion_controlnet.py:22,125``)."""
    d: "OrderedDict[str, Shape]" = OrderedDict()
    c, p = cfg["hidden_size"], cfg["patch_size"]
    v = "vision_model."
    d[v + "embeddings.class_embedding"] = (c,)
    d[v + "embeddings.patch_embedding.weight"] = (c, cfg["num_channels"], p, p)
    d[v + "embeddings.position_embedding.weight"] = ((cfg["image_size"] // p) ** 2 + 1, c)
    _norm(d, v + "pre_layrnorm", c)
    for i in range(cfg["num_hidden_layers"]):
        b = f"{v}encoder.layers.{i}."
        for k in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _lin(d, b + "self_attn." + k, c, c)
        _norm(d, b + "layer_norm1", c)
        _lin(d, b + "mlp.fc1", c, cfg["intermediate_size"]); _lin(d, b + "mlp.fc2", cfg["intermediate_size"], c)
        _norm(d, b + "layer_norm2", c)
    _norm(d, v + "post_layernorm", c)
    _lin(d, "visual_projection", c, cfg["projection_dim"], False)
    return d


def n_params(spec: Dict[str, Shape]) -> int:
    t = 0
    for s in spec.values():
        k = 1
        for v in s:
            k *= v
        t += k
    return t
