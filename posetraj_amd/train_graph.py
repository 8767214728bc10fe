"""Training-mode forward of the ControlNet and of the frozen U-Net's up path on the tape of ``autodiff.py`` (SURVEY 8f4;
``/root/reference/scripts/train_svd_traj_VIPSeg_14.py:1347-1371,1388-1404``).

Same blocks as ``blocks.py`` (the composition the reference restates at ``models/modified_svd.py:50-348``), same channels-last
``[N = B F, H, W, C]`` layout - temporal convolutions as (3 x 1) kernels over the image (F, H W), temporal attention through
a row stride - but un-fused where a backward needs the intermediate (GEGLU keeps its projection, the time-embedding rows
and the collapsed cross-attention are separate adds), and every primitive records its backward.

What needs a gradient: all of the ControlNet (``controlnet.requires_grad_(True)``, ``:1053``), and of the U-Net
(``unet.requires_grad_(False)``, ``:953``) only the path from the places the residuals enter (``unet...:451-469``) to the
output: the up blocks, ``conv_norm_out``, ``conv_out``.  Its encoder half runs on the inference kernels.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import autodiff as AD
from . import ops
from .autodiff import Affine, Dense, Mix, Tape, Var
from .modeling import _tup


class TrainCtx:
    """Per-forward state: one clip (B = 1; the reference's spatial loss indexes by frame, ``:1396-1402``)."""

    def __init__(self, F: int, emb_silu: Var, ehs: Var):
        self.B, self.F, self.emb_silu, self.ehs = 1, F, emb_silu, ehs


class ResBlock:
    """``SpatioTemporalResBlock`` (blocks.SpatioTemporalResBlock) with a backward."""

    def __init__(self, P, p, eps):
        s, t = p + "spatial_res_block.", p + "temporal_res_block."
        self.eps = eps
        conv = lambda k: Dense(P, k + ".weight", k + ".bias", kind="conv", padding=1)
        tconv = lambda k: Dense(P, k + ".weight", k + ".bias", kind="conv_t3")
        lin = lambda k: Dense(P, k + ".weight", k + ".bias")
        self.n1, self.conv1, self.temb, self.n2, self.conv2 = Affine(P, s + "norm1"), conv(s + "conv1"), lin(s + "time_emb_proj"), Affine(P, s + "norm2"), conv(s + "conv2")
        self.shortcut = Dense(P, s + "conv_shortcut.weight", s + "conv_shortcut.bias", kind="conv", padding=0) if P.has(s + "conv_shortcut.weight") else None
        self.tn1, self.tconv1, self.ttemb, self.tn2, self.tconv2 = Affine(P, t + "norm1"), tconv(t + "conv1"), lin(t + "time_emb_proj"), Affine(P, t + "norm2"), tconv(t + "conv2")
        self.mix = Mix(P, p + "time_mixer.mix_factor")

    def run(self, tape: Tape, ctx: TrainCtx, x: Var, geom, x1: Optional[Var] = None) -> Var:
        N, H, W = geom
        S, F, B = H * W, ctx.F, ctx.B
        y = AD.groupnorm(tape, x, self.n1, rows_per_sample=S, n_samples=N, eps=self.eps, silu=True, x1=x1)
        h = AD.dense(tape, y, self.conv1, geom=geom)
        h = AD.add_rowvec(tape, h, AD.dense(tape, ctx.emb_silu, self.temb), F * S)
        y = AD.groupnorm(tape, h, self.n2, rows_per_sample=S, n_samples=N, eps=self.eps, silu=True)
        sc = x if self.shortcut is None else AD.dense(tape, x, self.shortcut, geom=geom, x1=x1)
        xs = AD.dense(tape, y, self.conv2, geom=geom, res=sc)
        tgeom = (B, F, S)
        y = AD.groupnorm(tape, xs, self.tn1, rows_per_sample=F * S, n_samples=B, eps=self.eps, silu=True)
        h = AD.dense(tape, y, self.tconv1, geom=tgeom)
        h = AD.add_rowvec(tape, h, AD.dense(tape, ctx.emb_silu, self.ttemb), F * S)
        y = AD.groupnorm(tape, h, self.tn2, rows_per_sample=F * S, n_samples=B, eps=self.eps, silu=True)
        xt = AD.dense(tape, y, self.tconv2, geom=tgeom, res=xs)
        return AD.blend(tape, xs, xt, self.mix)


class Transformer:
    """``TransformerSpatioTemporalModel`` (blocks.TransformerSpatioTemporalModel) with a backward.  The cross-attentions see ONE
    key (the image embedding): softmax over one logit is 1, so the block adds ``to_out(to_v(context))`` to every token and
    ``to_q`` / ``to_k`` / ``norm2`` receive exactly zero gradient - as under autograd."""

    def __init__(self, P, p, heads):
        self.heads = heads
        lin = lambda k, bias=True: Dense(P, k + ".weight", k + ".bias" if bias else None)
        qkv = lambda k: Dense(P, k + "to_q.weight", None, stack=(k + "to_q.weight", k + "to_k.weight", k + "to_v.weight"))
        self.norm, self.proj_in, self.proj_out = Affine(P, p + "norm"), lin(p + "proj_in"), lin(p + "proj_out")
        self.tpe1, self.tpe2 = lin(p + "time_pos_embed.linear_1"), lin(p + "time_pos_embed.linear_2")
        self.mix = Mix(P, p + "time_mixer.mix_factor")
        self.layers = []
        i = 0
        while P.has(f"{p}transformer_blocks.{i}.norm1.weight"):
            a, b = f"{p}transformer_blocks.{i}.", f"{p}temporal_transformer_blocks.{i}."
            L = type("Layer", (), {})()
            L.ln1, L.qkv, L.o = Affine(P, a + "norm1"), qkv(a + "attn1."), lin(a + "attn1.to_out.0")
            L.xv, L.xo = lin(a + "attn2.to_v", bias=False), lin(a + "attn2.to_out.0")
            L.ln3, L.ff1, L.ff2 = Affine(P, a + "norm3"), lin(a + "ff.net.0.proj"), lin(a + "ff.net.2")
            L.ln_in, L.fi1, L.fi2 = Affine(P, b + "norm_in"), lin(b + "ff_in.net.0.proj"), lin(b + "ff_in.net.2")
            L.tln1, L.tqkv, L.to = Affine(P, b + "norm1"), qkv(b + "attn1."), lin(b + "attn1.to_out.0")
            L.txv, L.txo = lin(b + "attn2.to_v", bias=False), lin(b + "attn2.to_out.0")
            L.tln3, L.tf1, L.tf2 = Affine(P, b + "norm3"), lin(b + "ff.net.0.proj"), lin(b + "ff.net.2")
            self.layers.append(L)
            i += 1

    def run(self, tape: Tape, ctx: TrainCtx, x: Var, geom) -> Var:
        N, H, W = geom
        S, F, B, heads = H * W, ctx.F, ctx.B, self.heads
        C = x.v.shape[-1]
        hd = C // heads
        h = AD.dense(tape, AD.groupnorm(tape, x, self.norm, rows_per_sample=S, n_samples=N, eps=1e-6, silu=False), self.proj_in)
        t = Var(ops.timestep_embedding(torch.arange(F, dtype=torch.float32, device=x.v.device), C), need=False)
        emb = AD.dense(tape, AD.silu(tape, AD.dense(tape, t, self.tpe1)), self.tpe2)                      # [F, C]
        ff = lambda y, w1, w2, res: AD.dense(tape, AD.geglu(tape, AD.dense(tape, y, w1)), w2, res=res)
        for L in self.layers:
            a = AD.attn_spatial(tape, AD.dense(tape, AD.layernorm(tape, h, L.ln1), L.qkv), N, S, heads, hd)
            h = AD.dense(tape, a, L.o, res=h)
            h = AD.add_rowvec(tape, h, AD.dense(tape, AD.dense(tape, ctx.ehs, L.xv), L.xo), F * S)
            hs = ff(AD.layernorm(tape, h, L.ln3), L.ff1, L.ff2, h)
            u = AD.add_rowvec(tape, hs, emb, S)
            u = ff(AD.layernorm(tape, u, L.ln_in), L.fi1, L.fi2, u)
            a = AD.attn_temporal(tape, AD.dense(tape, AD.layernorm(tape, u, L.tln1), L.tqkv), B, F, S, heads, hd)
            u = AD.dense(tape, a, L.to, res=u)
            u = AD.add_rowvec(tape, u, AD.dense(tape, AD.dense(tape, ctx.ehs, L.txv), L.txo), F * S)
            u = ff(AD.layernorm(tape, u, L.tln3), L.tf1, L.tf2, u)
            h = AD.blend(tape, hs, u, self.mix)
        return AD.dense(tape, h, self.proj_out, res=x)


def _count(P, fmt):
    i = 0
    while P.has(fmt.format(i)):
        i += 1
    return i


class TimeEmbedding:
    """``time_embedding(time_proj(t)) + add_embedding(add_time_proj(ids))`` (``controlnet_sdv.py:551-590``), then the SiLU every
    consumer applies first."""

    def __init__(self, P, ch0, add_dim):
        lin = lambda k: Dense(P, k + ".weight", k + ".bias")
        self.ch0, self.add_dim = ch0, add_dim
        self.t1, self.t2, self.a1, self.a2 = lin("time_embedding.linear_1"), lin("time_embedding.linear_2"), lin("add_embedding.linear_1"), lin("add_embedding.linear_2")

    def run(self, tape, timestep: torch.Tensor, added_time_ids: torch.Tensor, device) -> Var:
        t = timestep.to(device=device, dtype=torch.float32).reshape(-1)[:1].contiguous()
        te = Var(ops.timestep_embedding(t, self.ch0), need=False)
        emb = AD.dense(tape, AD.silu(tape, AD.dense(tape, te, self.t1)), self.t2)
        ids = added_time_ids.to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        ae = Var(ops.timestep_embedding(ids, self.add_dim).view(1, -1), need=False)
        emb = AD.dense(tape, AD.silu(tape, AD.dense(tape, ae, self.a1)), self.a2, res=emb)
        return AD.silu(tape, emb)


class ControlNetGraph:
    """``ControlNetSDVModel.forward`` (``controlnet_sdv.py:516-650``) in training mode: 12 + 1 residuals as ``Var``s."""

    def __init__(self, P, config):
        cfg = config
        ch = tuple(cfg["block_out_channels"])
        heads = _tup(cfg["num_attention_heads"], len(ch))
        self.P, self.ch = P, ch
        conv = lambda k, **kw: Dense(P, k + ".weight", k + ".bias", kind="conv", **kw)
        e = "controlnet_cond_embedding."
        self.ce_in = conv(e + "conv_in", padding=1)
        self.ce_blocks = [conv(f"{e}blocks.{i}", padding=1, stride=2 if i % 2 else 1) for i in range(_count(P, e + "blocks.{}.weight"))]
        self.ce_out = conv(e + "conv_out", padding=1)
        self.cc = None
        if cfg.get("camera"):                                     # the camera twin: Linear(C + 12 -> C) over [features | R|T]
            cw = P.shapes[e + "cc_projection.weight"] if hasattr(P, "shapes") else tuple(P.value(e + "cc_projection.weight").shape)
            self.cc = Dense(P, e + "cc_projection.weight", e + "cc_projection.bias", kpad=(cw[1] + 7) // 8 * 8, dgrad_cols=cw[0])
        self.conv_in = conv("conv_in", padding=1)
        self.time = TimeEmbedding(P, ch[0], cfg["addition_time_embed_dim"])
        self.down = []
        for i, typ in enumerate(cfg["down_block_types"]):
            cross = typ == "CrossAttnDownBlockSpatioTemporal"
            p = f"down_blocks.{i}."
            eps = 1e-6 if cross else 1e-5
            n = _count(P, p + "resnets.{}.spatial_res_block.norm1.weight")
            blk = type("Down", (), {})()
            blk.resnets = [ResBlock(P, f"{p}resnets.{j}.", eps) for j in range(n)]
            blk.attns = [Transformer(P, f"{p}attentions.{j}.", heads[i]) for j in range(n)] if cross else []
            blk.down = conv(p + "downsamplers.0.conv", padding=1, stride=2) if P.has(p + "downsamplers.0.conv.weight") else None
            self.down.append(blk)
        self.mid = (ResBlock(P, "mid_block.resnets.0.", 1e-5), Transformer(P, "mid_block.attentions.0.", heads[-1]), ResBlock(P, "mid_block.resnets.1.", 1e-5))
        zc = lambda k: Dense(P, k + ".weight", k + ".bias")                           # 1 x 1 zero-convs as linear layers
        self.zero = [zc(f"controlnet_down_blocks.{k}") for k in range(_count(P, "controlnet_down_blocks.{}.weight"))]
        self.zero_mid = zc("controlnet_mid_block")

    def run(self, tape: Tape, sample_cl: torch.Tensor, geom, timestep, ehs: torch.Tensor, added_time_ids, cond: torch.Tensor,
            conditioning_scale: float = 1.0, camera_cond: Optional[torch.Tensor] = None):
        """``sample_cl``: the network input channels-last ``[F h w, 8]``; ``cond``: ``[F, 3, H, W]`` trajectory maps; ``camera_cond``:
        ``[F, 12]`` (R|T per frame) for the camera twin."""
        N, h, w = geom
        dev = sample_cl.device
        ctx = TrainCtx(N, self.time.run(tape, timestep, added_time_ids, dev), Var(ehs, need=False))
        Fc, Cc, H, W = cond.shape
        c = Var(ops.to_channels_last(cond, cpad=8).view(Fc * H * W, 8), need=False)
        c = AD.silu(tape, AD.dense(tape, c, self.ce_in, geom=(Fc, H, W)))
        hh, ww = H, W
        for b in self.ce_blocks:
            c = AD.silu(tape, AD.dense(tape, c, b, geom=(Fc, hh, ww)))
            if b.stride == 2:
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
        if (hh, ww) != (h, w):
            raise ValueError(f"controlnet_cond of {H} x {W} gives a {hh} x {ww} embedding for a {h} x {w} latent")
        if self.cc is not None and camera_cond is not None:       # controlnet_sdv_cam_infer.py:109-118
            cam = camera_cond.to(device=dev, dtype=torch.float16).reshape(Fc, -1).contiguous()
            if cam.shape[1] != 12:
                raise ValueError(f"camera_cond must have 12 values per frame (R|T); got {cam.shape[1]}")
            c = AD.dense(tape, AD.concat_camera(tape, c, (Fc, hh, ww), cam, self.cc.kpad), self.cc)
        c = AD.dense(tape, c, self.ce_out, geom=(Fc, hh, ww))
        x = AD.dense(tape, Var(sample_cl, need=False), self.conv_in, geom=geom, res=c)
        taps = [x]
        g = geom
        for blk in self.down:
            for j, r in enumerate(blk.resnets):
                x = r.run(tape, ctx, x, g)
                if blk.attns:
                    x = blk.attns[j].run(tape, ctx, x, g)
                taps.append(x)
            if blk.down is not None:
                x = AD.dense(tape, x, blk.down, geom=g)
                g = (g[0], (g[1] + 1) // 2, (g[2] + 1) // 2)
                taps.append(x)
        x = self.mid[2].run(tape, ctx, self.mid[1].run(tape, ctx, self.mid[0].run(tape, ctx, x, g), g), g)
        if conditioning_scale != 1.0:
            raise NotImplementedError("training uses conditioning_scale = 1.0 (controlnet_sdv.py:527)")
        outs = [AD.dense(tape, t, z) for t, z in zip(taps, self.zero)]
        return outs, AD.dense(tape, x, self.zero_mid)


class UNetDecoderGraph:
    """The frozen U-Net from the point its inputs depend on the ControlNet (``unet...:451-504``): residual adds with the
    multiplicities of SURVEY Q1, the up blocks, ``conv_norm_out`` + SiLU + ``conv_out``."""

    def __init__(self, P, config):
        cfg = config
        ch = tuple(cfg["block_out_channels"])
        rheads = _tup(cfg["num_attention_heads"], len(ch))[::-1]
        self.up = []
        for i, typ in enumerate(cfg["up_block_types"]):
            cross = typ == "CrossAttnUpBlockSpatioTemporal"
            p = f"up_blocks.{i}."
            n = _count(P, p + "resnets.{}.spatial_res_block.norm1.weight")
            blk = type("Up", (), {})()
            blk.resnets = [ResBlock(P, f"{p}resnets.{j}.", 1e-5) for j in range(n)]
            blk.attns = [Transformer(P, f"{p}attentions.{j}.", rheads[i]) for j in range(n)] if cross else []
            blk.up = Dense(P, p + "upsamplers.0.conv.weight", p + "upsamplers.0.conv.bias", kind="conv", padding=1) if P.has(p + "upsamplers.0.conv.weight") else None
            self.up.append(blk)
        self.norm_out = Affine(P, "conv_norm_out")
        self.conv_out = Dense(P, "conv_out.weight", "conv_out.bias", kind="conv", padding=1)

    def run(self, tape: Tape, state: dict, mult: List[int], residuals: List[Var], mid_residual: Var, emb_silu: torch.Tensor, ehs: torch.Tensor) -> Var:
        """``state``: what ``UNetSpatioTemporalConditionControlNetModel._encode`` returned (inference kernels)."""
        Bc, F = state["dims"]
        ctx = TrainCtx(F, Var(emb_silu, need=False), Var(ehs, need=False))
        skips = []
        for s, r, m in zip(state["skips"], residuals, mult):
            n, hh, ww, c = s.shape
            skips.append((AD.add_scaled_const(tape, s.reshape(n * hh * ww, c), r, m) if m else Var(s.reshape(n * hh * ww, c), need=False), (n, hh, ww)))
        xm = state["x"]
        n, hh, ww, c = xm.shape
        x, g = AD.add_scaled_const(tape, xm.reshape(n * hh * ww, c), mid_residual, 1.0), (n, hh, ww)
        for blk in self.up:
            for j, r in enumerate(blk.resnets):
                skip, sg = skips.pop()
                if sg != g:
                    raise RuntimeError(f"Sizes of tensors must match except in dimension 1: {g} vs {sg}")
                x = r.run(tape, ctx, x, g, x1=skip)
                if blk.attns:
                    x = blk.attns[j].run(tape, ctx, x, g)
            if blk.up is not None:
                x = AD.dense(tape, x, blk.up, geom=g, upsample2x=True)
                g = (g[0], 2 * g[1], 2 * g[2])
        y = AD.groupnorm(tape, x, self.norm_out, rows_per_sample=g[1] * g[2], n_samples=g[0], eps=1e-5, silu=True)
        return AD.dense(tape, y, self.conv_out, geom=g)
