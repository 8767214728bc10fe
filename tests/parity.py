"""Shared parity harness: CPU oracle (fp32) vs the HIP path on identical weights and inputs.
Used by tests/test_model_gpu.py and by __graft_entry__.smoke().  The oracle is only ever the checker here."""
from __future__ import annotations

import contextlib
import io
import os

import torch

from oracle import blocks as OB, init as OI, loop as OL, nets as ON, quant as OQ, sched as OS


def usable_cores() -> int:
    """Affinity mask capped by the cgroup CPU quota (the same rule as bench.py: usable_cores).  The GPU box shows 256
    logical CPUs behind a 16-CPU quota; with torch's default of 128 threads there the CPU oracle of the parity tests ran
    4-8 x slower (GPU suite 12:46 -> 2:52 with the count set)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


torch.set_num_threads(usable_cores())
TINY = ON.tiny_config()
TINY_CE = (8, 16, 32, 64)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def build_oracle_nets(seed=0, camera=False, cfg=None, ce=TINY_CE):
    cfg = cfg or TINY
    with contextlib.redirect_stdout(io.StringIO()):
        unet = OI.seeded_init_(ON.UNetSpatioTemporalConditionControlNetModel(**cfg), seed=seed + 1).eval()
        cn = OI.seeded_init_(ON.ControlNetSDVModel(**cfg, conditioning_embedding_out_channels=ce, camera=camera),
                             seed=seed + 2).eval()
    # both sides compute from the same fp16-representable weights
    with torch.no_grad():
        for m in (unet, cn):
            for p in m.parameters():
                p.copy_(p.half().float())
    return cn, unet


def build_hip_nets(cn_o, unet_o, device, camera=False, cfg=None, ce=TINY_CE):
    from posetraj_amd.controlnet_sdv import ControlNetSDVModel
    from posetraj_amd.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel
    cfg = cfg or TINY
    unet = UNetSpatioTemporalConditionControlNetModel(**cfg).load_state_dict(unet_o.state_dict(), device)
    cn = ControlNetSDVModel(**cfg, conditioning_embedding_out_channels=ce, camera=camera).load_state_dict(
        cn_o.state_dict(), device)
    return cn, unet


def tiny_inputs(seed=0, B=2, F=14, h=8, w=8, xdim=64):
    g = torch.Generator().manual_seed(seed)
    r16 = lambda t: t.half().float()
    return dict(
        sample=r16(torch.randn(B, F, 8, h, w, generator=g)),
        t=torch.tensor(1.137),
        ehs=r16(torch.cat([torch.zeros(B // 2, 1, xdim), torch.randn(B - B // 2, 1, xdim, generator=g)])),
        ids=torch.tensor([[6, 128, 0.02]] * B),
        cond=r16(torch.rand(B, F, 3, h * 8, w * 8, generator=g) * 2 - 1),
        cam=r16(torch.randn(B, F, 12, generator=g) * 0.3),
    )


def run_tiny_pipeline_parity(steps=2, latent_hw=(8, 8), frames=14, device="cuda:0", camera=False, seed=0,
                             return_all=False, modes=("fp32",), cfg=None, ce=TINY_CE, nets=None, **denoise_kw):
    """2-step (default) CFG denoise of one clip: oracle loop on CPU vs StableVideoDiffusionPipelineControlNet.denoise.
    ``modes``: oracle storage precisions to run (oracle/quant.py); the returned rel-L2 is against the first one, and
    with ``return_all`` a dict of every pairwise distance of the ladder comes back as well."""
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    h, w = latent_hw
    if nets is None:                                        # (cn_o, unet_o, cn_h, unet_h) built by the caller, or the tiny nets / cfg
        cn_o, unet_o = build_oracle_nets(seed, camera, cfg=cfg, ce=ce)
        cn_h, unet_h = build_hip_nets(cn_o, unet_o, device, camera, cfg=cfg, ce=ce)
    else:
        cn_o, unet_o, cn_h, unet_h = nets
    g = torch.Generator().manual_seed(seed + 5)
    r16 = lambda t: t.half().float()
    lat = torch.randn(1, frames, 4, h, w, generator=g)
    mode = r16(torch.randn(1, 4, h, w, generator=g))
    il = torch.cat([torch.zeros_like(mode), mode])                                       # [2,4,h,w]
    e = r16(torch.randn(1, 1, unet_o.config.cross_attention_dim, generator=g))
    emb = torch.cat([torch.zeros_like(e), e])
    cond1 = r16(torch.rand(1, frames, 3, h * 8, w * 8, generator=g) * 2 - 1)
    cond = torch.cat([cond1] * 2)
    cam = None
    if camera:
        c1 = r16(torch.randn(1, frames, 12, generator=g) * 0.3)
        cam = torch.cat([c1] * 2)
    refs = {}
    for m in modes:
        so = OS.OracleEulerDiscreteScheduler(**OS.SVD_SCHEDULER_CONFIG)
        so.set_timesteps(steps)
        lat0 = lat * so.init_noise_sigma
        with OQ.storage(m):
            refs[m] = OL.denoise(cn_o, unet_o, so, latents=lat0, image_latents=il.unsqueeze(1).repeat(1, frames, 1, 1, 1),
                                 image_embeddings=emb, controlnet_condition=cond, num_inference_steps=steps,
                                 controlnet_cond_scale=0.9, camera_cond=cam)
    ref = refs[modes[0]]
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet_h, controlnet=cn_h,
                                                  scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
    out = pipe.denoise(lat0.to(device), il.to(device), emb.to(device), cond.to(device), num_inference_steps=steps,
                       controlnet_cond_scale=0.9, camera_cond=None if cam is None else cam.to(device), **denoise_kw)
    torch.cuda.synchronize()
    r = rel_l2(out, ref)
    if not return_all:
        return r
    if len(modes) == 1:
        return r, out.cpu(), ref
    return r, out.cpu(), ref, ladder_distances(out, refs)


def loop_inputs(seed, frames, h, w, xdim):
    """The seeded inputs of one clip for the denoise loop - the same draws, in the same order, as run_tiny_pipeline_parity makes
    (tools/full_width_L_25step_parity.py and the stored-oracle test build both sides from this)."""
    g = torch.Generator().manual_seed(seed + 5)
    r16 = lambda t: t.half().float()
    lat = torch.randn(1, frames, 4, h, w, generator=g)
    mode = r16(torch.randn(1, 4, h, w, generator=g))
    il = torch.cat([torch.zeros_like(mode), mode])
    e = r16(torch.randn(1, 1, xdim, generator=g))
    emb = torch.cat([torch.zeros_like(e), e])
    cond1 = r16(torch.rand(1, frames, 3, h * 8, w * 8, generator=g) * 2 - 1)
    return lat, il, emb, torch.cat([cond1] * 2)


def build_oracle_camera_controlnet(seed=23):
    """The camera twin's ControlNet (controlnet_sdv_cam: cc_projection on the 1/8-resolution map) at the full SVD width, seeded, with
    fp16-representable parameters - BASELINE configs[4]."""
    with contextlib.redirect_stdout(io.StringIO()):
        cn = OI.seeded_init_(ON.ControlNetSDVModel(**SVD_CFG, conditioning_embedding_out_channels=SVD_CE, camera=True), seed=seed).eval()
    with torch.no_grad():
        for p_ in cn.parameters():
            p_.copy_(p_.half().float())
    return cn


def loop_camera_input(seed, frames):
    """Per-frame camera R|T ``[2, frames, 12]`` (CFG-doubled) for the 25-step camera fixtures: its own generator, so that the draws of
    loop_inputs stay what they are."""
    g = torch.Generator().manual_seed(seed + 105)
    c1 = (torch.randn(1, frames, 12, generator=g) * 0.3).half().float()
    return torch.cat([c1] * 2)


def tensor_digest(*ts):
    import hashlib
    m = hashlib.sha256()
    for t in ts:
        m.update(t.detach().contiguous().cpu().numpy().tobytes())
    return m.hexdigest()[:16]


def weights_digest(*nets):
    """sha256 over the names and (strided samples of) the values of every tensor of the given oracle networks."""
    import hashlib
    m = hashlib.sha256()
    for n in nets:
        for k, v in sorted(n.state_dict().items()):
            if v.numel() > 4096:
                v = v.flatten()[:: max(1, v.numel() // 4096)]
            m.update(k.encode()); m.update(v.detach().float().contiguous().numpy().tobytes())
    return m.hexdigest()[:16]


def ladder_distances(hip, refs):
    """rel-L2 of every pair of {HIP, oracle at each storage precision}; the denominator is always the second (more
    precise) member: 'hip|fp16-fused', 'hip|fp32', 'fp16-fused|fp32', 'fp16|fp32', ..."""
    order = [m for m in ("fp16", "fp16-fused", "fp32") if m in refs]
    d = {f"hip|{m}": rel_l2(hip, refs[m]) for m in order}
    for i, a in enumerate(order):
        for b in order[i + 1:]:
            d[f"{a}|{b}"] = rel_l2(refs[a], refs[b])
    return d


# the SVD configuration (BASELINE configs[1..4]): 1.52 B (U-Net) + 0.68 B (ControlNet) parameters
SVD_CFG = ON.svd_config()
SVD_CE = (16, 32, 96, 256)


def net_ladder(device="cuda:0", latent_hw=(16, 16), seed=0, modes=("fp32", "fp16-fused", "fp16"), cfg=None, ce=TINY_CE,
               return_nets=False, nets=None):
    """ControlNet mid tap and U-Net output of the tiny nets (or of ``cfg`` / ``ce``): HIP and the oracle at each storage
    precision.  Returns ``{"controlnet_mid": {pair: rel-L2}, "unet": {...}}``.  The U-Net legs all consume the fp32
    oracle's ControlNet residuals so that the two networks' errors are reported separately."""
    if nets is None:
        cn_o, unet_o = build_oracle_nets(seed, cfg=cfg, ce=ce)
        cn_h, unet_h = build_hip_nets(cn_o, unet_o, device, cfg=cfg, ce=ce)
    else:
        cn_o, unet_o, cn_h, unet_h = nets
    i = tiny_inputs(seed=seed + 1, h=latent_hw[0], w=latent_hw[1], xdim=unet_o.config.cross_attention_dim)
    j = {k: v.to(device) for k, v in i.items()}
    cn_refs, un_refs = {}, {}
    with torch.no_grad():
        down32, mid32 = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False)
        for m in modes:
            with OQ.storage(m):
                cn_refs[m] = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False)[1]
                un_refs[m] = unet_o(i["sample"], i["t"], i["ehs"], down32, mid32, return_dict=False,
                                    added_time_ids=i["ids"])[0]
    mid_h = cn_h(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(), return_dict=False)[1]
    y_h = unet_h(j["sample"].half(), j["t"], j["ehs"].half(), [d.half().to(device) for d in down32],
                 mid32.half().to(device), return_dict=False, added_time_ids=j["ids"])[0]
    d = {"controlnet_mid": ladder_distances(mid_h, cn_refs), "unet": ladder_distances(y_h, un_refs)}
    return (d, (cn_o, unet_o, cn_h, unet_h)) if return_nets else d


# --------------------------------------------------------------------------------------------- full-width blocks
# (channels, heads, latent h x w) of the four U-Net levels at the 14 x 576 x 1024 workload (BASELINE configs[2])
FULL_WIDTH_LEVELS = {0: (320, 5, (72, 128)), 1: (640, 10, (36, 64)), 2: (1280, 20, (18, 32)), 3: (1280, 20, (9, 16))}


def full_width_block(level, device="cuda:0", **kw):
    """full_width_level0_block at the width and geometry of U-Net level `level` (3 = the mid block's geometry)."""
    C, heads, hw = FULL_WIDTH_LEVELS[level]
    return full_width_level0_block(device, latent_hw=hw, C=C, heads=heads, seed=100 * level, **kw)


def full_width_level0_block(device="cuda:0", latent_hw=(72, 128), frames=14, B=2, C=320, heads=5, xdim=1024, temb_dim=1280,
                            seed=0, mode="fp32"):
    """One CrossAttnDownBlockSpatioTemporal layer pair at the FULL SVD width of level 0 - SpatioTemporalResBlock(C->C)
    followed by TransformerSpatioTemporalModel(heads x 64) - at the 14 x 576 x 1024 geometry, CFG batch 2: HIP blocks
    (posetraj_amd/blocks.py through the C ABI) against the oracle's modules (CPU, ~1.6 TFLOP).  Returns the rel-L2 after
    the residual block and after the transformer."""
    from posetraj_amd import blocks as HB, ops
    from posetraj_amd.packing import vec16  # noqa: F401  (packing is exercised through the block constructors)
    h, w = latent_hw
    N = B * frames
    res_o = OI.seeded_init_(OB.SpatioTemporalResBlock(C, C, temb_dim, eps=1e-6), seed=seed + 11).eval()
    att_o = OI.seeded_init_(OB.TransformerSpatioTemporalModel(heads, C // heads, C, num_layers=1, cross_attention_dim=xdim),
                            seed=seed + 12).eval()
    with torch.no_grad():
        for m in (res_o, att_o):
            for p in m.parameters():
                p.copy_(p.half().float())
    g = torch.Generator().manual_seed(seed + 13)
    r16 = lambda t: t.half().float()
    x = r16(torch.randn(N, C, h, w, generator=g))
    emb = r16(torch.randn(B, temb_dim, generator=g))                       # one time embedding per clip half
    ehs = r16(torch.randn(B, 1, xdim, generator=g))
    ind = torch.zeros(B, frames)
    with torch.no_grad(), OQ.storage(mode):
        y1_o = res_o(x, emb.repeat_interleave(frames, dim=0), ind)
        y2_o = att_o(y1_o, ehs.repeat_interleave(frames, dim=0), ind)
    # ---- HIP side: the same two blocks, weights from the oracle's state dicts
    sd = {("r." + k): v for k, v in res_o.state_dict().items()}
    sd.update({("a." + k): v for k, v in att_o.state_dict().items()})
    ts, xs = HB.RowStack(), HB.RowStack()
    res_h = HB.SpatioTemporalResBlock(sd, "r.", 1e-6, device, ts)
    att_h = HB.TransformerSpatioTemporalModel(sd, "a.", heads, device, xs)
    temb_all, xattn_all = ts.pack(device), xs.pack(device)
    e16 = emb.to(device).half()
    temb = ops.igemm(ops.silu(e16), temb_all)
    xat = ops.igemm(ehs.reshape(B, -1).to(device).half().contiguous(), xattn_all)
    ctx = HB.Ctx(B=B, F=frames, temb=temb, xattn=xat)
    xh = ops.to_channels_last(x.to(device).half())
    y1_h = res_h.run(ctx, xh)
    y2_h = att_h.run(ctx, y1_h)
    torch.cuda.synchronize()
    return rel_l2(y1_h.permute(0, 3, 1, 2), y1_o), rel_l2(y2_h.permute(0, 3, 1, 2), y2_o)



# --------------------------------------------------------------------------------------------- block-composition fixture
# tests/golden/blocks.npz: the reference's own forwards (models/modified_svd.py:50-348) run over these oracle modules by
# tests/golden/make_golden.py: gen_blocks.  CFG batch 2 x 14 frames (the batch-interleaved time_context, SURVEY Q3, is
# live); head_dim 64 so the HIP blocks run against the same fixture.
BLK = dict(B=2, F=14, C=64, C2=128, temb=128, xdim=32, h=4, w=6)


def blocks_modules():
    """The four oracle modules of the fixture, rebuilt from (seed, parameter name, shape); weights (and the inputs below)
    are fp16-representable so that the HIP blocks can be held to the same fixture without a weight-rounding term."""
    c, c2, te, xd = BLK["C"], BLK["C2"], BLK["temb"], BLK["xdim"]
    mods = dict(
        temporal=OI.seeded_init_(OB.TemporalBasicTransformerBlock(c, c, 1, 64, xd), seed=51).eval(),
        transformer=OI.seeded_init_(OB.TransformerSpatioTemporalModel(1, 64, c, num_layers=1, cross_attention_dim=xd), seed=52).eval(),
        down=OI.seeded_init_(OB.CrossAttnDownBlockSpatioTemporal(c, c2, te, 2, 1, 2, xd, True), seed=53).eval(),
        up=OI.seeded_init_(OB.CrossAttnUpBlockSpatioTemporal(c, c2, c2, te, 3, 1, 1e-5, 2, xd, True), seed=54).eval(),
    )
    with torch.no_grad():
        for m in mods.values():
            for p in m.parameters():
                p.copy_(p.half().float())
    return mods


def blocks_inputs():
    B, F, c, c2, te, xd, h, w = (BLK[k] for k in ("B", "F", "C", "C2", "temb", "xdim", "h", "w"))
    g = torch.Generator().manual_seed(61)
    e = torch.randn(1, 1, xd, generator=g)
    ehs = torch.cat([torch.zeros_like(e), e]).repeat_interleave(F, dim=0)                 # CFG: zeros | embedding
    r16 = lambda d: {k: v.half().float() for k, v in d.items()}
    return r16(dict(
        tokens=torch.randn(B * F, h * w, c, generator=g),
        tctx=torch.randn(h * w * B, 1, xd, generator=g),
        x=torch.randn(B * F, c, h, w, generator=g),
        ehs=ehs,
        temb=torch.randn(B, te, generator=g).repeat_interleave(F, dim=0),
        up_x=torch.randn(B * F, c2, h, w, generator=g),
        up_skips=torch.stack([torch.randn(B * F, c2, h, w, generator=g) for _ in range(2)]),   # popped last-first
        up_skip_in=torch.randn(B * F, c, h, w, generator=g),                                    # the last resnet's skip (in_channels)
    ))


def hip_blocks_vs_fixture(g, device="cuda:0"):
    """The HIP transformer / down block / up block (posetraj_amd/blocks.py through the C ABI), weights from the fixture's
    recipe, against tests/golden/blocks.npz - outputs of the REFERENCE's forwards.  Returns rel-L2 per output."""
    from posetraj_amd import blocks as HB, ops
    B, F = BLK["B"], BLK["F"]
    i = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    mods = blocks_modules()
    cl = lambda x: ops.to_channels_last(x.to(device).half())
    nchw = lambda y: y.permute(0, 3, 1, 2)
    emb = i["temb"][::F].contiguous()                                        # one time embedding per clip half
    ehs = i["ehs"][::F].reshape(B, -1).contiguous()                          # first-frame context per clip half

    def ctx_for(ts, xs):
        temb_all, xattn_all = ts.pack(device), xs.pack(device)
        temb = ops.igemm(ops.silu(emb.to(device).half()), temb_all) if temb_all is not None else None
        return HB.Ctx(B=B, F=F, temb=temb, xattn=ops.igemm(ehs.to(device).half(), xattn_all))

    out = {}
    # -- TransformerSpatioTemporalModel (1 head x 64)
    ts, xs = HB.RowStack(), HB.RowStack()
    sd = {"a." + k: v for k, v in mods["transformer"].state_dict().items()}
    tr = HB.TransformerSpatioTemporalModel(sd, "a.", 1, device, xs)
    y = tr.run(ctx_for(ts, xs), cl(i["x"]))
    out["transformer"] = rel_l2(nchw(y), torch.from_numpy(g["transformer"]))
    # -- CrossAttnDownBlockSpatioTemporal (64 -> 128, 2 layers, downsampler)
    ts, xs = HB.RowStack(), HB.RowStack()
    sd = {"d." + k: v for k, v in mods["down"].state_dict().items()}
    dn = HB.DownBlock(sd, "d.", True, 2, device, ts, xs)
    y, taps = dn.run(ctx_for(ts, xs), cl(i["x"]))
    out["down"] = rel_l2(nchw(y), torch.from_numpy(g["down"]))
    for j, tp in enumerate(taps):
        out[f"down_tap{j}"] = rel_l2(nchw(tp), torch.from_numpy(g[f"down_tap{j}"]))
    # -- CrossAttnUpBlockSpatioTemporal (3 layers, skips popped last-first, upsampler)
    ts, xs = HB.RowStack(), HB.RowStack()
    sd = {"u." + k: v for k, v in mods["up"].state_dict().items()}
    up = HB.UpBlock(sd, "u.", True, 2, device, ts, xs)
    skips = [cl(i["up_skip_in"]), cl(i["up_skips"][0]), cl(i["up_skips"][1])]
    y = up.run(ctx_for(ts, xs), cl(i["up_x"]), skips)
    out["up"] = rel_l2(nchw(y), torch.from_numpy(g["up"]))
    torch.cuda.synchronize()
    return out
