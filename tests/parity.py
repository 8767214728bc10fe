"""Shared parity harness: CPU oracle (fp32) vs the HIP path on identical weights and inputs.
Used by tests/test_model_gpu.py and by __graft_entry__.smoke().  The oracle is only ever the checker here."""
from __future__ import annotations

import contextlib
import io

import torch

from oracle import init as OI, loop as OL, nets as ON, sched as OS

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
                             return_all=False):
    """2-step (default) CFG denoise of one clip: oracle loop on CPU vs StableVideoDiffusionPipelineControlNet.denoise."""
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    h, w = latent_hw
    cn_o, unet_o = build_oracle_nets(seed, camera)
    cn_h, unet_h = build_hip_nets(cn_o, unet_o, device, camera)
    g = torch.Generator().manual_seed(seed + 5)
    r16 = lambda t: t.half().float()
    lat = torch.randn(1, frames, 4, h, w, generator=g)
    mode = r16(torch.randn(1, 4, h, w, generator=g))
    il = torch.cat([torch.zeros_like(mode), mode])                                       # [2,4,h,w]
    e = r16(torch.randn(1, 1, TINY["cross_attention_dim"], generator=g))
    emb = torch.cat([torch.zeros_like(e), e])
    cond1 = r16(torch.rand(1, frames, 3, h * 8, w * 8, generator=g) * 2 - 1)
    cond = torch.cat([cond1] * 2)
    cam = None
    if camera:
        c1 = r16(torch.randn(1, frames, 12, generator=g) * 0.3)
        cam = torch.cat([c1] * 2)
    so = OS.OracleEulerDiscreteScheduler(**OS.SVD_SCHEDULER_CONFIG)
    so.set_timesteps(steps)
    lat0 = lat * so.init_noise_sigma
    ref = OL.denoise(cn_o, unet_o, so, latents=lat0, image_latents=il.unsqueeze(1).repeat(1, frames, 1, 1, 1),
                     image_embeddings=emb, controlnet_condition=cond, num_inference_steps=steps,
                     controlnet_cond_scale=0.9, camera_cond=cam)
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet_h, controlnet=cn_h,
                                                  scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
    out = pipe.denoise(lat0.to(device), il.to(device), emb.to(device), cond.to(device), num_inference_steps=steps,
                       controlnet_cond_scale=0.9, camera_cond=None if cam is None else cam.to(device))
    torch.cuda.synchronize()
    r = rel_l2(out, ref)
    return (r, out.cpu(), ref) if return_all else r
