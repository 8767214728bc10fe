#!/usr/bin/env python3
"""Image-to-video at the FULL model sizes against the oracle chain (a tool: ~10 min of host time): U-Net (1.52 B) + ControlNet
(0.68 B) + AutoencoderKLTemporalDecoder (97.7 M) + CLIP ViT-H/14 (632 M, all 32 layers), seeded random init, one 14-frame clip at
128 x 128 px (latent 16 x 16), `steps` Euler steps, trajectory maps from the rasteriser - `pipe(PIL image, maps, ...)` with the
default output_type="pil" on the MI355X vs resize -> CLIP -> VAE encode -> loop -> decode_latents(chunk 8) -> tensor2vid built
from oracle/ on the CPU.      python tools/e2e_full_width_parity.py [--steps 3]"""
import argparse, contextlib, io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import PIL.Image
from tests import parity as P
from oracle import clip as OCL, init as OI, loop as OL, raster as ORA, resize as OR, sched as OS, vae as OV
from posetraj_amd import (AutoencoderKLTemporalDecoder, CLIPVisionModelWithProjection, EulerDiscreteScheduler, SVD_SCHEDULER_CONFIG,
                          StableVideoDiffusionPipelineControlNet)
from posetraj_amd.trajectory import trajectory_maps
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--px", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda:0")
t0 = time.time()
cn_o, un_o = P.build_oracle_nets(7, cfg=P.SVD_CFG, ce=P.SVD_CE)
cn_h, un_h = P.build_hip_nets(cn_o, un_o, dev, cfg=P.SVD_CFG, ce=P.SVD_CE)
vae_o = OI.seeded_init_(OV.AutoencoderKLTemporalDecoder(**OV.svd_vae_config()), seed=77).eval()
clip_o = OI.seeded_init_(OCL.CLIPVisionModelWithProjection(**OCL.vit_h_config()), seed=78).eval()
with torch.no_grad():
    for m in (vae_o, clip_o):
        for p in m.parameters():
            p.copy_(p.half().float())
vae_h = AutoencoderKLTemporalDecoder(**OV.svd_vae_config()).load_state_dict(vae_o.state_dict(), dev)
clip_h = CLIPVisionModelWithProjection(**OCL.vit_h_config()).load_state_dict(clip_o.state_dict(), dev)
print(f"models built in {time.time() - t0:.0f} s", flush=True)
H = W = a.px
F = 14
rng = np.random.default_rng(5)
image = PIL.Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
tracks = bench.synth_tracks(F, H, W, 3)
maps_h = trajectory_maps(tracks, [H, W], (H, W, 3), num_frames=F, device=dev)
pipe = StableVideoDiffusionPipelineControlNet(vae=vae_h, image_encoder=clip_h, unet=un_h, controlnet=cn_h,
                                              scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
t0 = time.time()
out = pipe(image, maps_h, height=H, width=W, num_frames=F, decode_chunk_size=8, num_inference_steps=a.steps,
           generator=torch.Generator().manual_seed(21))
got = np.stack([np.asarray(im) for im in out.frames[0]])
lat_h = pipe(image, maps_h, height=H, width=W, num_frames=F, num_inference_steps=a.steps, generator=torch.Generator().manual_seed(21),
             output_type="latent").frames
torch.cuda.synchronize()
print(f"HIP: two calls in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
Pp = StableVideoDiffusionPipelineControlNet
with torch.no_grad():
    e = clip_o(OR.resize_with_antialiasing(Pp._to_unit_tensor(image), (224, 224))).image_embeds.unsqueeze(1)
    emb = torch.cat([torch.zeros_like(e), e])
    img = Pp.preprocess_condition(image, H, W)
    g = torch.Generator().manual_seed(21)
    img = img + 0.02 * torch.randn(img.shape, generator=g)
    mode = vae_o.encode(img).latent_dist.mode()
    il = torch.cat([torch.zeros_like(mode), mode]).unsqueeze(1).repeat(1, F, 1, 1, 1)
    s = OS.OracleEulerDiscreteScheduler(**OS.SVD_SCHEDULER_CONFIG)
    s.set_timesteps(a.steps)
    lat0 = torch.randn((1, F, 4, H // 8, W // 8), generator=g, dtype=torch.float16).float() * s.init_noise_sigma
    cond = torch.from_numpy(ORA.trajectory_maps(tracks, [H, W], (H, W, 3), num_frames=F))
    lat = OL.denoise(cn_o, un_o, s, latents=lat0, image_latents=il, image_embeddings=emb,
                     controlnet_condition=torch.cat([cond.unsqueeze(0)] * 2), num_inference_steps=a.steps)
    want = np.stack([np.asarray(im) for im in OV.tensor2vid(OV.decode_latents(vae_o, lat, F, 8), None, "pil")[0]])
d = np.abs(got.astype(int) - want.astype(int))
print(f"oracle chain: {time.time() - t0:.0f} s")
print(f"full-size models, image -> {F} x {H} x {W} video, {a.steps} steps: latents after the loop rel-L2 {P.rel_l2(lat_h, lat):.3e}; "
      f"PIL frames: max |diff| {d.max()} grey levels, mean {d.mean():.3f}, {100 * np.mean(d > 0):.1f} % of values differ "
      f"(CLIP embedding rel-L2 {P.rel_l2(clip_h(OR.resize_with_antialiasing(Pp._to_unit_tensor(image), (224, 224)).to(dev)).image_embeds, e[:, 0]):.2e})")
