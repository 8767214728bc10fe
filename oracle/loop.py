"""Oracle restatement of the denoise loop of
``/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:481-583`` (camera twin
``..._cam.py``: ``camera_cond`` passed to the ControlNet, ``:505-509,549``).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  PINNED: the reference ``__call__`` itself is run over these
oracle networks (CLIP / VAE replaced by fixed tensors) and compared with this function
(``tests/golden/loop_*.npz``).
"""
from __future__ import annotations

import torch

from .quant import q


def append_dims(x, target_dims):
    """``:62-67``."""
    return x[(...,) + (None,) * (target_dims - x.ndim)]


def hot_added_time_ids(dtype):
    """``:513-523``: the U-Net micro-conditioning is hard-coded to fps=6, motion_bucket_id=128,
    noise_aug=0.02, whatever the caller asked for; duplicated for the two CFG halves."""
    ids = torch.tensor([[6, 128, 0.02]], dtype=dtype)
    return torch.cat([ids] * 2)


def guidance_ramp(min_scale, max_scale, num_frames, batch, dtype, ndim=5):
    """``:506-509``."""
    g = torch.linspace(min_scale, max_scale, num_frames).unsqueeze(0).to(dtype).repeat(batch, 1)
    return append_dims(g, ndim)


@torch.no_grad()
def denoise(controlnet, unet, scheduler, *, latents, image_latents, image_embeddings, controlnet_condition,
            num_inference_steps=25, min_guidance_scale=1.0, max_guidance_scale=3.0, controlnet_cond_scale=1.0,
            camera_cond=None, record=None):
    """latents ``[1,F,4,h,w]`` already multiplied by ``init_noise_sigma`` (``:298``); image_latents
    ``[2,F,4,h,w]`` (neg half zeros); image_embeddings ``[2,1,D]``; controlnet_condition ``[2,F,3,H,W]``
    (the same maps in both halves, ``:500-503``).

    ``max_guidance_scale <= 1`` (``:438``: no classifier-free guidance): the reference feeds the un-doubled latents
    (``:532``; image_latents / image_embeddings ``[1,...]``) but still doubles the control maps (``:501-503``) and
    ``added_time_ids`` (``:521``), so the ControlNet's ``add_embedding`` receives ``[1, 2 * 3 * D]`` and raises
    ``RuntimeError`` (tests/golden/loop.npz: ``*_noncfg_raises``); the restatement reaches the same error the same way."""
    scheduler.set_timesteps(num_inference_steps)
    cfg = max_guidance_scale > 1.0
    nf = latents.shape[1]
    g = guidance_ramp(min_guidance_scale, max_guidance_scale, nf, latents.shape[0], latents.dtype, latents.ndim)
    ids = hot_added_time_ids(image_embeddings.dtype)
    for t in scheduler.timesteps:
        x = q(scheduler.scale_model_input(torch.cat([latents] * 2) if cfg else latents, t))
        x = torch.cat([x, image_latents], dim=2)
        kw = dict(camera_cond=camera_cond) if camera_cond is not None else {}
        down, mid = controlnet(x, t, encoder_hidden_states=image_embeddings, controlnet_cond=controlnet_condition,
                               added_time_ids=ids, conditioning_scale=controlnet_cond_scale, guess_mode=False,
                               return_dict=False, **kw)
        pred = unet(x, t, encoder_hidden_states=image_embeddings, down_block_additional_residuals=down,
                    mid_block_additional_residual=mid, added_time_ids=ids, return_dict=False)[0]
        if cfg:
            un, co = pred.chunk(2)
            pred = q(un + g * (co - un))        # fp16 in the reference; the MI355X path keeps the guidance + Euler update in fp32
        latents = q(scheduler.step(pred, t, latents).prev_sample)
        if record is not None:
            record.append(latents.clone())
    return latents
