"""The FORWARD half of the reference's ControlNet training step on the MI355X path (SURVEY 8f4;
``/root/reference/scripts/train_svd_traj_VIPSeg_14.py:1264-1414``): sigma sampling, noising + EDM preconditioning
(``pt_edm_train_input``), the training-time ``added_time_ids``, conditioning dropout, ControlNet + frozen U-Net forward on the
HIP kernels, the sigma-weighted MSE and the single-frame "spatial" loss (``pt_edm_loss``).  It evaluates the training objective
(validation loss, loss curves of a checkpoint); it does NOT train: this package has no backward kernels, optimizer or EMA - that
is the next row of the scope table, and ``controlnet_training_loss`` says so by returning plain floats / tensors without autograd
history.  The VAE encode (``tensor_to_vae_latent``, ``:495-503``: ``vae.encode(x).latent_dist.sample() * scaling_factor`` per
frame) and the CLIP embedding of the first frame are this package's own models; their outputs are this function's inputs.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import hip, ops

# scripts/train_svd_traj_VIPSeg_14.py:314-319, :1288
MIN_VALUE, MAX_VALUE, IMAGE_D, NOISE_D_LOW, NOISE_D_HIGH, SIGMA_DATA = 0.002, 700, 64, 32, 64, 0.5
TRAIN_NOISE_AUG = 0.02


def stratified_uniform(shape, group=0, groups=1, dtype=None, device=None, generator=None):
    """``:273-282``."""
    if groups <= 0:
        raise ValueError(f"groups must be positive, got {groups}")
    if group < 0 or group >= groups:
        raise ValueError(f"group must be in [0, {groups})")
    n = shape[-1] * groups
    offsets = torch.arange(group, n, groups, dtype=dtype, device=device)
    u = torch.rand(shape, dtype=dtype, device=device, generator=generator)
    return (offsets + u) / n


def rand_cosine_interpolated(shape, image_d=IMAGE_D, noise_d_low=NOISE_D_LOW, noise_d_high=NOISE_D_HIGH, sigma_data=SIGMA_DATA,
                             min_value=MIN_VALUE, max_value=MAX_VALUE, device="cpu", dtype=torch.float32, generator=None):
    """``:285-312``: one sigma per clip from the interpolated, shifted cosine log-SNR schedule.  A handful of scalars: host maths,
    identical to the reference's expression order."""
    def cosine(t, lo, hi):
        t_min = math.atan(math.exp(-0.5 * hi))
        t_max = math.atan(math.exp(-0.5 * lo))
        return -2 * torch.log(torch.tan(t_min + t * (t_max - t_min)))

    def shifted(t, noise_d, lo, hi):
        shift = 2 * math.log(noise_d / image_d)
        return cosine(t, lo - shift, hi - shift) + shift

    logsnr_min = -2 * math.log(min_value / sigma_data)
    logsnr_max = -2 * math.log(max_value / sigma_data)
    t = stratified_uniform(shape, group=0, groups=1, dtype=dtype, device=device, generator=generator)
    logsnr = torch.lerp(shifted(t, noise_d_low, logsnr_min, logsnr_max), shifted(t, noise_d_high, logsnr_min, logsnr_max), t)
    return torch.exp(-logsnr / 2) * sigma_data


def train_add_time_ids(fps, motion_bucket_ids, noise_aug_strength, dtype, batch_size, unet=None, device=None):
    """``:1177-1220``: rows ``[fps, noise_aug_strength, motion_bucket_id]`` (the training order - the inference pipeline's is
    ``[fps, motion_bucket_id, noise_aug_strength]``), with the reference's consistency checks."""
    target = device if device is not None else "cpu"
    m = motion_bucket_ids.to(device=target) if torch.is_tensor(motion_bucket_ids) else torch.tensor(motion_bucket_ids, dtype=dtype, device=target)
    if m.dim() == 1:
        m = m.view(-1, 1)
    if m.size(0) != batch_size:
        raise ValueError("The length of motion_bucket_ids must match the batch_size.")
    ids = torch.cat([torch.tensor([fps, noise_aug_strength], dtype=dtype, device=target).repeat(batch_size, 1), m.to(dtype)], dim=1)
    if unet is not None:
        passed = unet.config.addition_time_embed_dim * ids.size(1)
        expected = unet.add_embedding.linear_1.in_features
        if expected != passed:
            raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} was created. "
                             "The model has an incorrect config. Please check `unet.config.time_embedding_type` and `text_encoder_2.config.projection_dim`.")
    return ids


def _edm_loss(pred: torch.Tensor, noisy: torch.Tensor, target: torch.Tensor, sigmas: torch.Tensor) -> torch.Tensor:
    """``pred`` ``[B, F, 4, h, w]`` as the U-Net returns it (a view of a channels-last buffer) -> per-sample losses ``[B]``."""
    B, F, Cc, h, w = pred.shape
    cl = pred.permute(0, 1, 3, 4, 2)
    if not cl.is_contiguous():
        cl = cl.contiguous()
    if cl.dtype not in (torch.float16, torch.float32):
        cl = cl.float()
    out = torch.empty(B, dtype=torch.float32, device=pred.device)
    hip.check(hip.lib().pt_edm_loss(cl.data_ptr(), 1 if cl.dtype == torch.float32 else 0, Cc, noisy.data_ptr(), target.data_ptr(),
                                    sigmas.data_ptr(), B, F, h * w, out.data_ptr(), ops._stream()), "pt_edm_loss")
    return out


@torch.no_grad()
def controlnet_training_loss(controlnet, unet, latents: torch.Tensor, encoder_hidden_states: torch.Tensor, motion_values,
                             trajectories: torch.Tensor, *, scaling_factor: float = 0.18215,
                             conditioning_dropout_prob: Optional[float] = None, use_spatial: bool = True,
                             noise: Optional[torch.Tensor] = None, sigmas: Optional[torch.Tensor] = None,
                             random_p: Optional[torch.Tensor] = None, ran_idx: Optional[int] = None,
                             generator: Optional[torch.Generator] = None) -> dict:
    """One training step's forward and loss (``:1275-1407``) for ``latents`` ``[B, F, 4, h, w]`` (VAE latents x scaling_factor),
    ``encoder_hidden_states`` ``[B, 1, D]`` (CLIP embedding of the first frame), ``motion_values`` ``[B]``, ``trajectories``
    ``[B, F, 3, H, W]`` in [-1, 1].  ``noise`` / ``sigmas`` / ``random_p`` / ``ran_idx``: the step's random draws, sampled here
    (``generator``) when not given.  Returns ``loss`` (= temporal + 0.5 spatial), ``loss_temporal``, ``loss_spatial`` as floats
    and the intermediate tensors.  ``use_spatial`` follows the reference's hard-wired ``True`` and, like its
    ``sample[ran_idx]`` indexing, needs a batch of one clip."""
    dev = unet.device
    if dev is None:
        raise RuntimeError("controlnet_training_loss: the U-Net has no weights loaded")
    lat = latents.to(device=dev, dtype=torch.float32).contiguous()
    B, F, Cz, h, w = lat.shape
    if Cz != 4:
        raise ValueError(f"latents must be [batch, frames, 4, h, w]; got {tuple(latents.shape)}")
    if use_spatial and B != 1:
        raise ValueError("the spatial loss indexes the ControlNet residuals by frame (`sample[ran_idx]`, :1396-1398): batch size 1 only")
    if noise is None:
        noise = torch.randn(lat.shape, generator=generator, dtype=torch.float32)                 # torch.randn_like(latents)
    if sigmas is None:
        sigmas = rand_cosine_interpolated([B], generator=generator)
    if conditioning_dropout_prob is not None and random_p is None:
        random_p = torch.rand(B, generator=generator)
    if ran_idx is None:
        ran_idx = int(torch.randint(0, F, (1,), generator=generator).item())
    noise = noise.to(device=dev, dtype=torch.float32).contiguous()
    sig_host = sigmas.detach().to("cpu", torch.float32).reshape(B)
    ehs = encoder_hidden_states
    mask = torch.ones(B, dtype=torch.float32)
    if conditioning_dropout_prob is not None:                                                    # :1317-1339
        p, rp = conditioning_dropout_prob, random_p.detach().to("cpu", torch.float32)
        ehs = torch.where((rp < 2 * p).reshape(B, 1, 1).to(ehs.device), torch.zeros_like(ehs), ehs)
        mask = 1 - ((rp >= p).to(torch.float32) * (rp < 3 * p).to(torch.float32))
    sig = sig_host.to(dev)
    cond_scale = (mask / scaling_factor).to(dev)
    timesteps = torch.Tensor([0.25 * s.log() for s in sig_host])                                 # :1295-1296
    noisy = torch.empty_like(lat)
    x = torch.empty((B, F, h, w, 8), dtype=torch.float16, device=dev)
    hip.check(hip.lib().pt_edm_train_input(lat.data_ptr(), noise.data_ptr(), sig.data_ptr(), cond_scale.data_ptr(), TRAIN_NOISE_AUG, B, F,
                                           h * w, noisy.data_ptr(), x.data_ptr(), ops._stream()), "pt_edm_train_input")
    inp = x.permute(0, 1, 4, 2, 3)                                                               # [B, F, 8, h, w] view
    ids = train_add_time_ids(6, motion_values, TRAIN_NOISE_AUG, torch.float32, B, unet, device=dev)
    traj = trajectories.to(dev, torch.float16)
    down, mid = controlnet(inp, timesteps, ehs, added_time_ids=ids, controlnet_cond=traj, return_dict=False)
    pred = unet(inp, timesteps, ehs, added_time_ids=ids, down_block_additional_residuals=list(down),
                mid_block_additional_residual=mid, return_dict=False)[0]
    per_sample = _edm_loss(pred, noisy, lat, sig)
    loss_t = float(per_sample.mean())
    out = dict(loss=loss_t, loss_temporal=loss_t, loss_spatial=None, model_pred=pred, inp_noisy_latents=inp, timesteps=timesteps,
               added_time_ids=ids, encoder_hidden_states=ehs, noise=noise, sigmas=sig_host, ran_idx=ran_idx)
    if use_spatial:                                                                              # :1388-1407
        pred_s = unet(inp[:, ran_idx].unsqueeze(1), timesteps, ehs, added_time_ids=ids,
                      down_block_additional_residuals=[d[ran_idx].unsqueeze(0) for d in down],
                      mid_block_additional_residual=mid[ran_idx].unsqueeze(0), return_dict=False)[0]
        ls = _edm_loss(pred_s, noisy[:, ran_idx:ran_idx + 1].contiguous(), lat[:, ran_idx:ran_idx + 1].contiguous(), sig)
        out["loss_spatial"] = float(ls.mean())
        out["loss"] = loss_t + 0.5 * out["loss_spatial"]
    return out
