"""Oracle restatement of the FORWARD half of the reference's ControlNet training step
(``/root/reference/scripts/train_svd_traj_VIPSeg_14.py:1264-1414``): sigma sampling (``stratified_uniform`` /
``rand_cosine_interpolated``, ``:273-318``, constants ``:314-319``), noising, EDM preconditioning, the training-time
``_get_add_time_ids`` (``:1177-1220`` - ``[fps, noise_aug, motion_bucket]``, NOT the inference order), conditioning dropout
(``:1317-1339``), ControlNet + frozen U-Net forward, the sigma-weighted MSE (``:1373-1384``) and the single-frame "spatial" loss
(``:1388-1407``); ``training_step_grads`` adds ``accelerator.backward(loss)`` (``:1414``) as torch autograd over the same
modules - every ControlNet parameter's gradient - and the optimizer is ``torch.optim.AdamW`` itself (``:1051``), which the
tests instantiate directly.  EMA and the VAE / CLIP stages in front of the step are outside this restatement.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  PINNED: ``tests/golden/train.npz`` holds what the script's own statements
(extracted at generation time, executed over the reference networks on the oracle's blocks) produced - the sampler's draws, the
network input, ``timesteps``, ``added_time_ids``, the conditioning after dropout, both losses - with every random draw of the
step stored as an input; ``tests/golden/train_grads.npz`` holds the parameter gradients and the parameters after
``optimizer.step()`` that the script's statements produced (``accelerator.backward`` = ``loss.backward``, fp32).
"""
from __future__ import annotations

import math

import torch

# :314-319
MIN_VALUE, MAX_VALUE, IMAGE_D, NOISE_D_LOW, NOISE_D_HIGH, SIGMA_DATA = 0.002, 700, 64, 32, 64, 0.5
TRAIN_NOISE_AUG = 0.02          # :1288


def stratified_uniform(shape, group=0, groups=1, dtype=None, device=None, u=None):
    """``:273-282``; ``u``: the uniform draws (``torch.rand(shape)``) when the caller owns the randomness."""
    if groups <= 0:
        raise ValueError(f"groups must be positive, got {groups}")
    if group < 0 or group >= groups:
        raise ValueError(f"group must be in [0, {groups})")
    n = shape[-1] * groups
    offsets = torch.arange(group, n, groups, dtype=dtype, device=device)
    if u is None:
        u = torch.rand(shape, dtype=dtype, device=device)
    return (offsets + u) / n


def rand_cosine_interpolated(shape, image_d=IMAGE_D, noise_d_low=NOISE_D_LOW, noise_d_high=NOISE_D_HIGH, sigma_data=SIGMA_DATA,
                             min_value=MIN_VALUE, max_value=MAX_VALUE, device="cpu", dtype=torch.float32, u=None):
    """``:285-312``: sigmas from the interpolated, shifted cosine log-SNR schedule ("simple diffusion")."""
    def cosine(t, lo, hi):
        t_min = math.atan(math.exp(-0.5 * hi))
        t_max = math.atan(math.exp(-0.5 * lo))
        return -2 * torch.log(torch.tan(t_min + t * (t_max - t_min)))

    def shifted(t, noise_d, lo, hi):
        shift = 2 * math.log(noise_d / image_d)
        return cosine(t, lo - shift, hi - shift) + shift

    logsnr_min = -2 * math.log(min_value / sigma_data)
    logsnr_max = -2 * math.log(max_value / sigma_data)
    t = stratified_uniform(shape, group=0, groups=1, dtype=dtype, device=device, u=u)
    logsnr = torch.lerp(shifted(t, noise_d_low, logsnr_min, logsnr_max), shifted(t, noise_d_high, logsnr_min, logsnr_max), t)
    return torch.exp(-logsnr / 2) * sigma_data


def train_add_time_ids(fps, motion_bucket_ids, noise_aug_strength, dtype, batch_size):
    """``:1177-1220``: rows ``[fps, noise_aug_strength, motion_bucket_id]``."""
    m = torch.as_tensor(motion_bucket_ids, dtype=dtype)
    if m.dim() == 1:
        m = m.view(-1, 1)
    if m.size(0) != batch_size:
        raise ValueError("The length of motion_bucket_ids must match the batch_size.")
    return torch.cat([torch.tensor([fps, noise_aug_strength], dtype=dtype).repeat(batch_size, 1), m], dim=1)


def training_inputs(latents, noise, sigmas, encoder_hidden_states, scaling_factor, random_p=None, conditioning_dropout_prob=None):
    """``:1282-1345``: -> (inp_noisy_latents [B,F,8,h,w], noisy_latents, timesteps [B], encoder_hidden_states after dropout)."""
    bsz = latents.shape[0]
    s = sigmas.reshape(bsz, 1, 1, 1, 1)
    cond = (latents + noise * TRAIN_NOISE_AUG)[:, 0] / scaling_factor
    noisy = latents + noise * s
    timesteps = torch.Tensor([0.25 * sg.log() for sg in sigmas])
    inp = noisy / ((s ** 2 + 1) ** 0.5)
    ehs = encoder_hidden_states
    if conditioning_dropout_prob is not None:
        p = conditioning_dropout_prob
        ehs = torch.where((random_p < 2 * p).reshape(bsz, 1, 1), torch.zeros_like(ehs), ehs)
        mask = 1 - ((random_p >= p).to(cond.dtype) * (random_p < 3 * p).to(cond.dtype))
        cond = mask.reshape(bsz, 1, 1, 1) * cond
    cond = cond.unsqueeze(1).repeat(1, noisy.shape[1], 1, 1, 1)
    return torch.cat([inp, cond], dim=2), noisy, timesteps, ehs


def edm_loss(model_pred, noisy_latents, target, sigmas):
    """``:1372-1384``: c_out = -s / sqrt(s^2 + 1), c_skip = 1 / (s^2 + 1), weight (1 + s^2) / s^2; mean per sample, then over
    the batch.  ``sigmas`` broadcastable to the tensors."""
    c_out = -sigmas / ((sigmas ** 2 + 1) ** 0.5)
    c_skip = 1 / (sigmas ** 2 + 1)
    den = model_pred * c_out + c_skip * noisy_latents
    w = (1 + sigmas ** 2) * (sigmas ** -2.0)
    return torch.mean((w.float() * (den.float() - target.float()) ** 2).reshape(target.shape[0], -1), dim=1).mean()


def training_loss(controlnet, unet, latents, noise, sigmas, encoder_hidden_states, motion_values, trajectories, scaling_factor,
                  random_p=None, conditioning_dropout_prob=None, ran_idx=0, use_spatial=True, camera_cond=None):
    """The step's forward and loss (``:1275-1407``) for given draws.  Returns a dict with ``loss`` (= temporal + 0.5 spatial),
    ``loss_temporal``, ``loss_spatial``, ``model_pred`` and the intermediate inputs."""
    bsz = latents.shape[0]
    inp, noisy, timesteps, ehs = training_inputs(latents, noise, sigmas, encoder_hidden_states, scaling_factor, random_p,
                                                 conditioning_dropout_prob)
    ids = train_add_time_ids(6, motion_values, TRAIN_NOISE_AUG, ehs.dtype, bsz)
    cam = {} if camera_cond is None else {"camera_cond": camera_cond}        # the camera twin's step (..._cam_concat.py:1393,1409)
    down, mid = controlnet(inp, timesteps, ehs, added_time_ids=ids, controlnet_cond=trajectories, return_dict=False, **cam)
    pred = unet(inp, timesteps, ehs, added_time_ids=ids, down_block_additional_residuals=list(down),
                mid_block_additional_residual=mid, return_dict=False)[0]
    s5 = sigmas.reshape(bsz, 1, 1, 1, 1)
    loss_t = edm_loss(pred, noisy, latents, s5)
    out = dict(inp_noisy_latents=inp, timesteps=timesteps, added_time_ids=ids, encoder_hidden_states=ehs, model_pred=pred,
               loss_temporal=loss_t, loss_spatial=None, loss=loss_t)
    if use_spatial:                                              # :1388-1407 (written for a batch of one clip: the residuals are
        pred_s = unet(inp[:, ran_idx].unsqueeze(1), timesteps, ehs, added_time_ids=ids,      # indexed by FRAME, `sample[ran_idx]`)
                      down_block_additional_residuals=[d[ran_idx].unsqueeze(0) for d in down],
                      mid_block_additional_residual=mid[ran_idx].unsqueeze(0), return_dict=False)[0]
        c_out = -s5 / ((s5 ** 2 + 1) ** 0.5)
        c_skip = 1 / (s5 ** 2 + 1)
        den = pred_s[:, 0] * c_out + c_skip * noisy[:, ran_idx]          # broadcasts to [B, 1, 4, h, w] like the reference
        w = (1 + s5 ** 2) * (s5 ** -2.0)
        loss_s = torch.mean((w.float() * (den.float() - latents[:, ran_idx].float()) ** 2).reshape(bsz, -1), dim=1).mean()
        out.update(loss_spatial=loss_s, loss=loss_t + loss_s * 0.5)
    return out


def training_step_grads(controlnet, unet, latents, noise, sigmas, encoder_hidden_states, motion_values, trajectories, scaling_factor,
                        random_p=None, conditioning_dropout_prob=None, ran_idx=0, use_spatial=True, camera_cond=None):
    """``:1275-1414``: the step's forward, loss and ``loss.backward()`` with the U-Net frozen and the ControlNet trainable
    (``:953,1053``).  Returns ``training_loss``'s dict (detached) plus ``grads``: name -> gradient of every ControlNet
    parameter (zeros where autograd produced none)."""
    for p in unet.parameters():
        p.requires_grad_(False)
    for p in controlnet.parameters():
        p.requires_grad_(True)
        p.grad = None
    out = training_loss(controlnet, unet, latents, noise, sigmas, encoder_hidden_states, motion_values, trajectories, scaling_factor,
                        random_p, conditioning_dropout_prob, ran_idx, use_spatial, camera_cond)
    out["loss"].backward()
    grads = {k: (torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()) for k, p in controlnet.named_parameters()}
    res = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
    res["grads"] = grads
    for p in controlnet.parameters():
        p.grad = None
    return res
