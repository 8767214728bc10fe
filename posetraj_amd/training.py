"""The reference's ControlNet training step on the MI355X path (SURVEY 8f4;
``/root/reference/scripts/train_svd_traj_VIPSeg_14.py:1264-1425``): sigma sampling, noising + EDM preconditioning
(``pt_edm_train_input``), the training-time ``added_time_ids``, conditioning dropout, ControlNet + frozen U-Net forward, the
sigma-weighted MSE and the single-frame "spatial" loss (``pt_edm_loss``) - ``controlnet_training_loss``, the objective alone on the
inference kernels - and the whole step: ``ControlNetTrainer`` runs the forward on the tape of ``autodiff.py`` /
``train_graph.py``, the reverse pass through the frozen U-Net's up path into every ControlNet parameter, and ``torch.optim.AdamW``'s
update (``pt_adamw_f32``) with fp16-mixed-precision loss scaling and gradient accumulation as ``accelerate`` does them for
``start_ft.sh``.  The VAE encode (``tensor_to_vae_latent``, ``:495-503``) and the CLIP embedding of the first frame are this
package's own models; their outputs are the step's inputs.  The camera twin (``controlnet_sdv_cam``, ``scripts/train_svd_traj_VIPSeg_14_cam_concat.py``: the same step with
``camera_cond`` and without the spatial loss) trains through the same class.  Not here: EMA (``--use_ema``, off in the launch scripts), the 8-bit
optimizer (bitsandbytes), gradient checkpointing (activations of one 14-frame clip fit the 288 GB many times over), the data
loader.
"""
from __future__ import annotations

import gc
import math
import time
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


def _step_inputs(dev, latents, encoder_hidden_states, motion_values, trajectories, unet, *, scaling_factor, conditioning_dropout_prob, use_spatial,
                 noise, sigmas, random_p, ran_idx, generator, kernel: bool = True, lazy_traj: bool = False):
    """``:1275-1345``: the step's random draws (sampled here when not given), noising, EDM preconditioning, conditioning dropout,
    ``added_time_ids``.  Returns a dict of device tensors; ``x`` is the network input channels-last ``[B, F, h, w, 8]``."""
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
    ids = train_add_time_ids(6, motion_values, TRAIN_NOISE_AUG, torch.float32, B, unet, device=dev)
    I = dict(lat=lat, noise=noise, sig=sig, cond_scale=cond_scale, sig_host=sig_host, timesteps=timesteps, ids=ids, ehs=ehs,
             ran_idx=ran_idx, dims=(B, F, h, w))
    if lazy_traj:                                             # the largest input (15 MB at 320 x 576; a host-side cast + copy of ~6 ms): the trainer
        I["traj_src"] = trajectories                          # stages it AFTER it has launched the frozen encoder, which does not read it
    else:
        I["traj"] = trajectories.to(dev, torch.float16)
    if kernel:
        _edm_train_input(I)
    return I


def _edm_train_input(I: dict) -> dict:
    """The device half of ``_step_inputs``: ``noisy`` latents and the network input ``x`` (EDM preconditioning + image-latent concat)."""
    lat, (B, F, h, w) = I["lat"], I["dims"]
    I["noisy"] = torch.empty_like(lat)
    I["x"] = torch.empty((B, F, h, w, 8), dtype=torch.float16, device=lat.device)
    hip.check(hip.lib().pt_edm_train_input(lat.data_ptr(), I["noise"].data_ptr(), I["sig"].data_ptr(), I["cond_scale"].data_ptr(), TRAIN_NOISE_AUG,
                                           B, F, h * w, I["noisy"].data_ptr(), I["x"].data_ptr(), ops._stream()), "pt_edm_train_input")
    return I


@torch.no_grad()
def controlnet_training_loss(controlnet, unet, latents: torch.Tensor, encoder_hidden_states: torch.Tensor, motion_values,
                             trajectories: torch.Tensor, *, scaling_factor: float = 0.18215,
                             conditioning_dropout_prob: Optional[float] = None, use_spatial: bool = True,
                             noise: Optional[torch.Tensor] = None, sigmas: Optional[torch.Tensor] = None,
                             random_p: Optional[torch.Tensor] = None, ran_idx: Optional[int] = None,
                             generator: Optional[torch.Generator] = None) -> dict:
    """One training step's forward and loss (``:1275-1407``) on the inference kernels, for ``latents`` ``[B, F, 4, h, w]`` (VAE
    latents x scaling_factor), ``encoder_hidden_states`` ``[B, 1, D]`` (CLIP embedding of the first frame), ``motion_values``
    ``[B]``, ``trajectories`` ``[B, F, 3, H, W]`` in [-1, 1].  ``noise`` / ``sigmas`` / ``random_p`` / ``ran_idx``: the step's random
    draws, sampled here (``generator``) when not given.  Returns ``loss`` (= temporal + 0.5 spatial), ``loss_temporal``,
    ``loss_spatial`` as floats and the intermediate tensors.  ``use_spatial`` follows the reference's hard-wired ``True`` and, like
    its ``sample[ran_idx]`` indexing, needs a batch of one clip.  For the step WITH its backward see ``ControlNetTrainer``."""
    dev = unet.device
    if dev is None:
        raise RuntimeError("controlnet_training_loss: the U-Net has no weights loaded")
    I = _step_inputs(dev, latents, encoder_hidden_states, motion_values, trajectories, unet, scaling_factor=scaling_factor,
                     conditioning_dropout_prob=conditioning_dropout_prob, use_spatial=use_spatial, noise=noise, sigmas=sigmas,
                     random_p=random_p, ran_idx=ran_idx, generator=generator)
    lat, noisy, sig, timesteps, ehs, ids, ran_idx = I["lat"], I["noisy"], I["sig"], I["timesteps"], I["ehs"], I["ids"], I["ran_idx"]
    inp = I["x"].permute(0, 1, 4, 2, 3)                                                          # [B, F, 8, h, w] view
    down, mid = controlnet(inp, timesteps, ehs, added_time_ids=ids, controlnet_cond=I["traj"], return_dict=False)
    pred = unet(inp, timesteps, ehs, added_time_ids=ids, down_block_additional_residuals=list(down),
                mid_block_additional_residual=mid, return_dict=False)[0]
    per_sample = _edm_loss(pred, noisy, lat, sig)
    loss_t = float(per_sample.mean())
    out = dict(loss=loss_t, loss_temporal=loss_t, loss_spatial=None, model_pred=pred, inp_noisy_latents=inp, timesteps=timesteps,
               added_time_ids=ids, encoder_hidden_states=ehs, noise=I["noise"], sigmas=I["sig_host"], ran_idx=ran_idx)
    if use_spatial:                                                                              # :1388-1407
        pred_s = unet(inp[:, ran_idx].unsqueeze(1), timesteps, ehs, added_time_ids=ids,
                      down_block_additional_residuals=[d[ran_idx].unsqueeze(0) for d in down],
                      mid_block_additional_residual=mid[ran_idx].unsqueeze(0), return_dict=False)[0]
        ls = _edm_loss(pred_s, noisy[:, ran_idx:ran_idx + 1].contiguous(), lat[:, ran_idx:ran_idx + 1].contiguous(), sig)
        out["loss_spatial"] = float(ls.mean())
        out["loss"] = loss_t + 0.5 * out["loss_spatial"]
    return out


import contextlib

_null = contextlib.nullcontext


class ControlNetTrainer:
    """The reference's optimisation step for the ControlNet (``:1040-1076`` set-up, ``:1264-1425`` step): fp32 master parameters in
    one flat buffer, the frozen U-Net, ``torch.optim.AdamW`` semantics (``lr``, ``betas``, ``weight_decay``, ``eps`` = the script's
    ``--learning_rate`` / ``--adam_*`` arguments), ``accelerate``'s fp16 handling (loss scaled by ``loss_scale`` before the reverse
    pass, the step skipped and the scale halved when a gradient overflowed, doubled after ``growth_interval`` clean steps) and
    gradient accumulation (``--gradient_accumulation_steps``: each micro-batch's loss is divided by it).  Under
    ``torch.distributed`` (one process per GPU) the ranks train data-parallel like accelerate's DDP: parameters broadcast from rank 0,
    gradients averaged by bucketed all-reduces overlapped with the reverse pass (``grad_sync.py``).

    ``wgrad_stream`` (default on): every layer's weight / bias gradient runs on a second HIP stream beside its data gradient -
    the low-resolution layers fill a fraction of the chip each, and nobody waits for a weight gradient before the optimizer.
    ``spatial_stream`` (default on): the one-frame decoder pass of the spatial loss runs, forward and backward, on a third stream
    beside the temporal pass (its ~1 500 launches over a few hundred rows each are latency, not throughput).

    ``lr_scheduler``: the multiplier of ``learning_rate`` as a function of the optimizer steps taken so far, e.g.
    ``train_state.get_scheduler(args.lr_scheduler, args.lr_warmup_steps, args.max_train_steps)`` (``:1109-1114``, ``:1424``); None = constant.
    ``save_state`` / ``load_state`` / ``save_pretrained``: ``accelerator.save_state`` / ``load_state`` and ``controlnet.save_pretrained``
    (``train_state.py``).

    ``freeze_gc`` (opt-in; a process-wide side effect, undone by ``unfreeze_gc()``): ``gc.freeze()`` once the trainer is built and
    again after its first optimizer step.  A step
    allocates ~10^5 short-lived Python objects that stay alive until its reverse pass has run, which promotes them to the oldest
    generation and triggers a full collection every six or seven steps; each one walks the whole process (torch's modules, this
    package's layer and pack objects) for ~45 ms and frees nothing (profiles/r04/train_step_host_gc.txt).  Frozen, the
    long-lived objects are out of the collector's way; cyclic garbage among them would no longer be reclaimed (there is none).

    ``unet`` must have been loaded with ``keep_source=True`` (its up-path weights are re-packed for the data gradients).
    ``controlnet_state_dict``: the parameters to train, e.g. ``ControlNetSDVModel.from_unet(unet).state_dict()`` (``:935-938``)."""

    def __init__(self, controlnet_config, controlnet_state_dict, unet, *, learning_rate: float = 1e-4, adam_beta1: float = 0.9,
                 adam_beta2: float = 0.999, adam_weight_decay: float = 1e-2, adam_epsilon: float = 1e-8,
                 gradient_accumulation_steps: int = 1, loss_scale: float = 65536.0, growth_interval: int = 2000,
                 scaling_factor: float = 0.18215, conditioning_dropout_prob: Optional[float] = None, process_group=None,
                 bucket_mb: int = 256, wgrad_stream: bool = True, spatial_stream: bool = True, freeze_gc: bool = False,
                 lr_scheduler=None, use_graph: bool = False, device_scalars: Optional[bool] = None, encoder_stream: bool = True,
                 pack_stream: bool = True):
        from . import autodiff as AD
        from . import grad_sync
        from . import train_graph as TG
        dev = unet.device
        if dev is None:
            raise RuntimeError("ControlNetTrainer: the U-Net has no weights loaded")
        cfg = dict(controlnet_config)
        self.unet, self.device, self.config = unet, dev, cfg
        self.params = AD.ParamStore(controlnet_state_dict, dev)
        self.controlnet = TG.ControlNetGraph(self.params, cfg)
        self._frozen = AD.FrozenParams({k: v for k, v in unet.state_dict().items() if k.startswith(("up_blocks.", "conv_norm_out.", "conv_out."))}, dev)
        self.decoder = TG.UNetDecoderGraph(self._frozen, dict(unet.config))
        self.lr, self.betas, self.weight_decay, self.eps = learning_rate, (adam_beta1, adam_beta2), adam_weight_decay, adam_epsilon
        self.accumulation, self.loss_scale, self.growth_interval = int(gradient_accumulation_steps), float(loss_scale), int(growth_interval)
        self.scaling_factor, self.dropout = scaling_factor, conditioning_dropout_prob
        self.optimizer_steps, self.skipped_steps, self._micro, self._clean = 0, 0, 0, 0
        self.lr_scheduler, self.last_lr = lr_scheduler, learning_rate
        self._accum_scale = None
        self.wgrad_stream, self._side = bool(wgrad_stream), None
        self.spatial_stream, self._sp_stream, self._packs_built = bool(spatial_stream), None, False
        self.encoder_stream, self._enc_stream = bool(encoder_stream), None
        self.pack_stream, self._pk_stream, self._packs_pending = bool(pack_stream), None, False
        # data parallel (accelerate's DDP, :1117-1119): one process per GPU, every rank its own clips; all ranks start from rank
        # 0's parameters and average their gradients - bucketed all-reduces over the flat buffer, overlapped with the backward
        grad_sync.broadcast_parameters(self.params.flat, process_group)
        self.buckets = grad_sync.GradientBuckets(self.params.grad, self.params.spans(), process_group, bucket_bytes=bucket_mb << 20)
        self.world = self.buckets.world
        if self.world > 1:
            self.params.on_grad_ready = self.buckets.mark_ready
        # use_graph: forward + backward of a step replayed as ONE hipGraph (see loss_and_grads); device_scalars (implied by it): the
        # AlphaBlender weights of the trainable network are read from device memory instead of being passed as host floats
        self.use_graph = bool(use_graph)
        if self.use_graph and (self.world > 1 or self.accumulation != 1):
            raise ValueError("ControlNetTrainer(use_graph=True): one rank and gradient_accumulation_steps = 1 (the captured step holds no "
                             "collective and ends with complete gradients)")
        self.params.device_scalars = self.use_graph if device_scalars is None else bool(device_scalars)
        self._graphs, self._static, self._graph_pool = {}, None, None
        self._freeze_gc = 2 if freeze_gc else 0
        self._freeze()

    def _freeze(self) -> None:
        if self._freeze_gc > 0:
            self._freeze_gc -= 1
            gc.collect()
            gc.freeze()

    def unfreeze_gc(self) -> None:
        """Hand the objects ``freeze_gc`` moved to the permanent generation back to the collector (process-wide, like the freeze)."""
        gc.unfreeze()

    # -- forward + backward of one micro-batch; gradients ACCUMULATE in self.params.grad: after a whole cycle they are
    #    loss_scale x the gradient of the mean micro-batch loss
    def loss_and_grads(self, latents, encoder_hidden_states, motion_values, trajectories, *, use_spatial: bool = True, noise=None,
                       sigmas=None, random_p=None, ran_idx=None, generator=None, camera_cond=None) -> dict:
        """``camera_cond`` ``[1, F, 12]``: the camera twin's per-frame R|T (``scripts/train_svd_traj_VIPSeg_14_cam_concat.py:1393,1409``;
        that script has no spatial loss: ``use_spatial=False``)."""
        unet, dev = self.unet, self.device
        t_host = time.perf_counter()
        I = _step_inputs(dev, latents, encoder_hidden_states, motion_values, trajectories, unet, scaling_factor=self.scaling_factor,
                         conditioning_dropout_prob=self.dropout, use_spatial=use_spatial, noise=noise, sigmas=sigmas, random_p=random_p,
                         ran_idx=ran_idx, generator=generator, kernel=False, lazy_traj=True)
        B, F, h, w = I["dims"]
        if B != 1:
            raise ValueError("ControlNetTrainer takes one clip per step (the reference trains with --per_gpu_batch_size=1)")
        if self._accum_scale is None:                    # the loss scale of this accumulation cycle (it only changes between cycles)
            self._accum_scale = self.loss_scale
        scale = self._accum_scale / self.accumulation    # accelerate divides each micro-batch's loss by the number of micro-batches
        cam = None
        if camera_cond is not None:
            if not self.config.get("camera"):
                raise ValueError("camera_cond given to a ControlNet without the camera branch (config camera=False)")
            cam = torch.as_tensor(camera_cond, dtype=torch.float32)[0]
        ran_idx = I["ran_idx"]
        I["timesteps"] = I["timesteps"].to(device=dev, dtype=torch.float32)
        I["ehs"] = I["ehs"].to(device=dev, dtype=torch.float32)
        if cam is not None:
            I["cam"] = cam.to(device=dev, dtype=torch.float32).contiguous()
        # the step as a hipGraph: everything from the EDM preconditioning kernel to the last gradient is captured ONCE per (spatial frame
        # index, loss scale, input shapes) over static input buffers and replayed - ~6 000 launches, 60-70 ms of host enqueue time,
        # become one.  The first steps run eagerly: they build the packs, the side streams and take the first optimizer step.
        graphed = self.use_graph and self._packs_built and (self.optimizer_steps + self.skipped_steps) >= 1 and self._micro == 0 and self.accumulation == 1
        if graphed:
            I["traj"] = I.pop("traj_src").to(dev, torch.float16)              # (a replay reads every input from its static buffer)
            lt, ls = self._replay(I, ran_idx, bool(use_spatial), float(scale))
        else:
            lt, ls = self._run(I, ran_idx, bool(use_spatial), float(scale))
        self._micro += 1
        self._packs_built = self._packs_built or bool(use_spatial)
        t_host = time.perf_counter() - t_host                 # everything is enqueued; reading the loss is the first wait for the device
        loss_t = float(lt)
        out = dict(loss=loss_t, loss_temporal=loss_t, loss_spatial=None, grad_scale=scale, ran_idx=ran_idx, sigmas=I["sig_host"],
                   host_enqueue_ms=1000.0 * t_host, graph_replay=bool(graphed))
        if ls is not None:
            out["loss_spatial"] = float(ls)
            out["loss"] = loss_t + 0.5 * out["loss_spatial"]
        return out

    _STATIC_KEYS = ("lat", "noise", "sig", "cond_scale", "timesteps", "ids", "ehs", "traj", "cam")

    def _replay(self, I: dict, ran_idx: int, use_spatial: bool, scale: float):
        """Copy the step's inputs into the static buffers of its graph (captured on first use) and replay it."""
        names = [k for k in self._STATIC_KEYS if k in I]
        key = (ran_idx, use_spatial, scale) + tuple((k, tuple(I[k].shape), I[k].dtype) for k in names)
        if self._static is None or any(k not in self._static or self._static[k].shape != I[k].shape or self._static[k].dtype != I[k].dtype for k in names):
            self._static = {k: torch.empty_like(I[k]) for k in names}          # (new shapes: the graphs over the old buffers are dropped)
            self._graphs = {}
        for k in names:
            self._static[k].copy_(I[k], non_blocking=True)
        entry = self._graphs.get(key)
        if entry is None:
            S = dict(I)
            S.update(self._static)
            self.params.version += 1                      # every version-keyed refresh (fp16 mirror, weight packs) is part of the capture: the
            torch.cuda.synchronize()                      # replayed step always follows an optimizer step, whatever this one follows
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool):
                lt, ls = self._run(S, ran_idx, use_spatial, scale)
            if self._graph_pool is None:
                self._graph_pool = g.pool()               # one memory pool for all graphs of this trainer: they never run concurrently
            entry = self._graphs[key] = (g, lt, ls)
        entry[0].replay()
        return entry[1], entry[2]

    def warm_graphs(self, latents, encoder_hidden_states, motion_values, trajectories, *, use_spatial: bool = True, **draws) -> int:
        """Capture the step's graph for every spatial frame index (``ran_idx`` is drawn per step: 0 .. F - 1) WITHOUT training on them: the
        gradients each capture's first replay accumulates are discarded.  Returns the number of graphs held.  (A capture costs about
        one eager step; a training run that does not call this captures each index the first time it is drawn.)"""
        if not self.use_graph:
            return 0
        F = latents.shape[1]
        micro, accum = self._micro, self._accum_scale
        for r in range(F):
            d = dict(draws)
            d["ran_idx"] = r
            self.loss_and_grads(latents, encoder_hidden_states, motion_values, trajectories, use_spatial=use_spatial, **d)
            self._micro, self._accum_scale = micro, accum
        torch.cuda.synchronize()
        self.params.zero_grad()
        return len(self._graphs)

    def _run(self, I: dict, ran_idx: int, use_spatial: bool, scale: float):
        """Forward and backward of one micro-batch from device-resident inputs ``I`` - the part of the step that a hipGraph can hold: no
        host value that changes between steps is passed to a launch (the AlphaBlender weights come from ``ParamStore.alphas``, the loss
        scale and the spatial frame index are part of the graph's key).  Returns the two loss tensors (spatial: None)."""
        from . import autodiff as AD
        unet, dev = self.unet, self.device
        B, F, h, w = I["dims"]
        if self.params.device_scalars:
            self.params.refresh_alphas()
        _edm_train_input(I)
        lat, noisy, sig, timesteps, ids = I["lat"], I["noisy"], I["sig"], I["timesteps"], I["ids"]
        ehs16 = I["ehs"].to(dtype=torch.float16).reshape(1, -1).contiguous()
        inp = I["x"].permute(0, 1, 4, 2, 3)
        L = hip.lib()
        # three tapes: the ControlNet's forward, the decoder pass of the temporal loss and the one-frame decoder pass of the spatial
        # loss.  The spatial pass is 1/14 of the work in ~1 500 launches over 180 ... 2 880 rows - latency, not throughput - and
        # depends on the temporal pass nowhere between the ControlNet's outputs and the join of the residual gradients: it runs on
        # a stream of its own beside the temporal pass, forward and backward.
        tape_cn, tape, tape_sp = AD.Tape(), AD.Tape(), AD.Tape()
        cam = I.get("cam")
        main = torch.cuda.current_stream()
        # the frozen U-Net's encoder half (inference kernels, no tape) depends on the ControlNet nowhere: it runs on a stream of its own
        # beside the ControlNet's forward, as in the inference loop (round 6; the step is device-bound - as a hipGraph it takes what it
        # takes eagerly - and at 320 x 576 no launch of the two encoders fills the chip)
        state = None
        if self.encoder_stream:
            if self._enc_stream is None:
                self._enc_stream = torch.cuda.Stream()
            self._enc_stream.wait_stream(main)
            with torch.cuda.stream(self._enc_stream), torch.no_grad():
                state = unet._encode(inp, timesteps, I["ehs"], ids)
        if "traj" not in I:                                   # staged here, behind the encoder's launches: the device is busy meanwhile
            I["traj"] = I.pop("traj_src").to(dev, torch.float16)
        if self._packs_pending:                               # the packs re-written behind the last optimizer step (their own stream)
            main.wait_stream(self._pk_stream)
            self._packs_pending = False
        outs, mid = self.controlnet.run(tape_cn, I["x"].view(F * h * w, 8), (F, h, w), timesteps, ehs16, ids, I["traj"][0], camera_cond=cam)
        with torch.no_grad():
            emb_silu = unet.time.run(timesteps, ids, 1)

        def loss_of(p: AD.Var, nz, tg, frames, weight):
            out = torch.empty(1, dtype=torch.float32, device=dev)
            hip.check(L.pt_edm_loss(p.v.data_ptr(), 0, p.v.shape[-1], nz.data_ptr(), tg.data_ptr(), sig.data_ptr(), 1, frames, h * w,
                                    out.data_ptr(), ops._stream()), "pt_edm_loss")
            g = torch.empty((p.v.shape[0], 8), dtype=torch.float16, device=dev)
            hip.check(L.pt_edm_loss_bwd(p.v.data_ptr(), 0, p.v.shape[-1], nz.data_ptr(), tg.data_ptr(), sig.data_ptr(), 1, frames, h * w,
                                        float(scale * weight), g.data_ptr(), ops._stream()), "pt_edm_loss_bwd")
            p.g = g
            return out

        ls, deferred = None, []
        spatial_stream = None
        if use_spatial:                                                                          # :1388-1407
            # the frozen decoder's weight packs (forward, transposed, split) are built lazily by whichever pass touches a layer
            # first - device kernels on THAT pass's stream - and both passes read them: until every pack exists (the first step with
            # a spatial pass) the two passes share the main stream, so that no pack is read across streams without an event
            if self.spatial_stream and self._packs_built:
                if self._sp_stream is None:
                    self._sp_stream = torch.cuda.Stream()
                spatial_stream = self._sp_stream
                spatial_stream.wait_stream(main)                                                # ControlNet outputs, emb_silu, inputs
            with torch.cuda.stream(spatial_stream) if spatial_stream is not None else _null():
                with torch.no_grad():
                    state_s = unet._encode(inp[:, ran_idx].unsqueeze(1), timesteps, I["ehs"], ids)
                mult_s = unet._multiplicity(state_s, len(outs))
                res_s = [AD.rows(tape_sp, o, ran_idx * (o.v.shape[0] // F), (ran_idx + 1) * (o.v.shape[0] // F), defer=deferred) for o in outs]
                mid_s = AD.rows(tape_sp, mid, ran_idx * (mid.v.shape[0] // F), (ran_idx + 1) * (mid.v.shape[0] // F), defer=deferred)
                pred_s = self.decoder.run(tape_sp, state_s, mult_s, res_s, mid_s, emb_silu, ehs16)
                ls = loss_of(pred_s, noisy[:, ran_idx:ran_idx + 1].contiguous(), lat[:, ran_idx:ran_idx + 1].contiguous(), 1, 0.5)
        if state is None:
            with torch.no_grad():
                state = unet._encode(inp, timesteps, I["ehs"], ids)
        else:
            main.wait_stream(self._enc_stream)
            for tns in [state["x"], state["ctx"].temb, state["ctx"].xattn] + list(state["skips"]):
                if tns is not None:
                    tns.record_stream(main)                   # allocated on the encoder's stream, consumed on this one
                    if getattr(tns, "lo", None) is not None:
                        tns.lo.record_stream(main)
        mult = unet._multiplicity(state, len(outs))
        pred = self.decoder.run(tape, state, mult, outs, mid, emb_silu, ehs16)
        lt = loss_of(pred, noisy, lat, F, 1.0)
        sync = self._micro + 1 >= self.accumulation          # inside an accumulation cycle only the last micro-batch synchronises
        if sync:
            self.buckets.begin(signature=(bool(use_spatial), cam is not None))
        if self.wgrad_stream and self._side is None:
            self._side = torch.cuda.Stream()
        AD.WGRAD_STREAM = self._side if self.wgrad_stream else None
        self.buckets.streams = [main, AD.WGRAD_STREAM]
        try:
            if use_spatial:                                   # the frozen decoder has no weight gradients: nothing of this pass leaves its stream
                with torch.cuda.stream(spatial_stream) if spatial_stream is not None else _null():
                    tape_sp.backward()
            tape.backward()                                   # temporal pass: the decoder's reverse, on the main stream
            if spatial_stream is not None:
                main.wait_stream(spatial_stream)
            for x_, r0_, r1_, dy_ in deferred:                # join: the spatial pass's residual gradients land in frame ran_idx's rows
                AD.add_rows(x_, r0_, r1_, dy_)
            tape_cn.backward()
        finally:
            AD.WGRAD_STREAM = None
        if self.wgrad_stream:
            main.wait_stream(self._side)                      # every weight gradient is in before anyone reads the buffer
        if sync:
            self.buckets.finish()                             # the gradients are now the SUM over ranks
        return lt, ls

    def gradients(self) -> dict:
        """The accumulated gradients, un-scaled (and averaged over the ranks once synchronised), by parameter name (fp32)."""
        inv = 1.0 / ((self._accum_scale or 1.0) * (self.world if self._micro >= self.accumulation else 1))
        return {k: self.params.gradient(k) * inv for k in self.params.names}

    def grad_norm(self) -> float:
        """Global L2 norm of the (un-scaled) gradients; ``inf`` / ``nan`` when an fp16 gradient overflowed."""
        acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        hip.check(hip.lib().pt_sumsq_f32(self.params.grad.data_ptr(), self.params.numel, acc.data_ptr(), ops._stream()), "pt_sumsq_f32")
        return math.sqrt(float(acc)) / ((self._accum_scale or 1.0) * self.world) if math.isfinite(float(acc)) else float(acc)

    def optimizer_step(self, grad_norm: Optional[float] = None) -> bool:
        """``optimizer.step(); optimizer.zero_grad()`` (``:1423-1425``) under the GradScaler's rules.  Returns whether the
        parameters moved.  ``grad_norm``: the value of ``grad_norm()`` if the caller already has it."""
        if self._accum_scale is None:
            raise RuntimeError("ControlNetTrainer.optimizer_step: no gradients accumulated since the last step")
        norm = self.grad_norm() if grad_norm is None else grad_norm
        took = math.isfinite(norm)
        if took:
            if self.lr_scheduler is not None:            # LambdaLR: the rate of step k (0-based) is base x lambda(k); a skipped step
                self.last_lr = self.lr * float(self.lr_scheduler(self.optimizer_steps))          # does not advance the schedule
            else:
                self.last_lr = self.lr
            self.optimizer_steps += 1
            P = self.params
            # AdamW + the fp16 mirror of the new parameters + the gradient's zeroing in ONE pass over the buffers
            hip.check(hip.lib().pt_adamw_fused_f32(P.flat.data_ptr(), P.grad.data_ptr(), P.exp_avg.data_ptr(), P.exp_avg_sq.data_ptr(), P.numel,
                                                   self.last_lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.optimizer_steps,
                                                   1.0 / (self._accum_scale * self.world), P.flat16.data_ptr(), 1, ops._stream()),
                      "pt_adamw_fused_f32")
            P.version += 1
            P._mirror_version = P.version                 # (half_view() need not cast the buffer again)
            if self.pack_stream and self._packs_built and not self.use_graph:
                # the new weights' fp16 packs, all layers at once, on a stream of their own behind AdamW (the next step's ControlNet
                # forward waits for it; the frozen encoder, the input staging and the host's bookkeeping do not)
                if self._pk_stream is None:
                    self._pk_stream = torch.cuda.Stream()
                self._pk_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._pk_stream):
                    P.refresh_packs()
                self._packs_pending = True
            self._clean += 1
            if self._clean >= self.growth_interval:
                self.loss_scale, self._clean = self.loss_scale * 2.0, 0
        else:
            self.skipped_steps += 1
            self.loss_scale, self._clean = self.loss_scale * 0.5, 0
        if not took:
            self.params.zero_grad()
        self._micro, self._accum_scale = 0, None
        self._freeze()                                   # the first step built the fp16 packs, the streams, the gradient-producing set
        return took

    def step(self, latents, encoder_hidden_states, motion_values, trajectories, **draws) -> dict:
        """One iteration of the training loop's body: forward, backward and - every ``gradient_accumulation_steps`` calls - the
        optimizer step."""
        out = self.loss_and_grads(latents, encoder_hidden_states, motion_values, trajectories, **draws)
        out["stepped"] = None
        if self._micro >= self.accumulation:
            out["grad_norm"] = self.grad_norm()
            out["stepped"] = self.optimizer_step(out["grad_norm"])
        return out

    def state_dict(self) -> dict:
        """The ControlNet's current fp32 parameters (``controlnet.save_pretrained`` of ``:1440-1470`` writes these)."""
        return self.params.state_dict()

    def save_pretrained(self, path: str) -> None:
        """``controlnet.save_pretrained(path)``: ``ControlNetSDVModel.from_pretrained(path)`` reads it back."""
        from . import train_state
        train_state.save_controlnet(self, path)

    def save_state(self, output_dir: str) -> None:
        """``accelerator.save_state(output_dir)`` (``:1464-1466``): parameters, AdamW moments, step count, loss-scale state."""
        from . import train_state
        train_state.save_state(self, output_dir)

    def load_state(self, input_dir: str) -> dict:
        """``accelerator.load_state(input_dir)`` (``:1241``)."""
        from . import train_state
        return train_state.load_state(self, input_dir)
