"""MI355X-native ``StableVideoDiffusionPipelineControlNet`` - the denoise loop of
``/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:316-599`` (and of the ``_cam`` twin:
``camera_cond`` argument, ``..._cam.py:321,505-509,549``) behind the reference's ``__call__`` signature.

Scope: steps 4-8 of ``__call__`` - timesteps, latents, control tensor, guidance ramp, hard-coded
micro-conditioning, the 25-iteration loop - plus the stages either side of it: ``_encode_image`` (the
anti-aliased resize runs in ``pt_resize_antialias_f32``, the CLIP vision tower is ``posetraj_amd.clip_vision`` or any
callable with the ``transformers`` interface), ``_encode_vae_image`` with the noise augmentation of the first frame, and after
the loop ``decode_latents`` (``:225-251``) + ``tensor2vid`` (``:70-83``) over ``posetraj_amd.autoencoder_kl_temporal_decoder``,
so ``output_type`` "pil" / "np" / "pt" / "latent" all work as in the reference.  Without ``image_encoder`` / ``vae`` the
outputs of those stages can be passed in (``image_embeddings`` / ``image_latents``) and ``output_type`` must be "latent".

Per step the loop launches: one fused prologue (CFG duplicate + 1/sqrt(sigma^2+1) + image-latent concat, written
channels-last), ControlNet, U-Net, one fused epilogue (per-frame guidance + Euler update on fp32 latents).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import torch

from . import ops
from .controlnet_sdv import ControlNetSDVModel
from .modeling import BaseOutput
from .scheduling_euler_discrete_karras_fix import PREDICTION_TYPES, EulerDiscreteScheduler
from .unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel


class StableVideoDiffusionPipelineOutput(BaseOutput):
    """``frames`` (``pipeline...:85-96``)."""


def _append_dims(x, target_dims):
    """``pipeline...:62-67``."""
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def _get_add_time_ids(noise_aug_strength, dtype, batch_size, fps=4, motion_bucket_id=128, unet=None):
    """Module-level helper of the reference (``pipeline...:37-59``) including its config check."""
    add_time_ids = [fps, motion_bucket_id, noise_aug_strength]
    passed = unet.config.addition_time_embed_dim * len(add_time_ids)
    expected = unet.add_embedding.linear_1.in_features
    if expected != passed:
        raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} was created. The model has an incorrect config. Please check `unet.config.time_embedding_type` and `text_encoder_2.config.projection_dim`.")
    return torch.tensor([add_time_ids], dtype=dtype)


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """``diffusers.utils.torch_utils.randn_tensor`` (called at ``pipeline...:292,451``): N(0,1) of ``shape`` on ``device``.
    A generator on another device type (the usual seeded CPU generator) samples there and the result is moved; a list of
    generators samples one batch entry each."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    batch = shape[0]

    def one(shp, g):
        gdev = g.device if g is not None else device
        if gdev.type != device.type and gdev.type != "cpu":
            raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gdev.type}.")
        on = device if gdev.type == device.type else torch.device("cpu")
        return torch.randn(tuple(shp), generator=g, device=on, dtype=dtype).to(device)

    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        return torch.cat([one((1,) + tuple(shape[1:]), generator[i]) for i in range(batch)], dim=0)
    return one(shape, generator)


def postprocess(image: torch.Tensor, output_type: str = "pil"):
    """``VaeImageProcessor.postprocess`` (diffusers 0.24.0) for one clip ``[F, 3, H, W]`` in [-1, 1]:
    ``(x / 2 + 0.5).clamp(0, 1)`` then "pt": that tensor; "np": ``[F, H, W, 3]`` float32 on the host; "pil": a list of
    ``PIL.Image`` from ``round(255 x)``; "latent": the input.  Runs in ``pt_frames_postprocess``."""
    if output_type == "latent":
        return image
    if output_type not in ("pt", "np", "pil"):
        raise ValueError(f"output_type {output_type!r} is not supported; choose one of 'pil', 'np', 'pt', 'latent'")
    x = ops.frames_postprocess(image.to(torch.float32), output_type)
    if output_type == "pt":
        return x
    arr = x.cpu().numpy()
    if output_type == "np":
        return arr
    import PIL.Image
    return [PIL.Image.fromarray(a) for a in arr]


def tensor2vid(video: torch.Tensor, processor=None, output_type="np"):
    """``pipeline...:70-83``: ``video`` ``[batch, 3, frames, H, W]`` -> a list with one ``postprocess``-ed clip per batch entry."""
    post = processor.postprocess if processor is not None else postprocess
    outputs = []
    for batch_idx in range(video.shape[0]):
        outputs.append(post(video[batch_idx].permute(1, 0, 2, 3), output_type))
    return outputs


class _ImageProcessor:
    """The two ``VaeImageProcessor`` entry points the reference pipeline calls (``:143,454,588``)."""

    def __init__(self, vae_scale_factor=8):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image, height=None, width=None):
        return StableVideoDiffusionPipelineControlNet.preprocess_condition(image, height, width)

    @staticmethod
    def postprocess(image, output_type="pil"):
        return postprocess(image, output_type)


class StableVideoDiffusionPipelineControlNet:
    model_cpu_offload_seq = "image_encoder->unet->vae"
    _callback_tensor_inputs = ["latents"]

    def __init__(self, vae=None, image_encoder=None, unet: UNetSpatioTemporalConditionControlNetModel = None,
                 controlnet: ControlNetSDVModel = None, scheduler: EulerDiscreteScheduler = None, feature_extractor=None):
        self.vae, self.image_encoder, self.unet = vae, image_encoder, unet
        self.controlnet, self.scheduler, self.feature_extractor = controlnet, scheduler, feature_extractor
        self.vae_scale_factor = 8 if vae is None else 2 ** (len(vae.config.block_out_channels) - 1)
        self.image_processor = _ImageProcessor(self.vae_scale_factor)
        self._guidance_scale = None
        self._num_timesteps = 0
        self._graph_state = None                    # captured hipGraph of the per-iteration networks (denoise(use_graph=True))
        self._side_stream = None                    # second HIP stream of denoise(overlap_streams=True)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, controlnet: ControlNetSDVModel = None,
                        unet: UNetSpatioTemporalConditionControlNetModel = None, scheduler: EulerDiscreteScheduler = None,
                        vae=None, image_encoder=None, device="cuda", variant: Optional[str] = None, **kw):
        """The construction the reference's callers use (``scripts/run_inference_vipseg_json_repro.py:335-339``):
        ``from_pretrained(svd_dir, controlnet=controlnet, unet=unet)``.  Reads the diffusers directory layout of the SVD
        checkpoint: ``scheduler/scheduler_config.json`` for the sampler, ``unet/`` when no ``unet`` is passed, and - when the
        sub-directories exist and no module is passed - ``vae/`` (``AutoencoderKLTemporalDecoder``) and ``image_encoder/``
        (``CLIPVisionModelWithProjection``, transformers' ``config.json`` + ``model[.variant].safetensors``)."""
        from .autoencoder_kl_temporal_decoder import AutoencoderKLTemporalDecoder
        from .clip_vision import CLIPVisionModelWithProjection
        root = pretrained_model_name_or_path
        if scheduler is None:
            path = os.path.join(root, "scheduler", "scheduler_config.json")
            if os.path.exists(path):
                with open(path) as f:
                    cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
                scheduler = EulerDiscreteScheduler(**cfg)
            else:
                from .scheduling_euler_discrete_karras_fix import SVD_SCHEDULER_CONFIG
                scheduler = EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG)
        if unet is None:
            unet = UNetSpatioTemporalConditionControlNetModel.from_pretrained(root, subfolder="unet", device=device, variant=variant)
        if controlnet is None:
            raise ValueError("pass `controlnet=` (the reference always does: ControlNetSDVModel.from_pretrained(ckpt, subfolder='controlnet'))")
        if vae is None and os.path.exists(os.path.join(root, "vae", "config.json")):
            vae = AutoencoderKLTemporalDecoder.from_pretrained(root, subfolder="vae", device=device, variant=variant)
        if image_encoder is None and os.path.exists(os.path.join(root, "image_encoder", "config.json")):
            image_encoder = CLIPVisionModelWithProjection.from_pretrained(root, subfolder="image_encoder", device=device, variant=variant)
        return cls(vae=vae, image_encoder=image_encoder, unet=unet, controlnet=controlnet, scheduler=scheduler)

    @staticmethod
    def preprocess_condition(controlnet_condition, height: int, width: int) -> torch.Tensor:
        """``self.image_processor.preprocess(controlnet_condition, height=, width=)`` of ``pipeline...:500``
        (diffusers 0.24.0 ``VaeImageProcessor.preprocess`` [UNVERIFIED-MEMORY: diffusers is not in the tree]): a list of
        PIL images / ``[F, H, W, 3]`` [0,1] float (or uint8) arrays is resized to ``width x height`` (PIL lanczos), scaled to
        [0, 1], laid out ``[F, 3, H, W]`` and normalised to [-1, 1]; a ``[F, 3, H, W]`` tensor is resized with
        ``interpolate`` when its size differs and normalised only if it has no negative value (a tensor that already is
        in [-1, 1] passes through).  Host-side data formatting: a few MB once per clip."""
        c = controlnet_condition
        if torch.is_tensor(c):
            x = c.float()
            if x.dim() == 3:
                x = x.unsqueeze(0)
            if tuple(x.shape[-2:]) != (height, width):
                x = torch.nn.functional.interpolate(x, size=(height, width))
            return x if float(x.min()) < 0 else 2.0 * x - 1.0
        frames = list(c) if isinstance(c, (list, tuple)) else [c]
        out = []
        for fr in frames:
            if hasattr(fr, "resize") and hasattr(fr, "convert"):                       # PIL.Image
                import PIL.Image
                a = np.asarray(fr.convert("RGB").resize((width, height), resample=PIL.Image.LANCZOS), dtype=np.float32) / 255.0
            else:                                                                      # arrays are [0, 1] floats in diffusers;
                a = np.asarray(fr)                                                     # uint8 pixels are accepted as a convenience
                a = a.astype(np.float32) / 255.0 if a.dtype == np.uint8 else a.astype(np.float32)
                if a.shape[:2] != (height, width):
                    a = torch.nn.functional.interpolate(torch.from_numpy(a).permute(2, 0, 1)[None], size=(height, width))[0].permute(1, 2, 0).numpy()
            out.append(torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1))
        return 2.0 * torch.stack(out) - 1.0

    # -- no-op compatible surface of DiffusionPipeline used by the reference's callers
    def to(self, *a, **k):
        return self

    def enable_model_cpu_offload(self, *a, **k):
        pass

    def set_progress_bar_config(self, **k):
        pass

    def maybe_free_model_hooks(self):
        pass

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def num_timesteps(self):
        return self._num_timesteps

    def check_inputs(self, image, height, width):
        """``pipeline...:253-265``."""
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    def prepare_latents(self, batch_size, num_frames, num_channels_latents, height, width, dtype, device, generator,
                        latents=None):
        """``pipeline...:267-299``: N(0,1) * init_noise_sigma."""
        shape = (batch_size, num_frames, num_channels_latents // 2, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma.to(latents.device)

    # ------------------------------------------------------------------------------------------ pre-loop conditioning
    def _encode_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        """``pipeline...:145-172``: resize to 224 x 224 with ``_resize_with_antialiasing`` (no CLIP mean / std
        normalisation: the reference skips the feature extractor), ``image_encoder(image).image_embeds``,
        ``[B, 1, D]``, repeated per prompt; the CFG-negative half is zeros, concatenated in front."""
        if not torch.is_tensor(image):
            image = self._to_unit_tensor(image)               # pil_to_numpy + numpy_to_pt: [B, 3, H, W] in [0, 1]
        image = ops.resize_with_antialiasing(image.to(device=device, dtype=torch.float32), (224, 224))
        dtype = getattr(self.image_encoder, "dtype", None)
        if dtype is None and hasattr(self.image_encoder, "parameters"):
            dtype = next(self.image_encoder.parameters()).dtype
        out = self.image_encoder(image.to(dtype=dtype or torch.float32))
        image_embeddings = (out.image_embeds if hasattr(out, "image_embeds") else out).unsqueeze(1)
        bs_embed, seq_len, _ = image_embeddings.shape
        image_embeddings = image_embeddings.repeat(1, num_videos_per_prompt, 1).view(bs_embed * num_videos_per_prompt, seq_len, -1)
        if do_classifier_free_guidance:
            image_embeddings = torch.cat([torch.zeros_like(image_embeddings), image_embeddings])
        return image_embeddings

    def _encode_vae_image(self, image: torch.Tensor, device, num_videos_per_prompt, do_classifier_free_guidance):
        """``pipeline...:174-195``: ``vae.encode(image).latent_dist.mode()`` - NOT multiplied by ``scaling_factor`` - zeros
        for the CFG-negative half, repeated per prompt."""
        image_latents = self.vae.encode(image.to(device=device)).latent_dist.mode()
        if do_classifier_free_guidance:
            image_latents = torch.cat([torch.zeros_like(image_latents), image_latents])
        return image_latents.repeat(num_videos_per_prompt, 1, 1, 1)

    @staticmethod
    def _to_unit_tensor(image) -> torch.Tensor:
        """``VaeImageProcessor.pil_to_numpy`` + ``numpy_to_pt`` (``pipeline...:148-150``): PIL image(s) / ``[H, W, 3]`` arrays
        -> ``[B, 3, H, W]`` fp32 in [0, 1]."""
        frames = list(image) if isinstance(image, (list, tuple)) else [image]
        out = []
        for fr in frames:
            a = np.asarray(fr.convert("RGB") if hasattr(fr, "convert") else fr).astype(np.float32) / 255.0   # pil_to_numpy: always
            out.append(torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1))
        return torch.stack(out)

    # ------------------------------------------------------------------------------------------ after the loop
    def decode_latents(self, latents: torch.Tensor, num_frames: int, decode_chunk_size: int = 14) -> torch.Tensor:
        """``pipeline...:225-251``: ``[B, F, 4, h, w]`` -> fp32 ``[B, 3, F, 8h, 8w]``.  ``1 / scaling_factor``, then
        ``decode_chunk_size`` frames per ``vae.decode`` call with ``num_frames`` = the frames in that call (each call is one
        clip for the decoder's temporal layers; a chunk may span two clips when B > 1), results in frame order.  With this
        package's VAE every call writes its frames straight into the final buffer (no ``torch.cat`` pass)."""
        import inspect
        from .autoencoder_kl_temporal_decoder import AutoencoderKLTemporalDecoder
        vae, inv = self.vae, 1 / self.vae.config.scaling_factor
        flat = latents.flatten(0, 1)                                               # [B*F, 4, h, w]
        total = flat.shape[0]
        takes_num_frames = "num_frames" in inspect.signature(vae.forward).parameters
        spans = [(i, min(i + decode_chunk_size, total)) for i in range(0, total, decode_chunk_size)]
        if isinstance(vae, AutoencoderKLTemporalDecoder):
            z = ops.scale(flat if flat.dtype in (torch.float16, torch.float32) else flat.float(), inv)
            up = 2 ** (len(vae.config.block_out_channels) - 1)
            decoded = torch.empty((total, vae.config.out_channels, z.shape[2] * up, z.shape[3] * up), dtype=torch.float32, device=z.device)
            for lo, hi in spans:                                                   # each call writes its frames in place
                vae.decode(z[lo:hi], num_frames=hi - lo, out=decoded[lo:hi])
        else:                                                                      # any module with the diffusers call shape
            z = inv * flat
            parts = [vae.decode(z[lo:hi], **({"num_frames": hi - lo} if takes_num_frames else {})).sample for lo, hi in spans]
            decoded = torch.cat(parts, dim=0)
        video = decoded.reshape(-1, num_frames, *decoded.shape[1:]).permute(0, 2, 1, 3, 4)    # [B, 3, F, H, W]
        return video.float()

    # ------------------------------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def denoise(self, latents: torch.Tensor, image_latents: torch.Tensor, image_embeddings: torch.Tensor,
                controlnet_condition: torch.Tensor, num_inference_steps: int = 25, min_guidance_scale: float = 1.0,
                max_guidance_scale: float = 3.0, controlnet_cond_scale: float = 1.0,
                camera_cond: Optional[torch.Tensor] = None, callback_on_step_end: Optional[Callable] = None,
                callback_on_step_end_tensor_inputs: List[str] = ["latents"], use_graph: bool = False,
                overlap_streams: bool = False, _networks=None) -> torch.Tensor:
        """``pipeline...:481-583``.  ``latents`` ``[Bc, F, 4, h, w]`` already scaled by ``init_noise_sigma``;
        ``image_latents`` ``[2*Bc, 4, h, w]`` (uncond halves first, one frame - it is repeated over frames, ``:466``);
        ``image_embeddings`` ``[2*Bc, 1, D]``; ``controlnet_condition`` ``[2*Bc, F, 3, H, W]`` in [-1, 1].
        Returns the denoised latents in the dtype of ``latents``."""
        if max_guidance_scale <= 1.0:
            # The reference's non-CFG branch (:438, :532) cannot run: latents / embeddings stay [Bc, ...] but the control
            # maps (:501-503) and added_time_ids (:521) are doubled unconditionally, so ControlNetSDVModel.forward reshapes
            # the time ids to [Bc, 2 * 3 * D] and add_embedding's Linear raises (reference run: tests/golden/loop.npz,
            # `base_noncfg_error`).  Same exception type and message here.
            c = self.unet.config
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({latents.shape[0]}x{2 * 3 * c.addition_time_embed_dim} and "
                               f"{c.projection_class_embeddings_input_dim}x{c.block_out_channels[0] * 4})")
        dev = latents.device
        out_dtype = latents.dtype
        Bc, F = latents.shape[:2]
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        timesteps = self.scheduler.timesteps
        # per-frame guidance ramp (:506-511)
        g = torch.linspace(min_guidance_scale, max_guidance_scale, F).unsqueeze(0).to(dev, out_dtype).repeat(Bc, 1)
        self._guidance_scale = _append_dims(g, latents.ndim)
        guidance = g.to(torch.float32).contiguous()
        # micro-conditioning is hard-coded to fps 6 / bucket 128 / aug 0.02 whatever the caller asked (:513-523, Q4)
        ids = _get_add_time_ids(0.02, torch.float32, Bc, 6, 128, unet=self.unet)
        added_time_ids = ids.repeat(2 * Bc, 1).to(dev)                                     # torch.cat([ids] * 2)
        x = latents.to(torch.float32).contiguous().clone()
        il = image_latents.to(dev, torch.float16).contiguous()
        emb = image_embeddings.to(dev, torch.float16).contiguous()
        cond = controlnet_condition.to(dev, torch.float16)
        cam = None if camera_cond is None else camera_cond.to(dev, torch.float16)
        ptype = PREDICTION_TYPES[self.scheduler.config.prediction_type]
        sig = self.scheduler._sigmas_host
        self._num_timesteps = len(timesteps)
        self.scheduler._step_index = None
        def networks(sample, t, emb_, cond_, cam_):
            """ControlNet + U-Net of one iteration -> fp32 channels-last prediction [2Bc, F, h, w, 4].  The ControlNet
            residuals are accumulated straight into the U-Net's skips by the zero-convs' epilogues (no residual tensors,
            no separate add passes); with ``overlap_streams`` the two encoders run concurrently."""
            if _networks is not None:        # tools/variants/split_cfg.py: another schedule of the same two networks (A/B only)
                return _networks(self, sample, t, emb_, cond_, cam_, added_time_ids, controlnet_cond_scale)
            main = torch.cuda.current_stream(dev)
            if overlap_streams:
                # The U-Net's encoder half does not depend on the ControlNet (its outputs are added to the skips and to
                # the mid block's output afterwards): run it on a second HIP stream while the ControlNet runs on this
                # one.  Same kernels, same results; the two streams' workgroups fill each other's tail rounds and
                # memory-bound phases.  (Inside a hipGraph capture this becomes two branches of the graph.)
                if self._side_stream is None or self._side_stream.device != dev:
                    self._side_stream = torch.cuda.Stream(device=dev)
                side = self._side_stream
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    enc = self.unet._encode(sample, t, emb_, added_time_ids)
            else:
                enc = self.unet._encode(sample, t, emb_, added_time_ids)
            taps, xm = self.controlnet._features(sample, t, emb_, added_time_ids, cond_,
                                                 cam_ if self.controlnet.config.camera else None)
            if overlap_streams:
                main.wait_stream(side)
                for tns in [enc["x"], enc["ctx"].temb, enc["ctx"].xattn] + list(enc["skips"]):
                    if tns is not None:
                        tns.record_stream(main)               # allocated on the side stream, consumed on this one
                        if getattr(tns, "lo", None) is not None:
                            tns.lo.record_stream(main)        # ... and so is the low half of a wide-stream tensor
            self.controlnet._accumulate_into(taps, xm, controlnet_cond_scale, enc["skips"],
                                             self.unet._multiplicity(enc, len(taps)), enc["x"])
            pred = self.unet._decode(enc, None, None, return_dict=False, residuals_added=True, out_f32=True)[0]
            pred_cl = pred.permute(0, 1, 3, 4, 2)                                          # [2Bc, F, h, w, 4] contiguous
            return pred_cl if pred_cl.is_contiguous() else pred_cl.contiguous()

        # The ~2 500 launches of ControlNet + U-Net are captured ONCE into a hipGraph over static input buffers and
        # replayed per iteration (and for later clips of the same geometry); the two fused loop kernels around them
        # carry the per-step scalars and stay eager.  The condition encoder is not in the graph: it runs once per clip
        # and refreshes its cached output in place.
        gs = None
        if use_graph:
            key = (Bc, F, tuple(x.shape[3:]), tuple(cond.shape), None if cam is None else tuple(cam.shape),
                   float(controlnet_cond_scale), id(self.unet), id(self.controlnet), self.unet._generation,
                   self.controlnet._generation, bool(overlap_streams), getattr(_networks, '__name__', None))
            gs = self._graph_state if self._graph_state is not None and self._graph_state["key"] == key else None
            if gs is None:
                gs = dict(key=key, xin=torch.empty((2 * Bc, F, x.shape[3], x.shape[4], 8), dtype=torch.float16, device=dev),
                          t=torch.zeros(1, dtype=torch.float32, device=dev), emb=emb.clone(), cond=cond.clone(),
                          cam=None if cam is None else cam.clone(), ids=added_time_ids.clone())
            else:
                gs["emb"].copy_(emb); gs["cond"].copy_(cond); gs["ids"].copy_(added_time_ids)
                if cam is not None:
                    gs["cam"].copy_(cam)
            added_time_ids = gs["ids"]
            # once per clip: the condition encoder, eagerly (the in-place edits above bumped the tensors' versions),
            # into a buffer this graph state owns: the captured graph reads that address whatever the ControlNet's
            # cache holds after other (eager) calls
            gs["cond_embed"] = self.controlnet._cond_embedding(gs["cond"], gs["cam"] if self.controlnet.config.camera else None,
                                                              out=gs.get("cond_embed"))
            if "graph" not in gs:
                ops.scale_concat_input(x, il, sig[0], out=gs["xin"])
                gs["t"].fill_(float(self.scheduler._timesteps_host[0]))
                warm = torch.cuda.Stream(device=dev)
                warm.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(warm):                # warm-up: weight-only caches, allocator sizing
                    networks(gs["xin"].permute(0, 1, 4, 2, 3), gs["t"], gs["emb"], gs["cond"], gs["cam"])
                torch.cuda.current_stream(dev).wait_stream(warm)
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    gs["pred"] = networks(gs["xin"].permute(0, 1, 4, 2, 3), gs["t"], gs["emb"], gs["cond"], gs["cam"])
                gs["graph"] = g_
                self._graph_state = gs

        for i in range(len(timesteps)):
            t = self.scheduler._timesteps_host[i]
            if self.scheduler._step_index is None:
                self.scheduler._init_step_index(t)
            k = self.scheduler._step_index
            if gs is not None:
                ops.scale_concat_input(x, il, sig[k], out=gs["xin"])
                gs["t"].fill_(float(t))
                gs["graph"].replay()
                pred_cl = gs["pred"]
            else:
                xin = ops.scale_concat_input(x, il, sig[k])                                # [2Bc, F, h, w, 8]
                pred_cl = networks(xin.permute(0, 1, 4, 2, 3), t, emb, cond, cam)          # [2Bc, F, 8, h, w] view
            ops.cfg_euler_step(pred_cl, guidance, sig[k], sig[k + 1], ptype, x)
            self.scheduler._step_index += 1
            self.scheduler.is_scale_input_called = True
            if callback_on_step_end is not None:
                cb_latents = x.to(out_dtype)
                outs = callback_on_step_end(self, i, t, {"latents": cb_latents})
                new = outs.pop("latents", cb_latents)
                # always take what the callback hands back - it may have edited `latents` in place and returned the same
                # object.  For fp16 latents this rounds the state through fp16 once per step, as the reference does.
                if new is not x:
                    x = new.to(torch.float32).contiguous().clone()
        return x.to(out_dtype)

    @torch.no_grad()
    def __call__(self, image=None, controlnet_condition: torch.FloatTensor = None, height: int = 576, width: int = 1024,
                 num_frames: Optional[int] = None, num_inference_steps: int = 25, min_guidance_scale: float = 1.0,
                 max_guidance_scale: float = 3.0, fps: int = 7, motion_bucket_id: int = 127,
                 noise_aug_strength: float = 0.02, decode_chunk_size: Optional[int] = None,
                 num_videos_per_prompt: Optional[int] = 1, generator=None, latents: Optional[torch.FloatTensor] = None,
                 output_type: Optional[str] = "pil", callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], return_dict: bool = True,
                 controlnet_cond_scale=1.0, batch_size=1, camera_cond=None,
                 image_embeddings: Optional[torch.Tensor] = None, image_latents: Optional[torch.Tensor] = None,
                 use_graph: bool = True, overlap_streams: bool = True):
        """Same signature as the reference (``pipeline...:316-340``; ``camera_cond`` from the ``_cam`` twin) plus
        ``image_embeddings`` ``[2,1,D]`` / ``image_latents`` ``[2,4,h,w]`` (the outputs of the CLIP / VAE-encode stages,
        ``:441,457``, for callers that have them already) and ``use_graph`` / ``overlap_streams`` (the loop as a replayed
        hipGraph with the ControlNet and the U-Net encoder on two streams - the configuration ``bench.py`` measures;
        bit-identical to eager launches, ``tests/test_model_gpu.py``).  ``fps``, ``motion_bucket_id`` and
        ``noise_aug_strength`` do not reach the U-Net in the reference either (Q4)."""
        num_frames = num_frames if num_frames is not None else self.unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames                # :422
        self.check_inputs(image, height, width)
        if output_type != "latent" and self.vae is None:
            raise NotImplementedError(f"output_type={output_type!r} needs a `vae` (posetraj_amd.AutoencoderKLTemporalDecoder); "
                                      "construct the pipeline with vae=, or ask for output_type='latent'")
        dev = self.unet.device
        do_cfg = max_guidance_scale > 1.0                                                   # :438
        if image_embeddings is None:                                                        # :441
            if self.image_encoder is None:
                raise NotImplementedError("no `image_encoder` was given to the pipeline: pass `image_embeddings` [2,1,D], or "
                                          "construct the pipeline with image_encoder= (posetraj_amd.CLIPVisionModelWithProjection)")
            image_embeddings = self._encode_image(image, dev, num_videos_per_prompt, do_cfg)
        # :454-463 - an fp16 VAE with force_upcast is run in fp32 around encode(): .to(dtype=torch.float32) switches this package's
        # VAE to its fp32 encoder (csrc/vae_f32.hip), .to(dtype=torch.float16) back
        needs_upcasting = self.vae is not None and getattr(self.vae, "dtype", None) == torch.float16 and \
            bool(getattr(self.vae.config, "force_upcast", False))
        if image_latents is None:                                                           # :449-462
            if self.vae is None:
                raise NotImplementedError("no `vae` was given to the pipeline: pass `image_latents` [2,4,h,w], or construct the "
                                          "pipeline with vae= (posetraj_amd.AutoencoderKLTemporalDecoder)")
            img = self.preprocess_condition(image, height, width)                            # image_processor.preprocess -> [-1, 1]
            noise = randn_tensor(img.shape, generator=generator, device=img.device, dtype=img.dtype)
            img = img + noise_aug_strength * noise
            if needs_upcasting:
                self.vae.to(dtype=torch.float32)
            else:
                img = img.to(getattr(self.vae, "dtype", None) or img.dtype)
            try:
                image_latents = self._encode_vae_image(img, dev, num_videos_per_prompt, do_cfg).to(image_embeddings.dtype)
            finally:                                                                         # (an encode that throws leaves the VAE as it was)
                if needs_upcasting:
                    self.vae.to(dtype=torch.float16)
        self.scheduler.set_timesteps(num_inference_steps, device=dev)                       # :482, before init_noise_sigma is read (:298)
        lat = self.prepare_latents(batch_size * num_videos_per_prompt, num_frames, self.unet.config.in_channels, height,
                                   width, image_embeddings.dtype, dev, generator, latents)
        cond = self.preprocess_condition(controlnet_condition, height, width)               # :500
        cond = torch.cat([cond.unsqueeze(0)] * 2)                                           # :501-503 (Q5)
        cam = None
        if camera_cond is not None:
            cam = torch.as_tensor(camera_cond, dtype=torch.float32).unsqueeze(0)
            cam = torch.cat([cam] * 2)
        latents = self.denoise(lat, image_latents, image_embeddings, cond, num_inference_steps, min_guidance_scale,
                               max_guidance_scale, controlnet_cond_scale, cam, callback_on_step_end,
                               callback_on_step_end_tensor_inputs, use_graph=use_graph, overlap_streams=overlap_streams)
        if not output_type == "latent":                                                     # :585-592
            if needs_upcasting:
                self.vae.to(dtype=torch.float16)
            frames = self.decode_latents(latents, num_frames, decode_chunk_size)
            frames = tensor2vid(frames, self.image_processor, output_type=output_type)
        else:
            frames = latents
        self.maybe_free_model_hooks()
        if not return_dict:
            return frames
        return StableVideoDiffusionPipelineOutput(frames=frames)


# BASELINE.json names the class this way; the reference's name is the one above (pipeline...:99)
StableVideoDiffusionControlNetPipeline = StableVideoDiffusionPipelineControlNet
