#!/usr/bin/env python3
"""Generate the golden fixtures under ``tests/golden/`` by running the REFERENCE's own code.

Runs only in the build container (needs ``/root/reference``); nothing here is used at test time.  The
reference's Python never leaves that container - only the ``.npz`` data written by this script does.

How the reference is made importable
------------------------------------
Every hot-path module of the reference imports ``diffusers`` (0.24.0), which is neither vendored in the
reference nor installed here.  This script registers an in-memory stand-in for exactly the symbols those import
lines name:

* arithmetic-free plumbing, written here: ``ConfigMixin`` / ``register_to_config`` (config is registered
  before ``__init__`` runs, like diffusers, so the scheduler's early ``self.use_karras_sigmas`` read resolves
  through the config), ``BaseOutput``, ``logging``, ``randn_tensor``, ``SchedulerMixin``,
  ``KarrasDiffusionSchedulers``, ``ModelMixin``, loader mixins, ``DiffusionPipeline`` (module registry +
  progress bar), ``VaeImageProcessor`` (tensor inputs already in [-1, 1] pass through unchanged, which is what
  diffusers does for such tensors);
* the block classes (``get_down_block`` ... ``TimestepEmbedding``) are bound to ``oracle.blocks`` - so the
  "wiring" and "loop" fixtures pin the reference's *in-tree* code (forward wiring, residual multiplicity,
  zero-conv order, loop body) over the oracle's blocks, and say nothing about block internals
  (``oracle/blocks.py`` stays "parity unpinned").

Fixtures (G1-G4)
  sched.npz       EulerDiscreteScheduler tables + 2 Euler steps, 3 configs x n in {2, 25}
  cond_embed.npz  ControlNetConditioningEmbeddingSVD and _CAM outputs
  wiring.npz      ControlNetSDVModel / cam variant / UNet...ControlNetModel forwards (micro config)
  loop.npz        StableVideoDiffusionPipelineControlNet.__call__ (and the _cam twin), 2 and 3 steps
  blocks.npz      the reference's OWN executable copies of four diffusers forwards (models/modified_svd.py:50-348:
                  temporal transformer block, spatio-temporal transformer, cross-attn down / up block) run over the
                  oracle's LEAF modules (ResnetBlock2D, Attention, FeedForward, LayerNorm, AlphaBlender, Timesteps):
                  pins the composition of oracle/blocks.py rows a14 / a16 / a17; the leaves stay unpinned
  vae_io.npz      decode_latents / tensor2vid / the whole __call__ through `.frames` over the oracle's AutoencoderKLTemporalDecoder
  clip.npz        transformers.CLIPVisionModelWithProjection itself (the reference's image_encoder class) on random-init configs
  tracks.npz      the draw calls (cv2.line / circle / cvtColor arguments, in order) the reference's trajectory-map code issues
                  (scripts/run_inference_vipseg_json_repro.py:429-447, utils/dataset.py:741-766), logged by a recording cv2 stand-in
  train.npz       the forward half of the ControlNet training step (scripts/train_svd_traj_VIPSeg_14.py:1275-1414): the script's own
                  statements over the reference networks - sigma sampler, noising, preconditioning, dropout, both losses
  blocks_real.npz / vae_io_real.npz   ONLY on a machine where the real ``diffusers==0.24.0`` imports (never the build container of
                  rounds 1-6): diffusers' own TemporalBasicTransformerBlock, TransformerSpatioTemporalModel, CrossAttnDown/UpBlock
                  SpatioTemporal and AutoencoderKLTemporalDecoder, loaded with the oracle's seeded weights (state-dict names must match
                  one to one: ``strict=True``) and run on the inputs of blocks.npz / vae_io.npz.  These are the files that PIN the
                  leaves (ResnetBlock2D, Attention, FeedForward, AlphaBlender, the VAE networks); tests/test_oracle_golden.py compares
                  oracle/blocks.py and oracle/vae.py with them and skips while they are absent
  resize.npz      _resize_with_antialiasing (pipeline/pipeline_stable_video_diffusion_controlnet.py:604-712: Gaussian blur with
                  reflect padding + bicubic, align_corners=True) - the first pre-loop stage of _encode_image
"""
from __future__ import annotations

import contextlib
import enum
import functools
import inspect
import os
import sys
import types
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import blocks as OB                      # noqa: E402
from oracle import cond_embed as OC, init as OI, loop as OL, nets as ON, sched as OS, vae as OV   # noqa: E402


# ------------------------------------------------------------------------------------ stand-in plumbing
class FrozenDict(OrderedDict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    config_name = None

    def register_to_config(self, **kw):
        object.__setattr__(self, "_internal_dict", FrozenDict(kw))

    @property
    def config(self):
        return self._internal_dict

    def __getattr__(self, name):
        d = self.__dict__.get("_internal_dict")
        if d is not None and name in d:
            return d[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        params = [(n, p.default) for i, (n, p) in enumerate(inspect.signature(init).parameters.items()) if i > 0]
        cfg = {}
        for a, (n, _) in zip(args, params):
            cfg[n] = a
        for n, d in params:
            if n not in cfg:
                cfg[n] = kwargs.get(n, d)
        self.register_to_config(**cfg)
        init(self, *args, **kwargs)
    return inner


class BaseOutput(OrderedDict):
    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class _Logger:
    def warning(self, *a, **k):
        pass
    info = debug = error = warn = warning


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


class SchedulerMixin:
    pass


class KarrasDiffusionSchedulers(enum.Enum):
    EulerDiscreteScheduler = 1


class ModelMixin(nn.Module):
    @property
    def config(self):
        return self.__dict__["_internal_dict"]

    def register_to_config(self, **kw):
        object.__setattr__(self, "_internal_dict", FrozenDict(kw))

    @property
    def dtype(self):
        return next(self.parameters()).dtype


class _Empty:
    pass


class DiffusionPipeline:
    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return torch.device("cpu")

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        yield types.SimpleNamespace(update=lambda *a: None)

    def maybe_free_model_hooks(self):
        pass


class VaeImageProcessor:
    def __init__(self, vae_scale_factor=8):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image, height=None, width=None):
        assert torch.is_tensor(image) and image.min() >= -1.0 - 1e-6
        return image

    def postprocess(self, image, output_type="pil"):
        return OV.postprocess(image, output_type)          # diffusers' part of tensor2vid: restated, unpinned (oracle/vae.py)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_standins():
    _mod("diffusers")
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.utils", BaseOutput=BaseOutput, logging=_Logging, is_torch_version=lambda *a: True)
    _mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin,
         KarrasDiffusionSchedulers=KarrasDiffusionSchedulers)
    _mod("diffusers.loaders", FromOriginalControlnetMixin=type("FromOriginalControlnetMixin", (), {}),
         UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}))
    _mod("diffusers.models", UNetSpatioTemporalConditionModel=_Empty, AutoencoderKLTemporalDecoder=_Empty)
    _mod("diffusers.models.attention_processor", ADDED_KV_ATTENTION_PROCESSORS=(), CROSS_ATTENTION_PROCESSORS=(),
         AttentionProcessor=_Empty, AttnAddedKVProcessor=_Empty, AttnProcessor=_Empty)
    _mod("diffusers.models.embeddings", TextImageProjection=_Empty, TextImageTimeEmbedding=_Empty,
         TextTimeEmbedding=_Empty, TimestepEmbedding=OB.TimestepEmbedding, Timesteps=OB.Timesteps)
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.unet_3d_blocks", get_down_block=OB.get_down_block, get_up_block=OB.get_up_block,
         UNetMidBlockSpatioTemporal=lambda c, temb_channels, transformer_layers_per_block, cross_attention_dim,
         num_attention_heads: OB.UNetMidBlockSpatioTemporal(c, temb_channels, transformer_layers_per_block,
                                                            num_attention_heads, cross_attention_dim))
    _mod("diffusers.image_processor", VaeImageProcessor=VaeImageProcessor)
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline)
    try:
        import transformers  # noqa: F401
        from transformers import CLIPImageProcessor, CLIPVisionModelWithProjection  # noqa: F401
    except Exception:
        _mod("transformers", CLIPImageProcessor=_Empty, CLIPVisionModelWithProjection=_Empty)
    sys.path.insert(0, REF)


# ------------------------------------------------------------------------------------ G1 scheduler
SCHED_CFGS = {
    "svd": OS.SVD_SCHEDULER_CONFIG,
    "eps_linspace": dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                         prediction_type="epsilon", timestep_spacing="linspace"),
    "v_trailing_karras": dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                              prediction_type="v_prediction", timestep_spacing="trailing", use_karras_sigmas=True),
}


def gen_sched(out):
    from utils.scheduling_euler_discrete_karras_fix import EulerDiscreteScheduler
    for name, cfg in SCHED_CFGS.items():
        for n in (2, 25):
            s = EulerDiscreteScheduler(**cfg)
            k = f"{name}_n{n}_"
            out[k + "init_sigmas"] = s.sigmas.numpy()
            out[k + "init_timesteps"] = s.timesteps.numpy()
            out[k + "init_noise_sigma_before"] = np.float64(float(s.init_noise_sigma))
            s.set_timesteps(n)
            out[k + "sigmas"] = s.sigmas.numpy()
            out[k + "timesteps"] = s.timesteps.numpy()
            out[k + "init_noise_sigma"] = np.float64(float(s.init_noise_sigma))
            for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
                s.set_timesteps(n)
                g = torch.Generator().manual_seed(11)
                x = (torch.randn(1, 14, 4, 8, 8, generator=g) * float(s.init_noise_sigma)).to(dt)
                out[k + f"x0_{tag}"] = x.float().numpy()
                for i in range(2):
                    t = s.timesteps[i]
                    xin = s.scale_model_input(x, t)
                    out[k + f"scaled{i}_{tag}"] = xin.float().numpy()
                    mo = torch.randn(x.shape, generator=g).to(dt)
                    out[k + f"model_out{i}_{tag}"] = mo.float().numpy()
                    x = s.step(mo, t, x).prev_sample
                    out[k + f"prev{i}_{tag}"] = x.float().numpy()


def gen_add_noise(out):
    """``EulerDiscreteScheduler.add_noise`` (``utils/scheduling_...:530-553``): x + noise * sigma[index of t], per sample."""
    from utils.scheduling_euler_discrete_karras_fix import EulerDiscreteScheduler
    for name, cfg in SCHED_CFGS.items():
        s = EulerDiscreteScheduler(**cfg)
        s.set_timesteps(25)
        g = torch.Generator().manual_seed(17)
        for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
            x = torch.randn(3, 14, 4, 8, 8, generator=g).to(dt)
            n = torch.randn(3, 14, 4, 8, 8, generator=g).to(dt)
            ts = s.timesteps[torch.tensor([0, 7, 24])]
            out[f"{name}_{tag}_x"] = x.float().numpy()
            out[f"{name}_{tag}_noise"] = n.float().numpy()
            out[f"{name}_{tag}_t"] = ts.numpy()
            out[f"{name}_{tag}_y"] = s.add_noise(x, n, ts).float().numpy()


# ------------------------------------------------------------------------------------ G2 cond embed
CE_CH = (8, 16, 32, 64)
CE_OUT = 64


def gen_cond_embed(out):
    from models.controlnet_sdv import ControlNetConditioningEmbeddingSVD as RefCE
    from models.controlnet_sdv_cam_infer import ControlNetConditioningEmbeddingSVD_CAM as RefCAM
    ref = OI.seeded_init_(RefCE(CE_OUT, 3, CE_CH), seed=21).eval()
    cam = OI.seeded_init_(RefCAM(CE_OUT, 3, CE_CH), seed=22).eval()
    g = torch.Generator().manual_seed(5)
    for b in (1, 2):
        x = torch.rand(b, 14, 3, 32, 32, generator=g) * 2 - 1
        rt = torch.randn(b, 14, 12, generator=g) * 0.3
        out[f"x_b{b}"] = x.numpy()
        out[f"rt_b{b}"] = rt.numpy()
        with torch.no_grad():
            out[f"y_b{b}"] = ref(x).numpy()
            out[f"ycam_b{b}"] = cam(x, rt).numpy()
            out[f"ycam_none_b{b}"] = cam(x, None).numpy()
            out[f"ycam_zero_b{b}"] = cam(x, torch.zeros_like(rt)).numpy()


# ------------------------------------------------------------------------------------ G3 wiring
MICRO = dict(block_out_channels=(32, 32, 64, 64), num_attention_heads=(1, 1, 2, 2), cross_attention_dim=16,
             addition_time_embed_dim=8, projection_class_embeddings_input_dim=24, layers_per_block=2, num_frames=4)
MICRO_CE = (4, 8, 8, 16)


def micro_inputs(seed, b=2, f=4, h=8, w=8):
    g = torch.Generator().manual_seed(seed)
    return dict(
        sample=torch.randn(b, f, 8, h, w, generator=g),
        t=torch.tensor(0.731),
        ehs=torch.randn(b, 1, 16, generator=g),
        ids=torch.tensor([[6, 128, 0.02]] * b),
        cond=torch.rand(b, f, 3, h * 8, w * 8, generator=g) * 2 - 1,
        cam=torch.randn(b, f, 12, generator=g) * 0.3,
    )


def gen_wiring(out):
    from models.controlnet_sdv import ControlNetSDVModel as RefCN
    from models.controlnet_sdv_cam_infer import ControlNetSDVModel as RefCNCam
    from models.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel as RefUNet
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        cn = OI.seeded_init_(RefCN(**MICRO, conditioning_embedding_out_channels=MICRO_CE), seed=31).eval()
        cncam = OI.seeded_init_(RefCNCam(**MICRO, conditioning_embedding_out_channels=MICRO_CE), seed=32).eval()
        unet = OI.seeded_init_(RefUNet(**MICRO), seed=33).eval()
    out["n_keys"] = np.array([len(cn.state_dict()), len(cncam.state_dict()), len(unet.state_dict())])
    i = micro_inputs(41)
    for k, v in i.items():
        out["in_" + k] = v.numpy()
    with torch.no_grad():
        for scale, tag in ((1.0, "s1"), (0.6, "s06")):
            down, mid = cn(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False,
                           conditioning_scale=scale)
            for j, d in enumerate(down):
                out[f"cn_{tag}_down{j}"] = d.numpy()
            out[f"cn_{tag}_mid"] = mid.numpy()
        dn, md = cn(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=None, return_dict=False)
        out["cn_nocond_mid"] = md.numpy()
        dc, mc = cncam(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], camera_cond=i["cam"],
                       return_dict=False)
        for j, d in enumerate(dc):
            out[f"cncam_down{j}"] = d.numpy()
        out["cncam_mid"] = mc.numpy()
        down, mid = cn(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False)
        y = unet(i["sample"], i["t"], i["ehs"], down_block_additional_residuals=down,
                 mid_block_additional_residual=mid, added_time_ids=i["ids"], return_dict=False)[0]
        out["unet_out"] = y.numpy()
        # python float / int timesteps take the non-tensor branch (controlnet_sdv.py:552-560)
        y2 = unet(i["sample"], 0.731, i["ehs"], down_block_additional_residuals=down,
                  mid_block_additional_residual=mid, added_time_ids=i["ids"], return_dict=False)[0]
        out["unet_out_pyfloat"] = y2.numpy()
        try:
            unet(i["sample"], i["t"], i["ehs"], added_time_ids=i["ids"])
            out["unet_none_residuals_raises"] = np.array(0)
        except TypeError:
            out["unet_none_residuals_raises"] = np.array(1)
    # from_unet copies conv_in/time_embedding/down/mid but not add_embedding
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        cn2 = RefCN.from_unet(unet, conditioning_embedding_out_channels=MICRO_CE)
    sd_u, sd_c = unet.state_dict(), cn2.state_dict()
    same = [k for k in sd_c if k in sd_u and torch.equal(sd_c[k], sd_u[k])]
    out["from_unet_copied_prefixes"] = np.array(sorted({k.split(".")[0] for k in same}))
    out["from_unet_add_embedding_copied"] = np.array(int(any(k.startswith("add_embedding") for k in same)))


# ------------------------------------------------------------------------------------ G4 loop
class FakeVAE(nn.Module):
    """Stands in for AutoencoderKLTemporalDecoder: encode -> 8x average pool, 4 channels."""
    def __init__(self):
        super().__init__()
        self.p = nn.Parameter(torch.zeros(1))
        self.config = types.SimpleNamespace(block_out_channels=(1, 1, 1, 1), scaling_factor=0.18215, force_upcast=True)
        self.seen = None

    @property
    def dtype(self):
        return self.p.dtype

    def encode(self, image):
        pooled = torch.nn.functional.avg_pool2d(image, 8)
        lat = torch.cat([pooled, pooled.mean(1, keepdim=True)], dim=1)
        self.seen = lat
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(mode=lambda: lat))


class FakeCLIP(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.p = nn.Parameter(torch.zeros(1))
        self.dim = dim
        self.seen = None

    def forward(self, image):
        v = image.mean(dim=(2, 3))                                  # [1, 3]
        e = torch.sin(torch.arange(self.dim)[None, :] * 0.37 + v.sum()) * 0.8
        self.seen = e
        return types.SimpleNamespace(image_embeds=e)


def gen_loop(out):
    from models.controlnet_sdv import ControlNetSDVModel as RefCN
    from models.controlnet_sdv_cam_infer import ControlNetSDVModel as RefCNCam
    from models.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel as RefUNet
    from utils.scheduling_euler_discrete_karras_fix import EulerDiscreteScheduler
    from pipeline.pipeline_stable_video_diffusion_controlnet import StableVideoDiffusionPipelineControlNet as Pipe
    from pipeline.pipeline_stable_video_diffusion_controlnet_cam import StableVideoDiffusionPipelineControlNet as PipeCam
    f, hh, ww = 4, 64, 64
    g = torch.Generator().manual_seed(77)
    image = torch.rand(1, 3, hh, ww, generator=g) * 2 - 1
    cond = torch.rand(f, 3, hh, ww, generator=g) * 2 - 1
    latents = torch.randn(1, f, 4, hh // 8, ww // 8, generator=g)
    cam = (torch.randn(f, 12, generator=g) * 0.3)
    out["image"], out["cond"], out["latents"], out["cam"] = image.numpy(), cond.numpy(), latents.numpy(), cam.numpy()
    for variant in ("base", "cam"):
        with contextlib.redirect_stdout(open(os.devnull, "w")):
            CN = RefCN if variant == "base" else RefCNCam
            cn = OI.seeded_init_(CN(**MICRO, conditioning_embedding_out_channels=MICRO_CE),
                                 seed=31 if variant == "base" else 32).eval()
            unet = OI.seeded_init_(RefUNet(**MICRO), seed=33).eval()
        for steps, gs in ((2, (1.0, 3.0)), (3, (1.5, 2.5))):
            vae, clip = FakeVAE(), FakeCLIP(16)
            P = Pipe if variant == "base" else PipeCam
            pipe = P(vae=vae, image_encoder=clip, unet=unet, controlnet=cn,
                     scheduler=EulerDiscreteScheduler(**OS.SVD_SCHEDULER_CONFIG), feature_extractor=None)
            torch.manual_seed(123)
            kw = dict(camera_cond=cam.tolist()) if variant == "cam" else {}
            res = pipe(image, controlnet_condition=cond, height=hh, width=ww, num_frames=f,
                       num_inference_steps=steps, min_guidance_scale=gs[0], max_guidance_scale=gs[1],
                       fps=9, motion_bucket_id=33, noise_aug_strength=0.05,
                       generator=torch.Generator().manual_seed(9), latents=latents.clone(), output_type="latent",
                       return_dict=False, controlnet_cond_scale=0.8, **kw)
            k = f"{variant}_n{steps}_"
            out[k + "final"] = res.numpy()
            out[k + "vae_mode"] = vae.seen.numpy()
            out[k + "clip_embed"] = clip.seen.numpy()
            out[k + "guidance"] = pipe.guidance_scale.numpy()
        # the non-CFG branch (max_guidance_scale <= 1, pipeline...:438,532): control maps and added_time_ids are doubled
        # unconditionally (:501-503, :521) while latents / embeddings are not, so the ControlNet's add_embedding receives
        # [1, 2 * 3 * 8] instead of [1, 3 * 8] and the reference raises.  Recorded so the build mirrors the error.
        try:
            pipe(image, controlnet_condition=cond, height=hh, width=ww, num_frames=f, num_inference_steps=2,
                 min_guidance_scale=1.0, max_guidance_scale=1.0, generator=torch.Generator().manual_seed(9),
                 latents=latents.clone(), output_type="latent", return_dict=False, **kw)
            out[f"{variant}_noncfg_raises"] = np.array(0)
        except RuntimeError as e:
            out[f"{variant}_noncfg_raises"] = np.array(1)
            out[f"{variant}_noncfg_error"] = np.array(str(e).splitlines()[0])


# ------------------------------------------------------------------------------------ G5 block composition
# Geometry: CFG batch 2 x 14 frames so that the batch-interleaved time_context (modified_svd.py:152-159) is live;
# head_dim 64 so that the HIP blocks can be tested against the same fixture.
from tests.parity import BLK, blocks_inputs, blocks_modules      # noqa: E402  (the recipe the tests rebuild the modules with)


def bind_reference_forwards(root: nn.Module) -> nn.Module:
    """Re-class every composite oracle block under ``root`` so that its ``forward`` IS the reference's function from
    ``/root/reference/models/modified_svd.py`` (the leaves - resnets, Attention, FeedForward, norms - stay the oracle's).
    The only glue: the keyword ``camera_para`` (always None on this path) the reference passes to ``time_mixer``."""
    import models.modified_svd as MS

    class RefMixer(OB.AlphaBlender):
        def forward(self, x_spatial, x_temporal, image_only_indicator, camera_para=None):
            assert camera_para is None
            return OB.AlphaBlender.forward(self, x_spatial, x_temporal, image_only_indicator)

    class RefTemporal(OB.TemporalBasicTransformerBlock):
        _chunk_size, _chunk_dim = None, 0
        forward = MS.forward_TemporalBasicTransformerBlock

    class RefTransformer(OB.TransformerSpatioTemporalModel):
        gradient_checkpointing = False
        forward = MS.forward_TransformerSpatioTemporalModel

    class RefDown(OB.CrossAttnDownBlockSpatioTemporal):
        gradient_checkpointing = False
        forward = MS.forward_CrossAttnDownBlockSpatioTemporal

    class RefUp(OB.CrossAttnUpBlockSpatioTemporal):
        gradient_checkpointing = False
        forward = MS.forward_CrossAttnUpBlockSpatioTemporal

    swap = {OB.AlphaBlender: RefMixer, OB.TemporalBasicTransformerBlock: RefTemporal,
            OB.TransformerSpatioTemporalModel: RefTransformer, OB.CrossAttnDownBlockSpatioTemporal: RefDown,
            OB.CrossAttnUpBlockSpatioTemporal: RefUp}
    for m in root.modules():
        if type(m) in swap:
            m.__class__ = swap[type(m)]
    return root


def gen_blocks(out):
    mods = {k: bind_reference_forwards(m) for k, m in blocks_modules().items()}
    i = blocks_inputs()
    ind = torch.zeros(BLK["B"], BLK["F"])
    for k, v in i.items():
        out["in_" + k] = v.numpy()
    with torch.no_grad():
        out["temporal"] = mods["temporal"](i["tokens"], num_frames=BLK["F"], encoder_hidden_states=i["tctx"]).numpy()
        out["transformer"] = mods["transformer"](i["x"], encoder_hidden_states=i["ehs"], image_only_indicator=ind,
                                                 return_dict=False)[0].numpy()
        y, taps = mods["down"](i["x"], temb=i["temb"], encoder_hidden_states=i["ehs"], image_only_indicator=ind)
        out["down"] = y.numpy()
        for j, t in enumerate(taps):
            out[f"down_tap{j}"] = t.numpy()
        skips = (i["up_skip_in"], i["up_skips"][0], i["up_skips"][1])
        out["up"] = mods["up"](i["up_x"], skips, temb=i["temb"], encoder_hidden_states=i["ehs"],
                               image_only_indicator=ind).numpy()


# ------------------------------------------------------------------------------------ G6 pre-loop image resize
RESIZE_CASES = {"down_L": ((1, 3, 144, 256), (56, 56)), "down_frac": ((2, 3, 50, 70), (24, 24)), "up": ((1, 3, 20, 28), (56, 56)),
                "chw": ((3, 64, 96), (16, 24)), "clip224": ((1, 3, 160, 288), (224, 224))}


def gen_resize(out):
    from pipeline.pipeline_stable_video_diffusion_controlnet import _resize_with_antialiasing as ref_resize
    g = torch.Generator().manual_seed(91)
    for name, (shape, size) in RESIZE_CASES.items():
        x = torch.rand(shape, generator=g) * 2 - 1
        out[f"{name}_x"] = x.numpy()
        out[f"{name}_size"] = np.array(size)
        out[f"{name}_y"] = ref_resize(x, size).numpy()


# ------------------------------------------------------------------------------------ G7 VAE: decode_latents / tensor2vid / end of __call__
VAE_SEED = 51
# head_dim 64 at every level, so that the HIP networks can run the same call (tests/test_vae_gpu.py)
CALL_CFG = dict(block_out_channels=(64, 64, 128, 128), num_attention_heads=(1, 1, 2, 2), cross_attention_dim=16,
                addition_time_embed_dim=8, projection_class_embeddings_input_dim=24, layers_per_block=1, num_frames=4)
CALL_CE = (8, 8, 16, 32)


def gen_vae_io(out):
    """The reference's own ``decode_latents`` (pipeline...:225-251) and ``tensor2vid`` (:70-83) executed over the oracle's
    ``AutoencoderKLTemporalDecoder`` (oracle/vae.py: tiny config, seeded), and the reference ``__call__`` run through to
    ``.frames`` with that VAE for every ``output_type``.  Pins the in-tree code around the VAE; the VAE's own arithmetic is
    diffusers' and stays unpinned."""
    import pipeline.pipeline_stable_video_diffusion_controlnet as RP
    from models.controlnet_sdv import ControlNetSDVModel as RefCN
    from models.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel as RefUNet
    from utils.scheduling_euler_discrete_karras_fix import EulerDiscreteScheduler
    vae = OI.seeded_init_(OV.AutoencoderKLTemporalDecoder(**OV.tiny_vae_config()), seed=VAE_SEED).eval()
    for prm in vae.parameters():                            # fp16-representable weights: the GPU tests compute from the same values
        prm.data.copy_(prm.data.half().float())
    me = types.SimpleNamespace(vae=vae)
    g = torch.Generator().manual_seed(61)
    cases = {"b1f6_c14": (1, 6, 14), "b1f6_c4": (1, 6, 4), "b2f4_c3": (2, 4, 3), "b1f14_c8": (1, 14, 8)}
    with torch.no_grad():
        for name, (b, f, chunk) in cases.items():
            lat = torch.randn(b, f, 4, 4, 4, generator=g) * 0.18215 * 1.3
            out[f"dl_{name}_latents"] = lat.numpy()
            fr = RP.StableVideoDiffusionPipelineControlNet.decode_latents(me, lat, f, chunk)
            out[f"dl_{name}_frames"] = fr.numpy()
        video = torch.randn(2, 3, 5, 8, 12, generator=g) * 0.8                 # values beyond [-1, 1] exercise the clamp
        proc = VaeImageProcessor()
        out["t2v_video"] = video.numpy()
        out["t2v_np"] = np.stack(RP.tensor2vid(video, proc, output_type="np"))
        out["t2v_pt"] = torch.stack(RP.tensor2vid(video, proc, output_type="pt")).numpy()
        out["t2v_pil"] = np.stack([np.stack([np.asarray(im) for im in clip]) for clip in RP.tensor2vid(video, proc, output_type="pil")])
        # the whole reference __call__ through decode_latents + tensor2vid (small head_dim-64 nets, real-structure VAE)
        f, hh, ww = 4, 64, 64
        image = torch.rand(1, 3, hh, ww, generator=g) * 2 - 1
        cond = torch.rand(f, 3, hh, ww, generator=g) * 2 - 1
        latents = torch.randn(1, f, 4, hh // 8, ww // 8, generator=g)
        out["call_image"], out["call_cond"], out["call_latents"] = image.numpy(), cond.numpy(), latents.numpy()
        with contextlib.redirect_stdout(open(os.devnull, "w")):
            cn = OI.seeded_init_(RefCN(**CALL_CFG, conditioning_embedding_out_channels=CALL_CE), seed=31).eval()
            unet = OI.seeded_init_(RefUNet(**CALL_CFG), seed=33).eval()
            for m in (cn, unet):                            # fp16-representable weights, like the VAE's above
                for prm in m.parameters():
                    prm.data.copy_(prm.data.half().float())
        for ot in ("np", "pt", "pil", "latent"):
            clip = FakeCLIP(16)
            pipe = RP.StableVideoDiffusionPipelineControlNet(vae=vae, image_encoder=clip, unet=unet, controlnet=cn,
                                                             scheduler=EulerDiscreteScheduler(**OS.SVD_SCHEDULER_CONFIG),
                                                             feature_extractor=None)
            res = pipe(image, controlnet_condition=cond, height=hh, width=ww, num_frames=f, num_inference_steps=2,
                       decode_chunk_size=3, generator=torch.Generator().manual_seed(9), latents=latents.clone(),
                       output_type=ot, controlnet_cond_scale=0.8).frames
            if ot == "pil":
                res = np.stack([np.stack([np.asarray(im) for im in c]) for c in res])
            elif ot == "pt":
                res = torch.stack(res).numpy()
            elif ot == "np":
                res = np.stack(res)
            else:
                res = res.numpy()
            out[f"call_{ot}"] = res
            if ot == "np":
                out["call_vae_mode"] = pipe._encode_vae_image(
                    image + 0.0, "cpu", 1, False).numpy()            # the un-noised first-frame latent (encoder sanity)


# ------------------------------------------------------------------------------------ G8 CLIP vision tower (transformers itself)
CLIP_CASES = {"tiny_gelu": ("tiny", dict(hidden_act="gelu"), 2, 71), "tiny_quick": ("tiny", dict(hidden_act="quick_gelu"), 1, 72),
              "vith2": ("vith", dict(num_hidden_layers=2), 1, 73)}


def clip_case_inputs(name):
    """(config dict, uint8 images [B, 224, 224, 3], weight seed) of a case - shared with the tests."""
    from oracle import clip as OCL
    kind, over, batch, seed = CLIP_CASES[name]
    cfg = (OCL.tiny_clip_config if kind == "tiny" else OCL.vit_h_config)(**over)
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (batch, 224, 224, 3), generator=g, dtype=torch.uint8)
    return cfg, img, seed


def gen_clip(out):
    """``transformers.CLIPVisionModelWithProjection`` - the class the reference imports (pipeline...:22) - on random-init
    configs, weights from oracle.init.seeded_init_ (name-keyed, rounded to fp16-representable values)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    for name in CLIP_CASES:
        cfg, img, seed = clip_case_inputs(name)
        m = CLIPVisionModelWithProjection(CLIPVisionConfig(**cfg)).eval()
        OI.seeded_init_(m, seed=seed)
        with torch.no_grad():
            for prm in m.parameters():
                prm.copy_(prm.half().float())
            x = img.permute(0, 3, 1, 2).float() / 255.0                    # what _encode_image feeds: [0, 1], no CLIP normalisation (Q7)
            y = m(pixel_values=x)
        out[f"{name}_image_u8"] = img.numpy()
        out[f"{name}_image_embeds"] = y.image_embeds.numpy()
        out[f"{name}_hidden_head"] = y.last_hidden_state[:, :4].numpy()        # class token + first three patches
        out[f"{name}_n_keys"] = np.array(len(m.state_dict()))


# ------------------------------------------------------------------------------------ G9 trajectory maps: the reference's draw calls
class RecordingCV2:
    """Stand-in for ``cv2`` that logs what the reference asks OpenCV to draw (the primitives themselves are OpenCV's)."""
    COLOR_BGR2RGB = 4

    def __init__(self):
        self.calls = []

    def line(self, img, p0, p1, color, thickness):
        assert all(isinstance(v, int) for v in (*p0, *p1))
        self.calls.append((0, p0[0], p0[1], p1[0], p1[1], *color, thickness))

    def circle(self, img, c, radius, color, thickness):
        assert thickness == -1 and all(isinstance(v, int) for v in c)
        self.calls.append((1, c[0], c[1], 0, 0, *color, radius))

    def cvtColor(self, img, code):
        assert code == self.COLOR_BGR2RGB
        self.calls.append((2, 0, 0, 0, 0, 0, 0, 0, 0))
        return img[..., ::-1].copy()


def _reference_statements(path, pick):
    """Source text of the AST nodes of ``path`` selected by ``pick(node)`` - read from /root/reference at generation time and
    executed, never stored."""
    import ast
    src = open(path).read()
    tree = ast.parse(src)
    return [ast.get_source_segment(src, n) for n in ast.walk(tree) if pick(n)]


def synth_tracks(seed, n_tracks, n_points, w0, h0):
    rng = np.random.default_rng(seed)
    tracks = {}
    for i in range(n_tracks):
        p = rng.uniform([0.1 * w0, 0.1 * h0], [0.9 * w0, 0.9 * h0])
        v = rng.normal(0, 0.02 * w0, size=2)
        pts = []
        for _ in range(n_points):
            pts.append([int(p[0]), int(p[1])])
            v = 0.8 * v + rng.normal(0, 0.012 * w0, size=2)
            p = np.clip(p + v, 0, [w0 - 1, h0 - 1])
        tracks[str(i * 7)] = pts
    return tracks


TRACK_CASES = {"a": (5, 14, (720, 1280, 3), [320, 576]), "b": (3, 20, (1080, 1920, 3), [576, 1024]), "c": (4, 14, (333, 517, 3), [320, 576]),
               "d": (3, 14, (490, 564, 3), [320, 576])}     # d holds the point (188, 147): int(x * (W / W0)) = 191 but int(x / W0 * W) = 192


def gen_tracks(out):
    import ast
    import textwrap
    from PIL import Image
    script = os.path.join(REF, "scripts", "run_inference_vipseg_json_repro.py")
    # the two loops of the inference script (:429-444): scaling, then the 13 maps
    def is_loop(n, var):
        return isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == var
    scale_src = [s for s in _reference_statements(script, lambda n: is_loop(n, "index")) if "trajectory_json[index]" in s]
    draw_src = [s for s in _reference_statements(script, lambda n: is_loop(n, "len_index")) if "cv2.line" in s and "pil_mask_img" in s]
    assert len(set(scale_src)) == 1 and len(set(draw_src)) == 1, (len(scale_src), len(draw_src))
    ds = os.path.join(REF, "utils", "dataset.py")
    fn = [s for s in _reference_statements(ds, lambda n: isinstance(n, ast.FunctionDef) and n.name == "draw_traj")]
    assert len(fn) == 1
    for name, (n_tracks, n_points, original_size, size) in TRACK_CASES.items():
        tracks = synth_tracks(ord(name), n_tracks, n_points, original_size[1], original_size[0])
        if name == "d":
            tracks[next(iter(tracks))][3] = [188, 147]
        out[f"{name}_tracks"] = np.array([tracks[k] for k in tracks], dtype=np.int64)
        out[f"{name}_keys"] = np.array(list(tracks))
        out[f"{name}_original_size"] = np.array(original_size)
        out[f"{name}_size"] = np.array(size)
        # inference script
        rec = RecordingCV2()
        ns = dict(trajectory_json=tracks, size=size, original_size=original_size, cv2=rec, np=np, Image=Image,
                  trajectory_list=[], validation_control_images=[])
        exec(textwrap.dedent(scale_src[0]), ns)
        exec(textwrap.dedent(draw_src[0]), ns)
        out[f"{name}_inference_scaled"] = np.array(ns["trajectory_list"], dtype=np.int64)
        out[f"{name}_inference_calls"] = np.array(rec.calls, dtype=np.int64)
        out[f"{name}_inference_n_maps"] = np.array(len(ns["validation_control_images"]))
        # training dataset (cvtColor inside the per-track loop)
        import tempfile
        rec = RecordingCV2()
        with tempfile.TemporaryDirectory() as td:
            import json as _json
            _json.dump(tracks, open(os.path.join(td, "vid.json"), "w"))
            ns = dict(os=os, json=_json, np=np, cv2=rec)
            exec(textwrap.dedent(fn[0]), ns)
            me = types.SimpleNamespace(trajectory_json=td)
            seq = ns["draw_traj"](me, "vid", 2, 2 + 6, size, original_size)
        out[f"{name}_dataset_calls"] = np.array(rec.calls, dtype=np.int64)
        out[f"{name}_dataset_n_maps"] = np.array(len(seq))


# ------------------------------------------------------------------------------------ G10 training objective (forward + loss)
TRAIN_CFG = dict(block_out_channels=(64, 64, 128, 128), num_attention_heads=(1, 1, 2, 2), cross_attention_dim=16,
                 addition_time_embed_dim=8, projection_class_embeddings_input_dim=24, layers_per_block=1, num_frames=4)
TRAIN_CE = (8, 8, 16, 32)


def gen_train(out):
    """The forward half of the reference's ControlNet training step (scripts/train_svd_traj_VIPSeg_14.py:1275-1414): sigma
    sampling, noising, EDM preconditioning, the training-time added_time_ids, conditioning dropout, ControlNet + frozen U-Net
    forward, the weighted MSE and the single-frame "spatial" loss - the script's own statements, extracted at generation time
    and executed over the reference networks (oracle blocks), with the VAE / CLIP stages replaced by given tensors.  Everything
    random the step draws is read back from its namespace and stored as an INPUT of the fixture."""
    import ast
    import math
    import textwrap
    from models.controlnet_sdv import ControlNetSDVModel as RefCN
    from models.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel as RefUNet
    path = os.path.join(REF, "scripts", "train_svd_traj_VIPSeg_14.py")
    src = open(path).read()
    tree = ast.parse(src)
    seg = lambda n: ast.get_source_segment(src, n)
    ns = dict(torch=torch, math=math)
    for n in tree.body:                                          # module level: the two samplers and their constants
        if isinstance(n, ast.FunctionDef) and n.name in ("stratified_uniform", "rand_cosine_interpolated"):
            exec(seg(n), ns)
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) and \
                n.targets[0].id in ("min_value", "max_value", "image_d", "noise_d_low", "noise_d_high", "sigma_data"):
            exec(seg(n), ns)
    main_fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    helper = [n for n in ast.walk(main_fn) if isinstance(n, ast.FunctionDef) and n.name == "_get_add_time_ids"]
    assert len(helper) == 1
    exec(textwrap.dedent(seg(helper[0])), ns)
    withs = [n for n in ast.walk(main_fn) if isinstance(n, ast.With) and "accelerator.accumulate" in seg(n.items[0].context_expr)]
    assert len(withs) == 1
    body = withs[0].body
    start = [i for i, st in enumerate(body) if seg(st).startswith("noise = torch.randn_like")][0]
    stop = [i for i, st in enumerate(body) if seg(st).startswith("avg_loss")][0]
    step_src = "\n".join(textwrap.dedent(seg(st)) for st in body[start:stop])
    # the sampler alone, pinned separately
    torch.manual_seed(3)
    out["sigma_draw_u"] = torch.rand([16]).numpy()
    torch.manual_seed(3)
    out["sigma_draw"] = ns["rand_cosine_interpolated"](shape=[16, ], image_d=ns["image_d"], noise_d_low=ns["noise_d_low"],
                                                       noise_d_high=ns["noise_d_high"], sigma_data=ns["sigma_data"],
                                                       min_value=ns["min_value"], max_value=ns["max_value"]).numpy()
    out["sigma_consts"] = np.array([ns[k] for k in ("min_value", "max_value", "image_d", "noise_d_low", "noise_d_high", "sigma_data")], dtype=np.float64)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        cn = OI.seeded_init_(RefCN(**TRAIN_CFG, conditioning_embedding_out_channels=TRAIN_CE), seed=81).eval()
        unet = OI.seeded_init_(RefUNet(**TRAIN_CFG), seed=82).eval()
        for m in (cn, unet):
            for prm in m.parameters():
                prm.data.copy_(prm.data.half().float())
    g = torch.Generator().manual_seed(83)
    f, hh, ww = 4, 8, 8
    for case, (bsz, drop, seed) in {"b1": (1, 0.1, 5), "b1_nodrop": (1, None, 6), "b1_dropped": (1, 0.45, 7)}.items():
        latents = (torch.randn(bsz, f, 4, hh, ww, generator=g) * 0.18215 * 5).half().float()
        emb = torch.randn(bsz, 1, 16, generator=g).half().float()
        traj = (torch.rand(bsz, f, 3, hh * 8, ww * 8, generator=g) * 2 - 1).half().float()
        env = dict(ns)
        env.update(latents=latents.clone(), vae=types.SimpleNamespace(config=types.SimpleNamespace(scaling_factor=0.18215)),
                   pixel_values=torch.zeros(bsz, f, 3, 1, 1), encode_image=lambda pv: emb.clone(),
                   batch={"motion_values": torch.tensor([127.0] * bsz), "trajectories": traj.clone()},
                   args=types.SimpleNamespace(conditioning_dropout_prob=drop), generator=torch.Generator().manual_seed(seed),
                   unet=unet, controlnet=cn, weight_dtype=torch.float32)
        torch.manual_seed(seed)
        with torch.no_grad(), contextlib.redirect_stdout(open(os.devnull, "w")):
            exec(step_src, env)
        k = case + "_"
        out[k + "latents"], out[k + "emb"], out[k + "traj"] = latents.numpy(), emb.numpy(), traj.numpy()
        out[k + "drop"] = np.array(-1.0 if drop is None else drop)
        out[k + "noise"] = env["noise"].numpy()
        out[k + "sigmas"] = env["sigmas"].reshape(bsz).numpy()
        out[k + "random_p"] = env["random_p"].numpy() if drop is not None else np.zeros(bsz, dtype=np.float32)
        out[k + "ran_idx"] = np.array(env["ran_idx"])
        out[k + "timesteps"] = env["timesteps"].numpy()
        out[k + "added_time_ids"] = env["added_time_ids"].numpy()
        out[k + "inp_noisy_latents"] = env["inp_noisy_latents"].numpy()
        out[k + "ehs"] = env["encoder_hidden_states"].numpy()
        out[k + "model_pred"] = env["model_pred"].numpy()
        out[k + "loss_spatial"] = np.array(float(env["loss_spatial"]))
        out[k + "loss"] = np.array(float(env["loss"]))


GRAD_SUBSAMPLE = 97          # gradients / updated parameters of tensors above GRAD_FULL elements are stored every 97th element
GRAD_FULL = 4096


def grad_sample(t):
    """The stored part of a parameter-shaped tensor: all of it when small, else a fixed stride over its flattening."""
    f = t.detach().reshape(-1)
    return (f if f.numel() <= GRAD_FULL else f[::GRAD_SUBSAMPLE]).numpy().copy()


def gen_train_grads(out):
    """The WHOLE training step of the reference (scripts/train_svd_traj_VIPSeg_14.py:1275-1425) - the statements of gen_train
    continued through `accelerator.backward(loss)`, `optimizer.step()`, `lr_scheduler.step()`, `optimizer.zero_grad()` - executed
    over the reference networks in fp32, with the optimizer built by the script's own constructor statement (:1070-1076) from
    torch.optim.AdamW (:1051) and its argparse defaults.  Stored: the draws (inputs), both losses, every ControlNet parameter's
    gradient (norm and sum of all, values of the small ones, a strided sample of the large ones) captured inside
    `optimizer.step()`, and the same sample of the parameters after the step."""
    import ast
    import math
    import textwrap
    from models.controlnet_sdv import ControlNetSDVModel as RefCN
    from models.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel as RefUNet
    path = os.path.join(REF, "scripts", "train_svd_traj_VIPSeg_14.py")
    src = open(path).read()
    tree = ast.parse(src)
    seg = lambda n: ast.get_source_segment(src, n)
    ns = dict(torch=torch, math=math)
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and n.name in ("stratified_uniform", "rand_cosine_interpolated"):
            exec(seg(n), ns)
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) and \
                n.targets[0].id in ("min_value", "max_value", "image_d", "noise_d_low", "noise_d_high", "sigma_data"):
            exec(seg(n), ns)
    main_fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    helper = [n for n in ast.walk(main_fn) if isinstance(n, ast.FunctionDef) and n.name == "_get_add_time_ids"]
    exec(textwrap.dedent(seg(helper[0])), ns)
    opt_stmt = [n for n in ast.walk(main_fn) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name) and
                n.targets[0].id == "optimizer" and isinstance(n.value, ast.Call) and getattr(n.value.func, "id", "") == "optimizer_cls"]
    assert len(opt_stmt) == 1
    parser_fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "parse_args"][0]
    defaults = {}
    for c in ast.walk(parser_fn):                                 # argparse defaults of the optimizer arguments
        if isinstance(c, ast.Call) and getattr(c.func, "attr", "") == "add_argument" and c.args and isinstance(c.args[0], ast.Constant):
            name = c.args[0].value.lstrip("-")
            for kw in c.keywords:
                if kw.arg == "default" and isinstance(kw.value, ast.Constant):
                    defaults[name] = kw.value.value
    withs = [n for n in ast.walk(main_fn) if isinstance(n, ast.With) and "accelerator.accumulate" in seg(n.items[0].context_expr)]
    body = withs[0].body
    start = [i for i, st in enumerate(body) if seg(st).startswith("noise = torch.randn_like")][0]
    stop = [i for i, st in enumerate(body) if seg(st).startswith("optimizer.zero_grad")][0] + 1
    step_src = "\n".join(textwrap.dedent(seg(st)) for st in body[start:stop])
    assert "accelerator.backward(loss)" in step_src and "optimizer.step()" in step_src
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        cn = OI.seeded_init_(RefCN(**TRAIN_CFG, conditioning_embedding_out_channels=TRAIN_CE), seed=81)
        unet = OI.seeded_init_(RefUNet(**TRAIN_CFG), seed=82)
        for m in (cn, unet):
            for prm in m.parameters():
                prm.data.copy_(prm.data.half().float())
    unet.requires_grad_(False)                                     # :953
    cn.requires_grad_(True)                                        # :1053
    args = types.SimpleNamespace(conditioning_dropout_prob=0.1, per_gpu_batch_size=1, gradient_accumulation_steps=1,
                                 learning_rate=defaults["learning_rate"], adam_beta1=defaults["adam_beta1"], adam_beta2=defaults["adam_beta2"],
                                 adam_weight_decay=defaults["adam_weight_decay"], adam_epsilon=defaults["adam_epsilon"])
    out["adam"] = np.array([args.learning_rate, args.adam_beta1, args.adam_beta2, args.adam_weight_decay, args.adam_epsilon], dtype=np.float64)
    env0 = dict(ns, optimizer_cls=torch.optim.AdamW, controlnet=cn, args=args)
    exec(textwrap.dedent(seg(opt_stmt[0])), env0)
    real_opt = env0["optimizer"]
    captured = {}

    class RecordingOptimizer:
        def step(self):
            captured.update({k: (torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()) for k, p in cn.named_parameters()})
            real_opt.step()

        def zero_grad(self):
            real_opt.zero_grad()

    g = torch.Generator().manual_seed(84)
    f, hh, ww = 4, 8, 8
    bsz, seed = 1, 9
    latents = (torch.randn(bsz, f, 4, hh, ww, generator=g) * 0.18215 * 5).half().float()
    emb = torch.randn(bsz, 1, 16, generator=g).half().float()
    traj = (torch.rand(bsz, f, 3, hh * 8, ww * 8, generator=g) * 2 - 1).half().float()
    env = dict(ns)
    env.update(latents=latents.clone(), vae=types.SimpleNamespace(config=types.SimpleNamespace(scaling_factor=0.18215)),
               pixel_values=torch.zeros(bsz, f, 3, 1, 1), encode_image=lambda pv: emb.clone(),
               batch={"motion_values": torch.tensor([127.0] * bsz), "trajectories": traj.clone()},
               args=args, generator=torch.Generator().manual_seed(seed), unet=unet, controlnet=cn, weight_dtype=torch.float32,
               accelerator=types.SimpleNamespace(gather=lambda x: x, backward=lambda l: l.backward()), train_loss=0.0,
               optimizer=RecordingOptimizer(), lr_scheduler=types.SimpleNamespace(step=lambda: None))
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        exec(step_src, env)
    out["latents"], out["emb"], out["traj"] = latents.numpy(), emb.numpy(), traj.numpy()
    out["noise"], out["sigmas"] = env["noise"].numpy(), env["sigmas"].reshape(bsz).numpy()
    out["random_p"], out["ran_idx"] = env["random_p"].numpy(), np.array(env["ran_idx"])
    out["loss"], out["loss_spatial"] = np.array(float(env["loss"])), np.array(float(env["loss_spatial"]))
    names = [k for k, _ in cn.named_parameters()]
    assert set(names) == set(captured)
    out["names"] = np.array(names)
    out["grad_norm"] = np.array([float(captured[k].norm()) for k in names], dtype=np.float64)
    out["grad_sum"] = np.array([float(captured[k].double().sum()) for k in names], dtype=np.float64)
    after = dict(cn.named_parameters())
    out["grad_samples"] = np.concatenate([grad_sample(captured[k]) for k in names])
    out["after_samples"] = np.concatenate([grad_sample(after[k]) for k in names])
    assert all(p.grad is None for p in cn.parameters())           # optimizer.zero_grad() ran


# ------------------------------------------------------------------------------------ the REAL diffusers 0.24.0, where a machine has it
def real_diffusers():
    """The installed package if it is the version the reference pins (requirements.txt:4), else None.  Must run BEFORE
    install_standins(), which shadows the name in sys.modules."""
    try:
        import importlib.metadata as md
        if md.version("diffusers") != "0.24.0":
            print(f"diffusers {md.version('diffusers')} is installed, the reference pins 0.24.0: leaves stay unpinned")
            return None
        import diffusers
        return diffusers
    except Exception:
        return None


def _load_strict(real: nn.Module, oracle: nn.Module, what: str) -> nn.Module:
    """The oracle's module names ARE diffusers' (the HIP models load diffusers-format state dicts through the same names): anything
    missing or unexpected is a finding about the restatement, printed in full."""
    missing, unexpected = real.load_state_dict(oracle.state_dict(), strict=False)
    if missing or unexpected:
        raise RuntimeError(f"{what}: state-dict names differ from diffusers 0.24.0 - missing {list(missing)[:8]}, unexpected {list(unexpected)[:8]}")
    return real.eval()


def gen_blocks_real(out):
    """blocks.npz's four modules as diffusers' OWN classes over the oracle's seeded weights, same inputs (tests/parity.py: blocks_modules /
    blocks_inputs).  Constructor keywords as of diffusers 0.24.0 (models/attention.py, transformer_temporal.py, unet_3d_blocks.py)."""
    from diffusers.models.attention import TemporalBasicTransformerBlock
    from diffusers.models.transformer_temporal import TransformerSpatioTemporalModel
    from diffusers.models.unet_3d_blocks import CrossAttnDownBlockSpatioTemporal, CrossAttnUpBlockSpatioTemporal
    c, c2, te, xd = BLK["C"], BLK["C2"], BLK["temb"], BLK["xdim"]
    o = blocks_modules()
    real = dict(
        temporal=_load_strict(TemporalBasicTransformerBlock(dim=c, time_mix_inner_dim=c, num_attention_heads=1, attention_head_dim=64,
                                                            cross_attention_dim=xd), o["temporal"], "TemporalBasicTransformerBlock"),
        transformer=_load_strict(TransformerSpatioTemporalModel(num_attention_heads=1, attention_head_dim=64, in_channels=c, num_layers=1,
                                                                cross_attention_dim=xd), o["transformer"], "TransformerSpatioTemporalModel"),
        down=_load_strict(CrossAttnDownBlockSpatioTemporal(in_channels=c, out_channels=c2, temb_channels=te, num_layers=2,
                                                           transformer_layers_per_block=1, num_attention_heads=2, cross_attention_dim=xd,
                                                           add_downsample=True), o["down"], "CrossAttnDownBlockSpatioTemporal"),
        up=_load_strict(CrossAttnUpBlockSpatioTemporal(in_channels=c, out_channels=c2, prev_output_channel=c2, temb_channels=te, num_layers=3,
                                                       transformer_layers_per_block=1, resnet_eps=1e-5, num_attention_heads=2,
                                                       cross_attention_dim=xd, add_upsample=True), o["up"], "CrossAttnUpBlockSpatioTemporal"),
    )
    i = blocks_inputs()
    ind = torch.zeros(BLK["B"], BLK["F"])
    import diffusers
    out["diffusers_version"] = np.array(diffusers.__version__)
    with torch.no_grad():
        out["temporal"] = real["temporal"](i["tokens"], num_frames=BLK["F"], encoder_hidden_states=i["tctx"]).numpy()
        out["transformer"] = real["transformer"](i["x"], encoder_hidden_states=i["ehs"], image_only_indicator=ind, return_dict=False)[0].numpy()
        y, taps = real["down"](i["x"], temb=i["temb"], encoder_hidden_states=i["ehs"], image_only_indicator=ind)
        out["down"] = y.numpy()
        for j, t in enumerate(taps):
            out[f"down_tap{j}"] = t.numpy()
        skips = (i["up_skip_in"], i["up_skips"][0], i["up_skips"][1])
        out["up"] = real["up"](i["up_x"], skips, temb=i["temb"], encoder_hidden_states=i["ehs"], image_only_indicator=ind).numpy()


def gen_vae_io_real(out):
    """diffusers' AutoencoderKLTemporalDecoder (tiny config, the oracle's seeded weights): decode of the latents of vae_io.npz's
    `dl_b1f6_c14` case (one chunk of 6 frames) and encode().latent_dist.mode() / .mean / .logvar of a seeded image."""
    from diffusers.models import AutoencoderKLTemporalDecoder
    ov = OI.seeded_init_(OV.AutoencoderKLTemporalDecoder(**OV.tiny_vae_config()), seed=VAE_SEED).eval()
    for prm in ov.parameters():
        prm.data.copy_(prm.data.half().float())
    real = _load_strict(AutoencoderKLTemporalDecoder(**OV.tiny_vae_config()), ov, "AutoencoderKLTemporalDecoder")
    g = torch.Generator().manual_seed(61)
    lat = torch.randn(1, 6, 4, 4, 4, generator=g) * 0.18215 * 1.3                      # the first draw of gen_vae_io: dl_b1f6_c14_latents
    img = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(62)) * 2 - 1
    import diffusers
    out["diffusers_version"] = np.array(diffusers.__version__)
    with torch.no_grad():
        z = lat.flatten(0, 1) / real.config.scaling_factor
        out["latents"], out["image"] = lat.numpy(), img.numpy()
        out["decoded"] = real.decode(z, num_frames=6).sample.numpy()
        d = real.encode(img).latent_dist
        out["enc_mode"], out["enc_mean"], out["enc_logvar"] = d.mode().numpy(), d.mean.numpy(), d.logvar.numpy()


def main():
    only = set(sys.argv[1:])
    if real_diffusers() is not None:                       # first, with the REAL package importable; then it is shadowed below
        for name, fn in (("blocks_real", gen_blocks_real), ("vae_io_real", gen_vae_io_real)):
            if only and name not in only:
                continue
            out = {}
            fn(out)
            path = os.path.join(HERE, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB   (diffusers {out['diffusers_version']})")
        for k in [k for k in sys.modules if k == "diffusers" or k.startswith("diffusers.")]:
            del sys.modules[k]
    elif only & {"blocks_real", "vae_io_real"}:
        raise SystemExit("blocks_real / vae_io_real need the real diffusers==0.24.0 (pip install diffusers==0.24.0): not importable here")
    install_standins()
    for name, fn in (("sched", gen_sched), ("add_noise", gen_add_noise), ("cond_embed", gen_cond_embed), ("wiring", gen_wiring),
                     ("loop", gen_loop), ("blocks", gen_blocks), ("resize", gen_resize), ("vae_io", gen_vae_io), ("clip", gen_clip), ("tracks", gen_tracks), ("train", gen_train), ("train_grads", gen_train_grads)):
        if only and name not in only:
            continue
        out = {}
        fn(out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
