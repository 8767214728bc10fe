"""Shared model plumbing: config objects, checkpoint IO in the diffusers directory layout, the encoder half that
ControlNet and the U-Net have in common.  No arithmetic here."""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import blocks as B
from . import ops
from .packing import pack_conv2d

DOWN_TYPES = ("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",)
UP_TYPES = ("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3


class FrozenConfig(OrderedDict):
    """``model.config.x`` and ``model.config["x"]`` both work, like diffusers' FrozenDict."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class BaseOutput(OrderedDict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to_tuple(self):
        return tuple(self.values())

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return self.to_tuple()[k]


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def check_block_args(down_block_types, up_block_types, block_out_channels, num_attention_heads, cross_attention_dim,
                     layers_per_block):
    """The constructor checks of ``controlnet_sdv.py:273-296`` / ``unet...:101-124`` (same messages)."""
    if len(down_block_types) != len(up_block_types):
        raise ValueError(f"Must provide the same number of `down_block_types` as `up_block_types`. `down_block_types`: {down_block_types}. `up_block_types`: {up_block_types}.")
    if len(block_out_channels) != len(down_block_types):
        raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. `block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
    if not isinstance(num_attention_heads, int) and len(num_attention_heads) != len(down_block_types):
        raise ValueError(f"Must provide the same number of `num_attention_heads` as `down_block_types`. `num_attention_heads`: {num_attention_heads}. `down_block_types`: {down_block_types}.")
    if isinstance(cross_attention_dim, list) and len(cross_attention_dim) != len(down_block_types):
        raise ValueError(f"Must provide the same number of `cross_attention_dim` as `down_block_types`. `cross_attention_dim`: {cross_attention_dim}. `down_block_types`: {down_block_types}.")
    if not isinstance(layers_per_block, int) and len(layers_per_block) != len(down_block_types):
        raise ValueError(f"Must provide the same number of `layers_per_block` as `down_block_types`. `layers_per_block`: {layers_per_block}. `down_block_types`: {down_block_types}.")


def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    return load_file(path)


class HipModel:
    """Common surface of the two networks: config, weights in / out, device handling.

    Weights arrive in the reference's state-dict format and are packed once for the kernels
    (``load_state_dict``).  The model only runs on a ROCm device; ``.to("cpu")`` for compute is not offered."""
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"
    _generations = 0
    _generation = 0

    def __init__(self, **cfg):
        # JSON round trips turn tuples into lists: keep one canonical form so configs compare equal
        self.config = FrozenConfig({k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items()})
        self.device = None
        self.dtype = torch.float16
        self._loaded = False
        self._source: Optional[Dict[str, torch.Tensor]] = None

    # -- spec / IO
    def param_spec(self):
        raise NotImplementedError

    def _validate(self, sd):
        spec = self.param_spec()
        missing = [k for k in spec if k not in sd]
        unexpected = [k for k in sd if k not in spec]
        bad = [k for k in spec if k in sd and tuple(sd[k].shape) != tuple(spec[k])]
        if missing or unexpected or bad:
            raise RuntimeError(f"{type(self).__name__}.load_state_dict: missing={missing[:5]}({len(missing)}) "
                               f"unexpected={unexpected[:5]}({len(unexpected)}) shape_mismatch={bad[:5]}({len(bad)})")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device=None, keep_source: bool = False):
        device = torch.device(device if device is not None else (self.device or "cuda"))
        if device.type != "cuda":
            raise RuntimeError("posetraj_amd models run on a ROCm GPU only (device must be cuda:N)")
        self._validate(sd)
        self.device = device
        self._pack(sd, device)
        HipModel._generations += 1                      # a captured hipGraph holds the packed tensors' addresses
        self._generation = HipModel._generations
        self._source = {k: v.detach().to("cpu", torch.float16) for k, v in sd.items()} if keep_source else None
        self._loaded = True
        return self

    def state_dict(self):
        if self._source is None:
            raise RuntimeError("weights were packed for the kernels; reload with keep_source=True to export them")
        return dict(self._source)

    def init_random_(self, seed: int = 0, device="cuda", zero_conv_std: float = 0.02, keep_source: bool = False):
        """Random-init weights of this architecture, created directly on the device (benchmarks: no checkpoint can be
        fetched).  PyTorch-default-like scales: weights U(-1/sqrt(fan_in), 1/sqrt(fan_in)), norm weights 1, biases small;
        the zero-initialised ControlNet output convs are re-randomised N(0, zero_conv_std^2)."""
        device = torch.device(device)
        g = torch.Generator(device=device).manual_seed(seed)
        sd = {}
        for k, shape in self.param_spec().items():
            leaf = k.rsplit(".", 1)[-1]
            if leaf == "mix_factor":
                t = torch.full(shape, 0.5, device=device)
            elif ("norm" in k.split(".")[-2]) and leaf == "weight":
                t = torch.ones(shape, device=device)
            elif ("norm" in k.split(".")[-2]) and leaf == "bias":
                t = torch.zeros(shape, device=device)
            else:
                fan_in = 1
                for v in shape[1:]:
                    fan_in *= v
                if leaf == "bias":
                    fan_in = max(shape[0], 1)
                bound = 1.0 / (fan_in ** 0.5)
                t = (torch.rand(shape, generator=g, device=device) * 2 - 1) * bound
                if k.startswith(("controlnet_down_blocks", "controlnet_mid_block", "controlnet_cond_embedding.conv_out")):
                    t = torch.randn(shape, generator=g, device=device) * zero_conv_std
            sd[k] = t.to(torch.float16)
        return self.load_state_dict(sd, device, keep_source=keep_source)

    @classmethod
    def from_config(cls, config, **kw):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, device="cuda", torch_dtype=None, variant=None,
                        keep_source: bool = False, **kw):
        """Reads ``<path>[/subfolder]/config.json`` + ``diffusion_pytorch_model[.variant].safetensors``."""
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = json.load(f)
        model = cls.from_config(cfg, **kw)
        name = cls.weights_name if not variant else cls.weights_name.replace(".safetensors", f".{variant}.safetensors")
        model.load_state_dict(load_state_dict_file(os.path.join(root, name)), device, keep_source=keep_source)
        return model

    def save_pretrained(self, path: str):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file(self.state_dict(), os.path.join(path, self.weights_name))

    # -- nn.Module-ish no-ops kept for API compatibility (inference only)
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def enable_forward_chunking(self, chunk_size=None, dim=0):
        if dim not in (0, 1):
            raise ValueError(f"Make sure to set `dim` to either 0 or 1, not {dim}")

    def set_attention_slice(self, slice_size):
        pass

    def set_default_attn_processor(self):
        pass

    # attention-processor / checkpointing surface of the reference classes (controlnet_sdv.py:408-482,
    # unet...:247-354): there is one attention implementation here (pt_attn_*_f16), so these only keep callers working
    @property
    def attn_processors(self):
        return {}

    def set_attn_processor(self, processor=None, _remove_lora=False):
        if isinstance(processor, dict) and len(processor) != 0:
            raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match the number of attention layers: 0. Please make sure to pass 0 processor classes.")

    def _set_gradient_checkpointing(self, module=None, value=False):
        if value:
            raise NotImplementedError("the model classes run inference; training goes through posetraj_amd.training.ControlNetTrainer, which keeps "
                                      "all activations (no gradient checkpointing)")

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    # -- shared encoder
    def _pack_encoder(self, sd, device):
        cfg = self.config
        ch = tuple(cfg.block_out_channels)
        n = len(ch)
        heads = _tup(cfg.num_attention_heads, n)
        self._temb_stack, self._xattn_stack = B.RowStack(), B.RowStack()
        self.conv_in = pack_conv2d(sd["conv_in.weight"], sd["conv_in.bias"], device)
        self.time = B.TimeEmbedding(sd, ch[0], cfg.addition_time_embed_dim, device)
        self.down_blocks = [B.DownBlock(sd, f"down_blocks.{i}.", typ == "CrossAttnDownBlockSpatioTemporal", heads[i], device,
                                        self._temb_stack, self._xattn_stack)
                            for i, typ in enumerate(cfg.down_block_types)]
        self.mid_block = B.MidBlock(sd, "mid_block.", heads[-1], device, self._temb_stack, self._xattn_stack)

    def _finish_pack(self, device):
        self.temb_all = self._temb_stack.pack(device)
        self.xattn_all = self._xattn_stack.pack(device)

    def _prologue(self, sample, timestep, encoder_hidden_states, added_time_ids, half=None):
        """time embeddings, stacked per-forward GEMMs, channels-last input.  Mirrors ``unet...:386-429``.
        ``half`` (pipeline-private): ``sample`` / ``added_time_ids`` hold slice ``half`` of a batch that was cut in two (the CFG
        halves, one clip each), ``encoder_hidden_states`` still holds the whole batch - the temporal cross-attention of a half
        needs the context rows of both."""
        if not self._loaded:
            raise RuntimeError(f"{type(self).__name__}: no weights loaded (load_state_dict / from_pretrained / init_random_)")
        if sample.dim() != 5:
            raise ValueError(f"sample must be [batch, frames, channels, height, width]; got {tuple(sample.shape)}")
        if not sample.is_cuda:
            raise RuntimeError("posetraj_amd: inputs must be on the ROCm device (no CPU path exists)")
        Bc, F, Cin, h, w = sample.shape
        dev = sample.device
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.float64 if isinstance(timestep, float) else torch.int64)
        emb_silu = self.time.run(timestep, added_time_ids, Bc)
        temb = ops.igemm(emb_silu, self.temb_all)
        nb = encoder_hidden_states.shape[0]
        ehs = encoder_hidden_states.to(device=dev, dtype=torch.float16).reshape(nb, -1).contiguous()
        xattn = ops.igemm(ehs, self.xattn_all) if self.xattn_all is not None else None
        if half is None:
            ctx = B.Ctx(B=Bc, F=F, temb=temb, xattn=xattn)
        else:
            if nb != 2 * Bc or Bc != 1:
                raise ValueError("a half forward takes one clip of a two-row (CFG) batch")
            own = None if xattn is None else xattn[half * Bc:(half + 1) * Bc]
            swapped = None if xattn is None else torch.roll(xattn, 1, 0)
            ctx = B.Ctx(B=Bc, F=F, temb=temb, xattn=own, half=half, xattn_full=xattn, xattn_full_swapped=swapped)
        x = ops.to_channels_last(sample.reshape(Bc * F, Cin, h, w), cpad=self.conv_in.cin)
        return ctx, x, (Bc, F, h, w)
