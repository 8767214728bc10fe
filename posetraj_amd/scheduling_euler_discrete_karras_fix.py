"""MI355X-native ``EulerDiscreteScheduler`` - drop-in for
``/root/reference/utils/scheduling_euler_discrete_karras_fix.py:133-556``.

The sigma / timestep tables are host-side numpy exactly as in the reference (they are a few hundred floats); the
per-step tensor maths (``scale_model_input``, ``step``) runs in ``libposetraj_hip.so`` on the device and refuses CPU
tensors.  The scheduler is stateful (``_step_index``): one instance per in-flight batch of clips.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import ops
from .modeling import BaseOutput, FrozenConfig

PREDICTION_TYPES = {"v_prediction": 0, "epsilon": 1, "sample": 2, "original_sample": 2}


class EulerDiscreteSchedulerOutput(BaseOutput):
    """``prev_sample`` and ``pred_original_sample`` (``scheduling...:32-48``); the latter is not materialised here."""


def betas_for_alpha_bar(n, max_beta=0.999):
    """Glide cosine schedule (``:52-93``)."""
    bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - bar((i + 1) / n) / bar(i / n), max_beta) for i in range(n)], dtype=torch.float32)


def rescale_zero_terminal_snr(betas):
    """``:97-130``."""
    a = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0, aT = a[0].clone(), a[-1].clone()
    a = (a - aT) * (a0 / (a0 - aT))
    abar = a ** 2
    return 1 - torch.cat([abar[0:1], abar[1:] / abar[:-1]])


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas: Optional[Union[np.ndarray, List[float]]] = None,
                 prediction_type: str = "epsilon", interpolation_type: str = "linear",
                 use_karras_sigmas: Optional[bool] = False, sigma_min: Optional[float] = None,
                 sigma_max: Optional[float] = None, timestep_spacing: str = "linspace", timestep_type: str = "discrete",
                 steps_offset: int = 0, rescale_betas_zero_snr: bool = False):
        self.config = FrozenConfig(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
            trained_betas=trained_betas, prediction_type=prediction_type, interpolation_type=interpolation_type,
            use_karras_sigmas=use_karras_sigmas, sigma_min=sigma_min, sigma_max=sigma_max,
            timestep_spacing=timestep_spacing, timestep_type=timestep_type, steps_offset=steps_offset,
            rescale_betas_zero_snr=rescale_betas_zero_snr)
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            self.betas = betas_for_alpha_bar(num_train_timesteps)
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        if rescale_betas_zero_snr:
            self.alphas_cumprod[-1] = 2 ** -24
        self.use_karras_sigmas = use_karras_sigmas
        sigmas = self._train_sigmas()[::-1].copy()
        timesteps = np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy()
        if use_karras_sigmas:                       # the "Karras fix" (:225-228)
            log_sigmas = np.log(sigmas)
            sigmas = self._convert_to_karras(sigmas, num_train_timesteps)
            timesteps = np.array([self._sigma_to_t(s, log_sigmas) for s in sigmas])
        self.num_inference_steps = None
        self._install(sigmas, timesteps, None)
        self.is_scale_input_called = False

    # ------------------------------------------------------------------ tables (host)
    def _train_sigmas(self) -> np.ndarray:
        return (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()

    def _install(self, sigmas: np.ndarray, timesteps: np.ndarray, device):
        sig = torch.from_numpy(sigmas).to(dtype=torch.float32)
        if self.config.timestep_type == "continuous" and self.config.prediction_type == "v_prediction":
            ts = torch.Tensor([0.25 * s.log() for s in sig])
        else:
            ts = torch.from_numpy(timesteps.astype(np.float32))
        self.timesteps = ts.to(device=device)
        self.sigmas = torch.cat([sig, torch.zeros(1)]).to(device=device)
        self._sigmas_host = [float(v) for v in torch.cat([sig, torch.zeros(1)])]       # no device sync per step
        self._timesteps_host = ts.clone()
        self._step_index = None

    def _convert_to_karras(self, in_sigmas, num_inference_steps):
        """rho = 7 ladder between sigma_max and sigma_min (``:376-399``)."""
        smin = self.config.sigma_min if self.config.sigma_min is not None else in_sigmas[-1].item()
        smax = self.config.sigma_max if self.config.sigma_max is not None else in_sigmas[0].item()
        rho = 7.0
        ramp = np.linspace(0, 1, num_inference_steps)
        a, b = smin ** (1 / rho), smax ** (1 / rho)
        return (b + ramp * (a - b)) ** rho

    @staticmethod
    def _sigma_to_t(sigma, log_sigmas):
        """``:352-373``."""
        log_sigma = np.log(np.maximum(sigma, 1e-10))
        dists = log_sigma - log_sigmas[:, np.newaxis]
        low_idx = np.cumsum((dists >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = log_sigmas[low_idx], log_sigmas[high_idx]
        w = np.clip((low - log_sigma) / (low - high), 0, 1)
        return ((1 - w) * low_idx + w * high_idx).reshape(sigma.shape)

    @property
    def init_noise_sigma(self):
        """``:248-255``."""
        m = self.sigmas.max()
        if self.config.timestep_spacing in ["linspace", "trailing"]:
            return m
        return (m ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device] = None):
        """``:290-350``."""
        cfg = self.config
        self.num_inference_steps = num_inference_steps
        T = cfg.num_train_timesteps
        if cfg.timestep_spacing == "linspace":
            timesteps = np.linspace(0, T - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif cfg.timestep_spacing == "leading":
            step_ratio = T // num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.float32)
            timesteps += cfg.steps_offset
        elif cfg.timestep_spacing == "trailing":
            step_ratio = T / num_inference_steps
            timesteps = (np.arange(T, 0, -step_ratio)).round().copy().astype(np.float32)
            timesteps -= 1
        else:
            raise ValueError(f"{cfg.timestep_spacing} is not supported. Please make sure to choose one of 'linspace', 'leading' or 'trailing'.")
        sigmas = self._train_sigmas()
        log_sigmas = np.log(sigmas)
        if cfg.interpolation_type == "linear":
            sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        elif cfg.interpolation_type == "log_linear":
            sigmas = torch.linspace(np.log(sigmas[-1]), np.log(sigmas[0]), num_inference_steps + 1).exp().numpy()
        else:
            raise ValueError(f"{cfg.interpolation_type} is not implemented. Please specify interpolation_type to either 'linear' or 'log_linear'")
        if self.use_karras_sigmas:
            sigmas = self._convert_to_karras(sigmas, num_inference_steps)
            timesteps = np.array([self._sigma_to_t(s, log_sigmas) for s in sigmas])
        self._install(sigmas, timesteps, device)

    def _init_step_index(self, timestep):
        """``:401-416``: the second match when a timestep value is duplicated."""
        t = timestep.detach().to("cpu") if isinstance(timestep, torch.Tensor) else timestep
        cand = (self._timesteps_host == t).nonzero()
        self._step_index = (cand[1] if len(cand) > 1 else cand[0]).item()

    # ------------------------------------------------------------------ per-step tensor maths (device)
    def scale_model_input(self, sample: torch.Tensor, timestep) -> torch.Tensor:
        """x / sqrt(sigma^2 + 1)  (``:264-288``)."""
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self._sigmas_host[self._step_index]
        self.is_scale_input_called = True
        return ops.scale(sample, 1.0 / math.sqrt(sigma * sigma + 1.0))

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator: Optional[torch.Generator] = None,
             return_dict: bool = True) -> Union[EulerDiscreteSchedulerOutput, Tuple]:
        """Deterministic Euler step (``:418-528``).  ``s_churn > 0`` (stochastic churn) is not on the PoseTraj path
        (the pipeline never passes it) and is rejected.  The reference draws an unused ``randn`` per step on the
        gamma == 0 path (``:487-489``); that dead draw is not reproduced, so global-RNG consumers downstream of the loop
        see a different stream position."""
        if isinstance(timestep, int) or isinstance(timestep, (torch.IntTensor, torch.LongTensor)) or \
                (isinstance(timestep, torch.Tensor) and timestep.dtype in (torch.int32, torch.int64)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to"
                             " `EulerDiscreteScheduler.step()` is not supported. Make sure to pass"
                             " one of the `scheduler.timesteps` as a timestep.")
        if s_churn != 0.0:
            raise NotImplementedError("posetraj_amd EulerDiscreteScheduler: s_churn > 0 is outside the PoseTraj hot path")
        if self.config.prediction_type not in PREDICTION_TYPES:
            raise ValueError(f"prediction_type given as {self.config.prediction_type} must be one of `epsilon`, or `v_prediction`")
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma, sigma_next = self._sigmas_host[self._step_index], self._sigmas_host[self._step_index + 1]
        prev = ops.euler_step(model_output, sample.to(torch.float32), sigma, sigma_next,
                              PREDICTION_TYPES[self.config.prediction_type]).to(model_output.dtype)
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return EulerDiscreteSchedulerOutput(prev_sample=prev, pred_original_sample=None)

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """``:530-553``: ``x + noise * sigma[index of t]`` per sample (the training script's forward-noising call site).
        The table look-up is host work; the arithmetic runs in ``pt_add_noise`` in the samples' own dtype."""
        ts = timesteps.detach().to("cpu") if isinstance(timesteps, torch.Tensor) else torch.as_tensor(timesteps)
        idx = []
        for t in ts.reshape(-1):
            cand = (self._timesteps_host == t).nonzero()
            if len(cand) != 1:                           # the reference's .item() raises on 0 or several matches
                raise ValueError(f"timestep {float(t)} matches {len(cand)} entries of the schedule")
            idx.append(int(cand[0]))
        sig = torch.tensor([self._sigmas_host[i] for i in idx], dtype=torch.float32, device=original_samples.device)
        return ops.add_noise(original_samples, noise.to(original_samples.device), sig)

    def __len__(self):
        return self.config.num_train_timesteps


SVD_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True, sigma_min=0.002,
    sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous", steps_offset=1,
    rescale_betas_zero_snr=False)
