"""Oracle restatement of ``/root/reference/utils/scheduling_euler_discrete_karras_fix.py``.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  PINNED: checked against the reference class itself
(golden fixtures ``tests/golden/sched_*.npz`` produced by ``tests/golden/make_golden.py``).

The arithmetic uses the same numpy/torch primitives in the same order as the reference so the fp32 tables
(`sigmas`, `timesteps`) come out bit-identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch


@dataclass
class EulerStepOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


def _betas(cfg) -> torch.Tensor:
    """``:196-207`` (+ cosine helper ``:52-93``)."""
    n = cfg.num_train_timesteps
    if cfg.trained_betas is not None:
        return torch.tensor(cfg.trained_betas, dtype=torch.float32)
    if cfg.beta_schedule == "linear":
        return torch.linspace(cfg.beta_start, cfg.beta_end, n, dtype=torch.float32)
    if cfg.beta_schedule == "scaled_linear":
        return torch.linspace(cfg.beta_start ** 0.5, cfg.beta_end ** 0.5, n, dtype=torch.float32) ** 2
    if cfg.beta_schedule == "squaredcos_cap_v2":
        bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return torch.tensor([min(1 - bar((i + 1) / n) / bar(i / n), 0.999) for i in range(n)], dtype=torch.float32)
    raise NotImplementedError(cfg.beta_schedule)


def _zero_snr(betas: torch.Tensor) -> torch.Tensor:
    """``rescale_zero_terminal_snr`` ``:97-130``."""
    abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt -= aT
    abar_sqrt *= a0 / (a0 - aT)
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


class OracleEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, prediction_type="epsilon", interpolation_type="linear",
                 use_karras_sigmas=False, sigma_min=None, sigma_max=None, timestep_spacing="linspace",
                 timestep_type="discrete", steps_offset=0, rescale_betas_zero_snr=False):
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, trained_betas=trained_betas, prediction_type=prediction_type,
            interpolation_type=interpolation_type, use_karras_sigmas=use_karras_sigmas, sigma_min=sigma_min,
            sigma_max=sigma_max, timestep_spacing=timestep_spacing, timestep_type=timestep_type,
            steps_offset=steps_offset, rescale_betas_zero_snr=rescale_betas_zero_snr)
        cfg = self.config
        self.betas = _betas(cfg)
        if rescale_betas_zero_snr:
            self.betas = _zero_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        if rescale_betas_zero_snr:
            self.alphas_cumprod[-1] = 2 ** -24                                                   # :218
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()[::-1].copy()   # :220-223
        ts = np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy()
        # the "Karras fix" (:225-228): conversion also at construction so init_noise_sigma is right
        # before set_timesteps.  The reference reads use_karras_sigmas through the config fallback here.
        self.use_karras_sigmas = use_karras_sigmas
        if use_karras_sigmas:
            log_sig = np.log(sig)
            sig = self._karras(sig, num_train_timesteps)
            ts = np.array([self._sigma_to_t(s, log_sig) for s in sig])
        sig_t = torch.from_numpy(sig).to(dtype=torch.float32)
        self.num_inference_steps = None
        if timestep_type == "continuous" and prediction_type == "v_prediction":
            self.timesteps = torch.Tensor([0.25 * s.log() for s in sig_t])                       # :236-237
        else:
            self.timesteps = torch.from_numpy(ts.astype(np.float32))
        self.sigmas = torch.cat([sig_t, torch.zeros(1)])
        self.is_scale_input_called = False
        self._step_index = None

    # ---- tables
    def _karras(self, in_sigmas, n):
        """``:376-399``; rho = 7."""
        smin = self.config.sigma_min if self.config.sigma_min is not None else in_sigmas[-1].item()
        smax = self.config.sigma_max if self.config.sigma_max is not None else in_sigmas[0].item()
        rho = 7.0
        ramp = np.linspace(0, 1, n)
        lo, hi = smin ** (1 / rho), smax ** (1 / rho)
        return (hi + ramp * (lo - hi)) ** rho

    @staticmethod
    def _sigma_to_t(sigma, log_sigmas):
        """``:352-373``."""
        ls = np.log(np.maximum(sigma, 1e-10))
        d = ls - log_sigmas[:, np.newaxis]
        lo_i = np.cumsum((d >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
        hi_i = lo_i + 1
        lo, hi = log_sigmas[lo_i], log_sigmas[hi_i]
        w = np.clip((lo - ls) / (lo - hi), 0, 1)
        return ((1 - w) * lo_i + w * hi_i).reshape(sigma.shape)

    @property
    def init_noise_sigma(self):
        """``:248-255``."""
        m = self.sigmas.max()
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return m
        return (m ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps, device=None):
        """``:290-350``."""
        cfg = self.config
        self.num_inference_steps = n = num_inference_steps
        T = cfg.num_train_timesteps
        if cfg.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, n, dtype=np.float32)[::-1].copy()
        elif cfg.timestep_spacing == "leading":
            ratio = T // n
            ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.float32)
            ts += cfg.steps_offset
        elif cfg.timestep_spacing == "trailing":
            ratio = T / n
            ts = (np.arange(T, 0, -ratio)).round().copy().astype(np.float32)
            ts -= 1
        else:
            raise ValueError(f"{cfg.timestep_spacing} is not supported.")
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        log_sig = np.log(sig)
        if cfg.interpolation_type == "linear":
            sig = np.interp(ts, np.arange(0, len(sig)), sig)
        elif cfg.interpolation_type == "log_linear":
            sig = torch.linspace(np.log(sig[-1]), np.log(sig[0]), n + 1).exp().numpy()
        else:
            raise ValueError(f"{cfg.interpolation_type} is not implemented.")
        if self.use_karras_sigmas:
            sig = self._karras(sig, n)
            ts = np.array([self._sigma_to_t(s, log_sig) for s in sig])
        sig_t = torch.from_numpy(sig).to(dtype=torch.float32, device=device)
        if cfg.timestep_type == "continuous" and cfg.prediction_type == "v_prediction":
            self.timesteps = torch.Tensor([0.25 * s.log() for s in sig_t]).to(device=device)
        else:
            self.timesteps = torch.from_numpy(ts.astype(np.float32)).to(device=device)
        self.sigmas = torch.cat([sig_t, torch.zeros(1, device=sig_t.device)])
        self._step_index = None

    def _init_step_index(self, timestep):
        """``:401-416``: second match if the timestep is duplicated."""
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.to(self.timesteps.device)
        cand = (self.timesteps == timestep).nonzero()
        self._step_index = (cand[1] if len(cand) > 1 else cand[0]).item()

    # ---- per-step maths
    def scale_model_input(self, sample, timestep):
        """``:264-288``: x / sqrt(sigma^2 + 1)."""
        if self._step_index is None:
            self._init_step_index(timestep)
        s = self.sigmas[self._step_index]
        self.is_scale_input_called = True
        return sample / ((s ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0,
             generator=None, return_dict=True):
        """``:418-528``.  The reference always draws a ``randn`` of ``model_output.shape`` (``:487-489``) even on
        the gamma == 0 path; it is drawn here too so the global RNG stream stays in lock-step."""
        if isinstance(timestep, (int, torch.IntTensor, torch.LongTensor)):
            raise ValueError("Passing integer indices as timesteps to `EulerDiscreteScheduler.step()` is not supported.")
        if self._step_index is None:
            self._init_step_index(timestep)
        x = sample.to(torch.float32)
        s = self.sigmas[self._step_index]
        gamma = min(s_churn / (len(self.sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= s <= s_tmax else 0.0
        noise = torch.randn(model_output.shape, dtype=model_output.dtype, device=model_output.device,
                            generator=generator)
        s_hat = s * (gamma + 1)
        if gamma > 0:
            x = x + noise * s_noise * (s_hat ** 2 - s ** 2) ** 0.5
        pt = self.config.prediction_type
        if pt in ("original_sample", "sample"):
            x0 = model_output
        elif pt == "epsilon":
            x0 = x - s_hat * model_output
        elif pt == "v_prediction":
            x0 = model_output * (-s / (s ** 2 + 1) ** 0.5) + (x / (s ** 2 + 1))                 # :506
        else:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, or `v_prediction`")
        deriv = (x - x0) / s_hat
        dt = self.sigmas[self._step_index + 1] - s_hat
        prev = (x + deriv * dt).to(model_output.dtype)
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return EulerStepOutput(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        """``:530-553``."""
        sig = self.sigmas.to(device=original_samples.device, dtype=original_samples.dtype)
        sched_t = self.timesteps.to(original_samples.device)
        idx = [(sched_t == t).nonzero().item() for t in timesteps.to(original_samples.device)]
        s = sig[idx].flatten()
        while s.ndim < original_samples.ndim:
            s = s.unsqueeze(-1)
        return original_samples + noise * s

    def __len__(self):
        return self.config.num_train_timesteps


SVD_SCHEDULER_CONFIG = dict(
    # SVD-img2vid scheduler/scheduler_config.json (not in the reference tree)
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True, sigma_min=0.002,
    sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous", steps_offset=1,
    rescale_betas_zero_snr=False)
