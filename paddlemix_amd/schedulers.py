"""Scheduler API surface kept for the denoising loop (plumbing, not accelerated math).

Mirrors the reference classes' public contract -- ``set_timesteps`` / ``scale_model_input`` / ``step`` /
``init_noise_sigma`` / ``timesteps`` -- for the schedulers BASELINE.json's configs use (PNDM is what the SD-1.x checkpoints ship with):
  DDIMScheduler                      ppdiffusers/ppdiffusers/schedulers/scheduling_ddim.py:131 (step :350-475)
  EulerDiscreteScheduler             scheduling_euler_discrete.py:94 (scale_model_input :216-238, step :375-478)
  FlowMatchEulerDiscreteScheduler    scheduling_flow_match_euler_discrete.py:44 (step :187-283)
  PNDMScheduler                      scheduling_pndm.py:69 (set_timesteps :178-235, step_prk :260-320, step_plms :322-395)
  DPMSolverMultistepScheduler        scheduling_dpmsolver_multistep.py:66 (deterministic variants, orders 1-2; step :802-878)
  LCMScheduler                       scheduling_lcm.py:140 (set_timesteps :330-466, step :476-566) -- 2-8 step latent-consistency sampling
Schedule tables are float32 numpy like the reference's float32 tensors; ``step`` works on torch tensors of any device.

For deterministic sampling every ``step`` is a linear map  prev = a*sample + b*model_output ; ``step_coefficients``
returns (a, b) so the latent update can run as the library's fused ``mi355x_sd_axpby`` inside a captured graph.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import numpy as np
import torch


def _make_betas(n: int, beta_start: float, beta_end: float, schedule: str) -> np.ndarray:
    if schedule == "linear":
        return np.linspace(beta_start, beta_end, n, dtype=np.float32)
    if schedule == "scaled_linear":  # the latent-diffusion schedule
        return np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float32) ** 2
    raise NotImplementedError(f"{schedule} is not implemented")


def _out(prev, return_dict, **extra):
    if not return_dict:
        return (prev,)
    return SimpleNamespace(prev_sample=prev, **extra)


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", clip_sample: bool = True, set_alpha_to_one: bool = True,
                 steps_offset: int = 0, prediction_type: str = "epsilon", clip_sample_range: float = 1.0,
                 timestep_spacing: str = "leading"):
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k != "self"})
        betas = _make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = np.cumprod(1.0 - betas, dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int):
        c = self.config
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError("`num_inference_steps` cannot be larger than `num_train_timesteps`")
        self.num_inference_steps = num_inference_steps
        if c.timestep_spacing == "leading":
            step_ratio = c.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += c.steps_offset
        elif c.timestep_spacing == "trailing":
            ts = np.round(np.arange(c.num_train_timesteps, 0, -c.num_train_timesteps / num_inference_steps))
            ts = ts.astype(np.int64) - 1
        elif c.timestep_spacing == "linspace":
            ts = np.linspace(0, c.num_train_timesteps - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported")
        self.timesteps = ts

    def _alphas(self, timestep: int) -> Tuple[float, float]:
        prev_t = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)

    def step_coefficients(self, timestep) -> Tuple[float, float]:
        """(a, b) with prev = a*sample + b*model_output; epsilon prediction, eta = 0, no clipping."""
        c = self.config
        if c.prediction_type != "epsilon" or c.clip_sample:
            raise NotImplementedError("linear-update form needs epsilon prediction without clip_sample")
        a_t, a_prev = (float(v) for v in self._alphas(int(timestep)))
        a = (a_prev / a_t) ** 0.5
        b = (1 - a_prev) ** 0.5 - (a_prev * (1 - a_t) / a_t) ** 0.5
        return a, b

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating "
                             "the scheduler")
        c = self.config
        t = int(timestep)
        a_t, a_prev = (float(v) for v in self._alphas(t))
        b_t = 1.0 - a_t
        if c.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif c.prediction_type == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif c.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(f"prediction_type given as {c.prediction_type} must be one of `epsilon`, `sample`, or "
                             "`v_prediction`")
        if c.clip_sample:
            x0 = x0.clamp(-c.clip_sample_range, c.clip_sample_range)
        prev_t = t - c.num_train_timesteps // self.num_inference_steps
        std = eta * float(self._get_variance(t, prev_t)) ** 0.5
        if use_clipped_model_output:
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        prev = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * eps
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                             dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return _out(prev, return_dict, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        a = torch.as_tensor(self.alphas_cumprod)[timesteps].to(original_samples.device)
        while a.dim() < original_samples.dim():
            a = a.unsqueeze(-1)
        return a ** 0.5 * original_samples + (1 - a) ** 0.5 * noise

    def __len__(self):
        return self.config.num_train_timesteps


class PNDMScheduler:
    """Pseudo numerical methods for diffusion models: Runge-Kutta warm-up (``step_prk``) + linear multistep (``step_plms``);
    Stable Diffusion uses ``skip_prk_steps=True`` (crowsonkb's PLMS, scheduling_pndm.py:216-223)."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", skip_prk_steps: bool = False, set_alpha_to_one: bool = False,
                 prediction_type: str = "epsilon", timestep_spacing: str = "leading", steps_offset: int = 0):
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k != "self"})
        betas = _make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = np.cumprod(1.0 - betas, dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.pndm_order = 4
        self.cur_model_output, self.counter, self.cur_sample, self.ets = 0, 0, None, []
        self.num_inference_steps = None
        self._timesteps = np.arange(0, num_train_timesteps)[::-1].copy()
        self.prk_timesteps = self.plms_timesteps = self.timesteps = None

    def set_timesteps(self, num_inference_steps: int):
        c = self.config
        self.num_inference_steps = n = num_inference_steps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, c.num_train_timesteps - 1, n).round().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ts = (np.arange(0, n) * (c.num_train_timesteps // n)).round().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            ts = np.round(np.arange(c.num_train_timesteps, 0, -c.num_train_timesteps / n))[::-1].astype(np.int64) - 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported. Please make sure to choose one of 'linspace', "
                             "'leading' or 'trailing'.")
        self._timesteps = ts
        if c.skip_prk_steps:
            self.prk_timesteps = np.array([], dtype=np.int64)
            self.plms_timesteps = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        else:
            prk = np.array(ts[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, c.num_train_timesteps // n // 2]),
                                                                        self.pndm_order)
            self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = ts[:-3][::-1].copy()
        self.timesteps = np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64)
        self.ets, self.counter, self.cur_model_output = [], 0, 0

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def step(self, model_output, timestep, sample, return_dict: bool = True):
        if self.counter < len(self.prk_timesteps) and not self.config.skip_prk_steps:
            return self.step_prk(model_output, timestep, sample, return_dict)
        return self.step_plms(model_output, timestep, sample, return_dict)

    def _require_steps(self):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")

    def step_prk(self, model_output, timestep, sample, return_dict: bool = True):
        self._require_steps()
        n, T = self.num_inference_steps, self.config.num_train_timesteps
        timestep = int(timestep)
        prev_timestep = timestep - (0 if self.counter % 2 else T // n // 2)
        timestep = int(self.prk_timesteps[self.counter // 4 * 4])
        if self.counter % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 6 * model_output
            self.ets.append(model_output)
            self.cur_sample = sample
        elif (self.counter - 1) % 4 == 0 or (self.counter - 2) % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
        else:
            model_output = self.cur_model_output + 1 / 6 * model_output
            self.cur_model_output = 0
        cur_sample = self.cur_sample if self.cur_sample is not None else sample
        prev = self._get_prev_sample(cur_sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return _out(prev, return_dict)

    def step_plms(self, model_output, timestep, sample, return_dict: bool = True):
        self._require_steps()
        if not self.config.skip_prk_steps and len(self.ets) < 3:
            raise ValueError(f"{self.__class__} can only be run AFTER scheduler has been run in 'prk' mode for at least 12 "
                             "iterations")
        step = self.config.num_train_timesteps // self.num_inference_steps
        timestep = int(timestep)
        prev_timestep = timestep - step
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep, timestep = timestep, timestep + step
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            model_output = (model_output + e[-1]) / 2
            sample, self.cur_sample = self.cur_sample, None
        elif len(e) == 2:
            model_output = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            model_output = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
        prev = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return _out(prev, return_dict)

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        """formula (9) of the PNDM paper (scheduling_pndm.py:410-453), coefficients in float32 like the reference's tables"""
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.config.prediction_type == "v_prediction":
            model_output = float(a_t ** 0.5) * model_output + float(b_t ** 0.5) * sample
        elif self.config.prediction_type != "epsilon":
            raise ValueError(f"prediction_type given as {self.config.prediction_type} must be one of `epsilon` or `v_prediction`")
        coeff = float((a_prev / a_t) ** 0.5)
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return coeff * sample - float((a_prev - a_t) / denom) * model_output

    def add_noise(self, original_samples, noise, timesteps):
        a = torch.as_tensor(self.alphas_cumprod)[timesteps].to(original_samples.device)
        while a.dim() < original_samples.dim():
            a = a.unsqueeze(-1)
        return a ** 0.5 * original_samples + (1 - a) ** 0.5 * noise

    def __len__(self):
        return self.config.num_train_timesteps


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", prediction_type: str = "epsilon", interpolation_type: str = "linear",
                 use_karras_sigmas: bool = False, timestep_spacing: str = "linspace", steps_offset: int = 0):
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k != "self"})
        betas = _make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = np.cumprod(1.0 - betas, dtype=np.float32)
        self._train_sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.sigmas = np.concatenate([self._train_sigmas[::-1], [0.0]]).astype(np.float32)
        self.timesteps = np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].astype(
            np.float32)
        self.num_inference_steps = None
        self._step_index: Optional[int] = None

    @property
    def init_noise_sigma(self):
        m = float(self.sigmas.max())
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return m
        return (m ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int):
        c = self.config
        self.num_inference_steps = n = num_inference_steps
        T = c.num_train_timesteps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, n, dtype=np.float32)[::-1].copy()
        elif c.timestep_spacing == "leading":
            ts = (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.float32) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            ts = (np.arange(T, 0, -T / n)).round().copy().astype(np.float32) - 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported")
        sig_all = np.array(self._train_sigmas)
        log_sig = np.log(sig_all)
        if c.interpolation_type == "linear":
            sig = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
        elif c.interpolation_type == "log_linear":
            sig = np.exp(np.linspace(np.log(sig_all[-1]), np.log(sig_all[0]), n + 1))
        else:
            raise ValueError(f"{c.interpolation_type} is not implemented")
        if c.use_karras_sigmas:
            rho = 7.0
            lo, hi = sig[-1] ** (1 / rho), sig[0] ** (1 / rho)
            sig = (hi + np.linspace(0, 1, n) * (lo - hi)) ** rho
            ts = np.array([self._sigma_to_t(s, log_sig) for s in sig])
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = ts.astype(np.float32)
        self._step_index = None

    @staticmethod
    def _sigma_to_t(sigma, log_sigmas):
        log_sigma = np.log(np.maximum(sigma, 1e-10))
        dists = log_sigma - log_sigmas[:, np.newaxis]
        low_idx = np.cumsum((dists >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = log_sigmas[low_idx], log_sigmas[high_idx]
        w = np.clip((low - log_sigma) / (low - high), 0, 1)
        return ((1 - w) * low_idx + w * high_idx).reshape(np.shape(sigma))

    def _init_step_index(self, timestep):
        t = float(timestep)
        idx = np.nonzero(self.timesteps == np.float32(t))[0]
        # "the sigma index that is taken for the **very** first step is always the second index" (duplicates)
        self._step_index = int(idx[1] if len(idx) > 1 else idx[0])

    def scale_model_input(self, sample, timestep):
        if self._step_index is None:
            self._init_step_index(timestep)
        return sample / ((float(self.sigmas[self._step_index]) ** 2 + 1) ** 0.5)

    def model_input_scale(self, timestep) -> float:
        if self._step_index is None:
            self._init_step_index(timestep)
        return 1.0 / ((float(self.sigmas[self._step_index]) ** 2 + 1) ** 0.5)

    def step_coefficients(self, timestep) -> Tuple[float, float]:
        """(a, b) with prev = a*sample + b*model_output for epsilon prediction, s_churn = 0; advances the index."""
        if self.config.prediction_type != "epsilon":
            raise NotImplementedError("linear-update form is for epsilon prediction")
        if self._step_index is None:
            self._init_step_index(timestep)
        s, s_next = float(self.sigmas[self._step_index]), float(self.sigmas[self._step_index + 1])
        self._step_index += 1
        return 1.0, s_next - s

    def step(self, model_output, timestep, sample, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None, return_dict: bool = True):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = float(self.sigmas[self._step_index])
        gamma = min(s_churn / (len(self.sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= sigma <= s_tmax else 0.0
        sigma_hat = sigma * (gamma + 1)
        if gamma > 0:
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                dtype=model_output.dtype)
            sample = sample + noise * s_noise * (sigma_hat ** 2 - sigma ** 2) ** 0.5
        pt = self.config.prediction_type
        if pt in ("original_sample", "sample"):
            x0 = model_output
        elif pt == "epsilon":
            x0 = sample - sigma_hat * model_output
        elif pt == "v_prediction":
            x0 = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (sample / (sigma ** 2 + 1))
        else:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, or `v_prediction`")
        derivative = (sample - x0) / sigma_hat
        dt = float(self.sigmas[self._step_index + 1]) - sigma_hat
        prev = sample + derivative * dt
        self._step_index += 1
        return _out(prev, return_dict, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        """scheduling_euler_discrete.py:480-500: x + noise * sigma[index of each timestep in the current schedule]"""
        ts = np.atleast_1d(np.asarray(timesteps.detach().cpu() if torch.is_tensor(timesteps) else timesteps, dtype=np.float32))
        idx = [int(np.nonzero(self.timesteps == t)[0][0]) for t in ts]
        sigma = torch.as_tensor(np.asarray(self.sigmas, dtype=np.float32)[idx]).to(original_samples.device,
                                                                                  original_samples.dtype)
        while sigma.dim() < original_samples.dim():
            sigma = sigma.unsqueeze(-1)
        return original_samples + noise * sigma

    def __len__(self):
        return self.config.num_train_timesteps


class DPMSolverMultistepScheduler:
    """The fast multistep solver most SD pipelines switch to (20-25 steps; the reference's DiT example uses it). Implemented:
    ``algorithm_type`` "dpmsolver++" / "dpmsolver", ``solver_order`` 1-2, "midpoint" / "heun", epsilon / v_prediction / sample,
    ``lower_order_final``, ``euler_at_final``, Karras sigmas. Not implemented (NotImplementedError): the SDE variants,
    third order, thresholding, ``use_lu_lambdas``, a finite ``lambda_min_clipped``."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", solver_order: int = 2, prediction_type: str = "epsilon",
                 thresholding: bool = False, dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0,
                 algorithm_type: str = "dpmsolver++", solver_type: str = "midpoint", lower_order_final: bool = True,
                 euler_at_final: bool = False, use_karras_sigmas: bool = False, use_lu_lambdas: bool = False,
                 lambda_min_clipped: float = -float("inf"), variance_type: Optional[str] = None,
                 timestep_spacing: str = "linspace", steps_offset: int = 0):
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k != "self"})
        if algorithm_type not in ("dpmsolver", "dpmsolver++"):
            raise NotImplementedError(f"{algorithm_type} is not implemented for {self.__class__}")
        if solver_type not in ("midpoint", "heun"):
            raise NotImplementedError(f"{solver_type} is not implemented for {self.__class__}")
        if solver_order not in (1, 2) or thresholding or use_lu_lambdas or lambda_min_clipped != -float("inf") or variance_type:
            raise NotImplementedError("DPMSolverMultistepScheduler(mi355x): solver_order 3, thresholding, use_lu_lambdas, "
                                      "lambda_min_clipped and learned variance are not implemented")
        betas = _make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = np.cumprod(1.0 - betas, dtype=np.float32)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32)[::-1].copy().astype(np.int64)
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int):
        c, n = self.config, num_inference_steps
        last = c.num_train_timesteps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, last - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ts = (np.arange(0, n + 1) * (last // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            ts = np.arange(last, 0, -c.num_train_timesteps / n).round().copy().astype(np.int64) - 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported. Please make sure to choose one of 'linspace', "
                             "'leading' or 'trailing'.")
        sigmas = np.array(((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5)
        log_sigmas = np.log(sigmas)
        if c.use_karras_sigmas:
            s = np.flip(sigmas).copy()
            rho, ramp = 7.0, np.linspace(0, 1, n)
            s = (s[0] ** (1 / rho) + ramp * (s[-1] ** (1 / rho) - s[0] ** (1 / rho))) ** rho
            ts = np.array([EulerDiscreteScheduler._sigma_to_t(x, log_sigmas) for x in s]).round()
            sigmas = np.concatenate([s, s[-1:]]).astype(np.float32)
        else:
            sigmas = np.interp(ts, np.arange(0, len(sigmas)), sigmas)
            sigma_last = ((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5
            sigmas = np.concatenate([sigmas, [sigma_last]]).astype(np.float32)
        self.sigmas, self.timesteps = sigmas, ts.astype(np.int64)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * c.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    @staticmethod
    def _sigma_to_alpha_sigma_t(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def convert_model_output(self, model_output, sample):
        """model output -> x0 prediction (dpmsolver++) or epsilon prediction (dpmsolver), :407-506"""
        a, s = (float(v) for v in self._sigma_to_alpha_sigma_t(self.sigmas[self._step_index]))
        pt = self.config.prediction_type
        if pt not in ("epsilon", "sample", "v_prediction"):
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction` for the "
                             "DPMSolverMultistepScheduler.")
        if self.config.algorithm_type == "dpmsolver++":
            return (sample - s * model_output) / a if pt == "epsilon" else model_output if pt == "sample" else a * sample - s * model_output
        return model_output if pt == "epsilon" else (sample - a * model_output) / s if pt == "sample" else a * model_output + s * sample

    def _init_step_index(self, timestep):
        c = np.nonzero(self.timesteps == int(timestep))[0]
        self._step_index = len(self.timesteps) - 1 if len(c) == 0 else int(c[1] if len(c) > 1 else c[0])

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._init_step_index(timestep)
        c, n, i = self.config, len(self.timesteps), self._step_index
        lower_order_final = i == n - 1 and (c.euler_at_final or (c.lower_order_final and n < 15))
        lower_order_second = i == n - 2 and c.lower_order_final and n < 15
        m = self.convert_model_output(model_output, sample)
        self.model_outputs = self.model_outputs[1:] + [m]
        at, st = self._sigma_to_alpha_sigma_t(self.sigmas[i + 1])
        a0, s0 = self._sigma_to_alpha_sigma_t(self.sigmas[i])
        lt, l0 = np.log(at) - np.log(st), np.log(a0) - np.log(s0)
        h = lt - l0
        pp = c.algorithm_type == "dpmsolver++"
        f = float
        if c.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final:    # first-order update (:508-575)
            prev = (f(st / s0) * sample - f(at * (np.exp(-h) - 1.0)) * m) if pp else (f(at / a0) * sample - f(st * (np.exp(h) - 1.0)) * m)
        else:                                                                          # second-order multistep (:577-698)
            a1, s1 = self._sigma_to_alpha_sigma_t(self.sigmas[i - 1])
            h_0 = l0 - (np.log(a1) - np.log(s1))
            with np.errstate(divide="ignore"):   # Karras schedules repeat the last sigma: h = 0 -> 1 / r0 = 0, like the reference
                r_inv = f(1.0 / (h_0 / h))
            d0, d1 = self.model_outputs[-1], r_inv * (self.model_outputs[-1] - self.model_outputs[-2])
            if pp:
                prev = f(st / s0) * sample - f(at * (np.exp(-h) - 1.0)) * d0
                prev = prev - f(0.5 * at * (np.exp(-h) - 1.0)) * d1 if c.solver_type == "midpoint" else \
                    prev + f(at * ((np.exp(-h) - 1.0) / h + 1.0)) * d1
            else:
                prev = f(at / a0) * sample - f(st * (np.exp(h) - 1.0)) * d0
                prev = prev - f(0.5 * st * (np.exp(h) - 1.0)) * d1 if c.solver_type == "midpoint" else \
                    prev - f(st * ((np.exp(h) - 1.0) / h - 1.0)) * d1
        _ = lower_order_second   # (with solver_order <= 2 the second-order branch is taken whenever history exists)
        if self.lower_order_nums < c.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return _out(prev, return_dict)

    def add_noise(self, original_samples, noise, timesteps):
        a = torch.as_tensor(self.alphas_cumprod)[timesteps].to(original_samples.device)
        while a.dim() < original_samples.dim():
            a = a.unsqueeze(-1)
        return a ** 0.5 * original_samples + (1 - a) ** 0.5 * noise

    def __len__(self):
        return self.config.num_train_timesteps


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift)
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = ts / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigmas = sig.astype(np.float32)
        self.timesteps = self.sigmas * num_train_timesteps
        self.sigma_min, self.sigma_max = float(self.sigmas[-1]), float(self.sigmas[0])
        self._step_index = None
        self.init_noise_sigma = 1.0

    def set_timesteps(self, num_inference_steps: int):
        T, shift = self.config.num_train_timesteps, self.config.shift
        ts = np.linspace(self.sigma_max * T, self.sigma_min * T, num_inference_steps)
        sig = ts / T
        sig = (shift * sig / (1 + (shift - 1) * sig)).astype(np.float32)
        self.timesteps = sig * T
        self.sigmas = np.concatenate([sig, np.zeros(1, np.float32)])
        self.num_inference_steps = num_inference_steps
        self._step_index = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict: bool = True, **unused):
        if self._step_index is None:
            idx = np.nonzero(self.timesteps == np.float32(float(timestep)))[0]
            self._step_index = int(idx[1] if len(idx) > 1 else idx[0])
        s, s_next = float(self.sigmas[self._step_index]), float(self.sigmas[self._step_index + 1])
        prev = (sample.to(torch.float32) + (s_next - s) * model_output.to(torch.float32)).to(model_output.dtype)
        self._step_index += 1
        return _out(prev, return_dict)


class LCMScheduler:
    """Multistep consistency sampling for latent-consistency models / LCM-LoRA (scheduling_lcm.py:140-631): the schedule is a
    subset of the ``original_inference_steps`` distillation schedule, each step predicts x0, applies the boundary-condition
    scalings (c_skip, c_out) and -- except on the last step -- re-noises to the next timestep with fresh noise."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", original_inference_steps: int = 50, clip_sample: bool = False,
                 clip_sample_range: float = 1.0, set_alpha_to_one: bool = True, steps_offset: int = 0,
                 prediction_type: str = "epsilon", timestep_spacing: str = "leading", timestep_scaling: float = 10.0):
        self.config = SimpleNamespace(**{k: v for k, v in locals().items() if k != "self"})
        betas = _make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = np.cumprod(1.0 - betas, dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64)
        self.custom_timesteps = False
        self._step_index: Optional[int] = None

    @property
    def step_index(self):
        return self._step_index

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: Optional[int] = None, original_inference_steps: Optional[int] = None,
                      timesteps=None, strength: float = 1.0):
        c = self.config
        if num_inference_steps is None and timesteps is None:
            raise ValueError("Must pass exactly one of `num_inference_steps` or `custom_timesteps`.")
        if num_inference_steps is not None and timesteps is not None:
            raise ValueError("Can only pass one of `num_inference_steps` or `custom_timesteps`.")
        original_steps = original_inference_steps if original_inference_steps is not None else c.original_inference_steps
        if original_steps > c.num_train_timesteps:
            raise ValueError(f"`original_steps`: {original_steps} cannot be larger than `self.config.train_timesteps`: "
                             f"{c.num_train_timesteps}")
        k = c.num_train_timesteps // original_steps                        # the paper's skipping step
        origin = np.arange(1, int(original_steps * strength) + 1) * k - 1    # distillation schedule, ascending
        if timesteps is not None:
            ts = np.array(timesteps, dtype=np.int64)
            if np.any(ts[1:] >= ts[:-1]):
                raise ValueError("`custom_timesteps` must be in descending order.")
            if ts[0] >= c.num_train_timesteps:
                raise ValueError(f"`timesteps` must start before `self.config.train_timesteps`: {c.num_train_timesteps}.")
            self.num_inference_steps, self.custom_timesteps = len(ts), True
            init = min(int(self.num_inference_steps * strength), self.num_inference_steps)
            ts = ts[max(self.num_inference_steps - init, 0) * self.order:]
        else:
            if num_inference_steps > c.num_train_timesteps:
                raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                                 f"`self.config.train_timesteps`: {c.num_train_timesteps}")
            if len(origin) // num_inference_steps < 1:
                raise ValueError(f"The combination of `original_steps x strength`: {original_steps} x {strength} is smaller "
                                 f"than `num_inference_steps`: {num_inference_steps}.")
            if num_inference_steps > original_steps:
                raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                                 f"`original_inference_steps`: {original_steps}")
            self.num_inference_steps, self.custom_timesteps = num_inference_steps, False
            rev = origin[::-1]
            idx = np.floor(np.linspace(0, len(rev), num=num_inference_steps, endpoint=False)).astype(np.int64)
            ts = rev[idx]
        self.timesteps = ts.astype(np.int64)
        self._step_index = None

    def _init_step_index(self, timestep):
        idx = np.nonzero(self.timesteps == int(timestep))[0]
        self._step_index = int(idx[1] if len(idx) > 1 else idx[0])

    def get_scalings_for_boundary_condition_discrete(self, timestep):
        sigma_data = 0.5
        st = float(timestep) * self.config.timestep_scaling
        return sigma_data ** 2 / (st ** 2 + sigma_data ** 2), st / (st ** 2 + sigma_data ** 2) ** 0.5

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True, *, noise=None):
        """-> prev_sample (and ``denoised``, the x0 estimate the LCM pipeline decodes). ``noise`` (extension): the re-noising
        draw, otherwise ``torch.randn`` from ``generator``."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating "
                             "the scheduler")
        if self._step_index is None:
            self._init_step_index(timestep)
        c, t = self.config, int(timestep)
        nxt = self._step_index + 1
        prev_t = int(self.timesteps[nxt]) if nxt < len(self.timesteps) else t
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod)
        c_skip, c_out = self.get_scalings_for_boundary_condition_discrete(t)
        if c.prediction_type == "epsilon":
            x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        elif c.prediction_type == "sample":
            x0 = model_output
        elif c.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - (1 - a_t) ** 0.5 * model_output
        else:
            raise ValueError(f"prediction_type given as {c.prediction_type} must be one of `epsilon`, `sample` or "
                             "`v_prediction` for `LCMScheduler`.")
        if c.clip_sample:
            x0 = x0.clamp(-c.clip_sample_range, c.clip_sample_range)
        denoised = c_out * x0 + c_skip * sample
        if self._step_index != self.num_inference_steps - 1:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                    dtype=denoised.dtype)
            prev = a_prev ** 0.5 * denoised + (1 - a_prev) ** 0.5 * noise
        else:
            prev = denoised
        self._step_index += 1
        if not return_dict:
            return (prev, denoised)
        return SimpleNamespace(prev_sample=prev, denoised=denoised)

    def add_noise(self, original_samples, noise, timesteps):
        a = torch.as_tensor(self.alphas_cumprod)[torch.as_tensor(timesteps).long().cpu()].to(original_samples.device)
        while a.dim() < original_samples.dim():
            a = a.unsqueeze(-1)
        return a ** 0.5 * original_samples + (1 - a) ** 0.5 * noise

    def __len__(self):
        return self.config.num_train_timesteps
