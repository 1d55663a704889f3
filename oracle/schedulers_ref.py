"""numpy restatement of the three schedulers on the hot path (oracle; TEST INFRASTRUCTURE ONLY).

Pinned against the reference's RNG-free full-loop known answers in tests/test_oracle_pins.py
(DDIM: /root/reference/ppdiffusers/tests/schedulers/test_scheduler_ddim.py:68,121-190;
Euler: test_scheduler_euler.py:84-163), and -- all six, FlowMatchEuler (which has no test in the
reference) included -- against the reference's own scheduler classes executed over
oracle/paddle_shim.py in whole sampling loops (tests/test_reference_modules.py, cases sched_*).

Paths relative to /root/reference/ppdiffusers/ppdiffusers/schedulers/.
The reference keeps betas / alphas_cumprod / sigmas as float32 tensors; so do we
(np.float32), because the pinned sums were produced that way.
"""
from __future__ import annotations

import numpy as np


def _betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    """scheduling_ddim.py:199-211 / scheduling_euler_discrete.py:160-172."""
    if beta_schedule == "linear":
        return np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
    if beta_schedule == "scaled_linear":
        return np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
    raise NotImplementedError(beta_schedule)


class DDIMRef:
    """DDIMScheduler: ctor :180-229, set_timesteps :305-348, step :350-475, add_noise :477-."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 clip_sample_range=1.0, timestep_spacing="leading"):
        self.T = num_train_timesteps
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = np.cumprod((1.0 - self.betas).astype(np.float32), dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.clip_sample, self.clip_sample_range = clip_sample, clip_sample_range
        self.steps_offset, self.prediction_type, self.timestep_spacing = steps_offset, prediction_type, timestep_spacing
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        if self.timestep_spacing == "leading":
            ratio = self.T // n
            ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        elif self.timestep_spacing == "trailing":
            ts = np.round(np.arange(self.T, 0, -self.T / n)).astype(np.int64) - 1
        elif self.timestep_spacing == "linspace":
            ts = np.linspace(0, self.T - 1, n).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(self.timestep_spacing)
        self.timesteps = ts

    def _get_variance(self, t, prev_t):
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)

    def scale_model_input(self, sample, t=None):
        return sample

    def step(self, model_output, t, sample, eta=0.0):
        t = int(t)
        prev_t = t - self.T // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        elif self.prediction_type == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        else:
            raise ValueError(self.prediction_type)
        if self.clip_sample:
            x0 = np.clip(x0, -self.clip_sample_range, self.clip_sample_range)
        std = eta * self._get_variance(t, prev_t) ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * eps
        return (a_prev ** 0.5 * x0 + direction).astype(np.float32)

    def add_noise(self, x, noise, t):
        a = self.alphas_cumprod[int(t)]
        return (a ** 0.5 * x + (1 - a) ** 0.5 * noise).astype(np.float32)


class EulerRef:
    """EulerDiscreteScheduler: ctor :145-200, scale_model_input :216-238, set_timesteps :240-311,
    karras :335-358, step :375-478."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 prediction_type="epsilon", use_karras_sigmas=False, timestep_spacing="linspace", steps_offset=0):
        self.T = num_train_timesteps
        betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)
        self.prediction_type, self.use_karras = prediction_type, use_karras_sigmas
        self.spacing, self.steps_offset = timestep_spacing, steps_offset
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5)[::-1].astype(np.float32)
        self.sigmas = np.concatenate([sig, np.zeros(1, np.float32)])
        self.timesteps = np.linspace(0, self.T - 1, self.T, dtype=float)[::-1].astype(np.float32)
        self.step_index = None

    @property
    def init_noise_sigma(self):
        m = self.sigmas.max()
        return m if self.spacing in ("linspace", "trailing") else (m ** 2 + 1) ** 0.5

    def set_timesteps(self, n):
        self.n = n
        if self.spacing == "linspace":
            ts = np.linspace(0, self.T - 1, n, dtype=np.float32)[::-1].copy()
        elif self.spacing == "leading":
            ts = (np.arange(0, n) * (self.T // n)).round()[::-1].copy().astype(np.float32) + self.steps_offset
        elif self.spacing == "trailing":
            ts = (np.arange(self.T, 0, -self.T / n)).round().copy().astype(np.float32) - 1
        else:
            raise ValueError(self.spacing)
        sig_all = np.array(((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5)
        log_sig = np.log(sig_all)
        sig = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
        if self.use_karras:
            smin, smax, rho = sig[-1], sig[0], 7.0
            ramp = np.linspace(0, 1, n)
            sig = (smax ** (1 / rho) + ramp * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho
            ts = np.array([self._sigma_to_t(s, log_sig) for s in sig])
        self.sigmas = np.concatenate([sig.astype(np.float32), np.zeros(1, np.float32)])
        self.timesteps = ts.astype(np.float32)
        self.step_index = None

    @staticmethod
    def _sigma_to_t(sigma, log_sigmas):
        log_sigma = np.log(np.maximum(sigma, 1e-10))
        dists = log_sigma - log_sigmas[:, np.newaxis]
        low = np.cumsum((dists >= 0), axis=0).argmax(axis=0).clip(max=log_sigmas.shape[0] - 2)
        high = low + 1
        w = np.clip((log_sigmas[low] - log_sigma) / (log_sigmas[low] - log_sigmas[high]), 0, 1)
        return ((1 - w) * low + w * high).reshape(np.shape(sigma))

    def _init_step_index(self, t):
        idx = np.nonzero(self.timesteps == np.float32(t))[0]
        self.step_index = int(idx[1] if len(idx) > 1 else idx[0])

    def scale_model_input(self, sample, t):
        if self.step_index is None:
            self._init_step_index(t)
        s = self.sigmas[self.step_index]
        return (sample / ((s ** 2 + 1) ** 0.5)).astype(np.float32)

    def add_noise(self, x, noise, timesteps):
        """scheduling_euler_discrete.py:480-500: x + noise * sigma[index of each timestep]"""
        idx = [int(np.nonzero(self.timesteps == np.float32(t))[0][0]) for t in np.atleast_1d(timesteps)]
        sigma = self.sigmas[idx].reshape((-1,) + (1,) * (x.ndim - 1))
        return (x + noise * sigma).astype(np.float32)

    def step(self, model_output, t, sample):
        if self.step_index is None:
            self._init_step_index(t)
        s = self.sigmas[self.step_index]
        if self.prediction_type == "epsilon":
            x0 = sample - s * model_output
        elif self.prediction_type == "v_prediction":
            x0 = model_output * (-s / (s ** 2 + 1) ** 0.5) + sample / (s ** 2 + 1)
        elif self.prediction_type in ("sample", "original_sample"):
            x0 = model_output
        else:
            raise ValueError(self.prediction_type)
        d = (sample - x0) / s
        dt = self.sigmas[self.step_index + 1] - s
        self.step_index += 1
        return (sample + d * dt).astype(np.float32)


class FlowMatchEulerRef:
    """FlowMatchEulerDiscreteScheduler (scheduling_flow_match_euler_discrete.py:63-283). Pinned by the live
    reference run (case sched_flow_match_sd3, bit-identical)."""

    def __init__(self, num_train_timesteps=1000, shift=1.0):
        self.T, self.shift = num_train_timesteps, shift
        ts = np.linspace(1, self.T, self.T, dtype=np.float32)[::-1].copy()
        sig = ts / self.T
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigmas = sig.astype(np.float32)
        self.timesteps = self.sigmas * self.T
        self.sigma_min, self.sigma_max = float(self.sigmas[-1]), float(self.sigmas[0])
        self.step_index = None

    def set_timesteps(self, n):
        ts = np.linspace(self.sigma_max * self.T, self.sigma_min * self.T, n)
        sig = ts / self.T
        sig = (self.shift * sig / (1 + (self.shift - 1) * sig)).astype(np.float32)
        self.timesteps = sig * self.T
        self.sigmas = np.concatenate([sig, np.zeros(1, np.float32)])
        self.step_index = None

    def step(self, model_output, t, sample):
        if self.step_index is None:
            idx = np.nonzero(self.timesteps == np.float32(t))[0]
            self.step_index = int(idx[1] if len(idx) > 1 else idx[0])
        sample = sample.astype(np.float32)
        s = self.sigmas[self.step_index]
        denoised = sample - model_output * s
        d = (sample - denoised) / s
        dt = self.sigmas[self.step_index + 1] - s
        self.step_index += 1
        return (sample + d * dt).astype(model_output.dtype)


class PNDMRef:
    """PNDMScheduler (the scheduler SD-1.x checkpoints ship with, ``skip_prk_steps=True``): ctor scheduling_pndm.py:117-176,
    set_timesteps :178-235, step_prk :260-320, step_plms :322-395, _get_prev_sample :410-453. Pinned by the reference's full-loop
    known answers (tests/schedulers/test_scheduler_pndm.py:224-256)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", skip_prk_steps=False,
                 set_alpha_to_one=False, prediction_type="epsilon", timestep_spacing="leading", steps_offset=0):
        self.T = num_train_timesteps
        self.alphas_cumprod = np.cumprod((1.0 - _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)).astype(np.float32),
                                         dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.skip_prk, self.prediction_type = skip_prk_steps, prediction_type
        self.spacing, self.steps_offset = timestep_spacing, steps_offset
        self.init_noise_sigma, self.order = 1.0, 4
        self.n = None

    def set_timesteps(self, n):
        self.n = n
        if self.spacing == "linspace":
            ts = np.linspace(0, self.T - 1, n).round().astype(np.int64)
        elif self.spacing == "leading":
            ts = (np.arange(0, n) * (self.T // n)).round().astype(np.int64) + self.steps_offset
        elif self.spacing == "trailing":
            ts = np.round(np.arange(self.T, 0, -self.T / n))[::-1].astype(np.int64) - 1
        else:
            raise ValueError(self.spacing)
        if self.skip_prk:
            self.prk_timesteps = np.array([], dtype=np.int64)
            self.plms_timesteps = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        else:
            prk = np.array(ts[-self.order:]).repeat(2) + np.tile(np.array([0, self.T // n // 2]), self.order)
            self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = ts[:-3][::-1].copy()
        self.timesteps = np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64)
        self.ets, self.counter, self.cur_model_output, self.cur_sample = [], 0, 0, None

    def scale_model_input(self, sample, t=None):
        return sample

    def step(self, model_output, t, sample):
        if self.counter < len(self.prk_timesteps) and not self.skip_prk:
            return self.step_prk(model_output, t, sample)
        return self.step_plms(model_output, t, sample)

    def step_prk(self, model_output, t, sample):
        diff = 0 if self.counter % 2 else self.T // self.n // 2
        prev_t = t - diff
        t = self.prk_timesteps[self.counter // 4 * 4]
        if self.counter % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 6 * model_output
            self.ets.append(model_output)
            self.cur_sample = sample
        elif (self.counter - 1) % 4 == 0 or (self.counter - 2) % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
        else:
            model_output = self.cur_model_output + 1 / 6 * model_output
            self.cur_model_output = 0
        cur = self.cur_sample if self.cur_sample is not None else sample
        prev = self._get_prev_sample(cur, t, prev_t, model_output)
        self.counter += 1
        return prev

    def step_plms(self, model_output, t, sample):
        if not self.skip_prk and len(self.ets) < 3:
            raise ValueError("PNDM can only be run AFTER scheduler has been run in 'prk' mode for at least 12 iterations")
        prev_t = t - self.T // self.n
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t, t = t, t + self.T // self.n
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
        prev = self._get_prev_sample(sample, t, prev_t, model_output)
        self.counter += 1
        return prev

    def _get_prev_sample(self, sample, t, prev_t, model_output):
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.prediction_type == "v_prediction":
            model_output = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        elif self.prediction_type != "epsilon":
            raise ValueError(self.prediction_type)
        coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return (coeff * sample - (a_prev - a_t) * model_output / denom).astype(np.float32)


class DPMSolverMultistepRef:
    """DPMSolverMultistepScheduler, deterministic variants (scheduling_dpmsolver_multistep.py: ctor :150-218, set_timesteps
    :226-299, convert_model_output :407-506, first / second order updates :508-698, step :802-878). Orders 1-2,
    algorithm_type dpmsolver++ / dpmsolver, midpoint / heun, Karras sigmas. Pinned by the reference's full-loop means
    (tests/schedulers/test_scheduler_dpm_multi.py:229-303)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", solver_order=2,
                 prediction_type="epsilon", algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True,
                 euler_at_final=False, use_karras_sigmas=False, timestep_spacing="linspace", steps_offset=0):
        self.T = num_train_timesteps
        self.alphas_cumprod = np.cumprod((1.0 - _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)).astype(np.float32),
                                         dtype=np.float32)
        self.order, self.prediction_type, self.algorithm, self.solver = solver_order, prediction_type, algorithm_type, solver_type
        self.lower_order_final, self.euler_at_final, self.karras = lower_order_final, euler_at_final, use_karras_sigmas
        self.spacing, self.steps_offset = timestep_spacing, steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        last = self.T   # lambda_min_clipped = -inf
        if self.spacing == "linspace":
            ts = np.linspace(0, last - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.spacing == "leading":
            ts = (np.arange(0, n + 1) * (last // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        elif self.spacing == "trailing":
            ts = np.arange(last, 0, -self.T / n).round().copy().astype(np.int64) - 1
        else:
            raise ValueError(self.spacing)
        sig = np.array(((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5)
        log_sig = np.log(sig)
        if self.karras:
            s = np.flip(sig).copy()
            rho, ramp = 7.0, np.linspace(0, 1, n)
            s = (s[0] ** (1 / rho) + ramp * (s[-1] ** (1 / rho) - s[0] ** (1 / rho))) ** rho
            ts = np.array([EulerRef._sigma_to_t(x, log_sig) for x in s]).round()
            sig = np.concatenate([s, s[-1:]]).astype(np.float32)
        else:
            sig = np.interp(ts, np.arange(0, len(sig)), sig)
            sig = np.concatenate([sig, [((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5]]).astype(np.float32)
        self.sigmas, self.timesteps = sig, ts.astype(np.int64)
        self.outs, self.lower, self.idx = [None] * self.order, 0, None

    def scale_model_input(self, sample, t=None):
        return sample

    @staticmethod
    def _as(sigma):
        a = 1 / ((sigma ** 2 + 1) ** 0.5)
        return a, sigma * a

    def _convert(self, m, x):
        a, s = self._as(self.sigmas[self.idx])
        if self.algorithm == "dpmsolver++":   # data prediction
            return {"epsilon": (x - s * m) / a, "sample": m, "v_prediction": a * x - s * m}[self.prediction_type]
        return {"epsilon": m, "sample": (x - a * m) / s, "v_prediction": a * m + s * x}[self.prediction_type]   # noise prediction

    def step(self, model_output, t, sample):
        if self.idx is None:
            c = np.nonzero(self.timesteps == t)[0]
            self.idx = len(self.timesteps) - 1 if len(c) == 0 else int(c[1] if len(c) > 1 else c[0])
        n = len(self.timesteps)
        final1 = self.idx == n - 1 and (self.euler_at_final or (self.lower_order_final and n < 15))
        final2 = self.idx == n - 2 and self.lower_order_final and n < 15
        m = self._convert(model_output, sample)
        self.outs = self.outs[1:] + [m]
        at, st = self._as(self.sigmas[self.idx + 1])
        a0, s0 = self._as(self.sigmas[self.idx])
        lt, l0 = np.log(at) - np.log(st), np.log(a0) - np.log(s0)
        h = lt - l0
        pp = self.algorithm == "dpmsolver++"
        if self.order == 1 or self.lower < 1 or final1:
            x = (st / s0) * sample - (at * (np.exp(-h) - 1.0)) * m if pp else (at / a0) * sample - (st * (np.exp(h) - 1.0)) * m
        elif self.order == 2 or self.lower < 2 or final2:
            a1, s1 = self._as(self.sigmas[self.idx - 1])
            h0 = l0 - (np.log(a1) - np.log(s1))
            with np.errstate(divide="ignore"):   # Karras schedules repeat the last sigma: h = 0 -> 1 / r0 = 0, like the reference
                d0, d1 = self.outs[-1], (1.0 / (h0 / h)) * (self.outs[-1] - self.outs[-2])
            if pp:
                x = (st / s0) * sample - (at * (np.exp(-h) - 1.0)) * d0
                x = x - 0.5 * (at * (np.exp(-h) - 1.0)) * d1 if self.solver == "midpoint" else x + (at * ((np.exp(-h) - 1.0) / h + 1.0)) * d1
            else:
                x = (at / a0) * sample - (st * (np.exp(h) - 1.0)) * d0
                x = x - 0.5 * (st * (np.exp(h) - 1.0)) * d1 if self.solver == "midpoint" else x - (st * ((np.exp(h) - 1.0) / h - 1.0)) * d1
        else:
            raise NotImplementedError("third-order update not restated")
        if self.lower < self.order:
            self.lower += 1
        self.idx += 1
        return x.astype(np.float32)


class LCMRef:
    """LCMScheduler (scheduling_lcm.py: schedule :362-464, boundary scalings :468-474, step :513-566), numpy float64 state,
    epsilon prediction, no clipping. ``step`` takes the re-noising draw explicitly (None on the last step)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 original_inference_steps=50, set_alpha_to_one=True, timestep_scaling=10.0, **_unused):
        self.n_train, self.original, self.scaling = num_train_timesteps, original_inference_steps, timestep_scaling
        if beta_schedule == "scaled_linear":
            betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
        else:
            betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
        self.alphas_cumprod = np.cumprod(1.0 - betas, dtype=np.float32)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, n, strength=1.0):
        k = self.n_train // self.original
        origin = [i * k - 1 for i in range(1, int(self.original * strength) + 1)][::-1]
        self.timesteps = np.array([origin[int(np.floor(j * len(origin) / n))] for j in range(n)], dtype=np.int64)
        self._i = 0

    def step(self, eps, t, x, noise=None):
        t = int(t)
        last = self._i == len(self.timesteps) - 1
        prev_t = t if last else int(self.timesteps[self._i + 1])
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        st = t * self.scaling
        c_skip, c_out = 0.25 / (st * st + 0.25), st / np.sqrt(st * st + 0.25)
        x0 = (x - np.sqrt(1 - a_t) * eps) / np.sqrt(a_t)
        denoised = c_out * x0 + c_skip * x
        self._i += 1
        if last:
            return denoised, denoised
        return np.sqrt(a_prev) * denoised + np.sqrt(1 - a_prev) * noise, denoised
