"""The product schedulers (paddlemix_amd/schedulers.py) against the reference's own RNG-free golden values
(ppdiffusers/tests/schedulers/test_scheduler_ddim.py:68,121-190, test_scheduler_euler.py:84-163, fixtures
test_schedulers.py:261-303) and against the numpy oracle; plus the linear-update form used by the HIP axpby path."""
import numpy as np
import pytest
import torch

from oracle import schedulers_ref as S
from paddlemix_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler, FlowMatchEulerDiscreteScheduler


def dummy_sample_deter():
    n = 4 * 3 * 8 * 8
    return torch.arange(n, dtype=torch.float32).reshape(3, 8, 8, 4).div(n).permute(3, 0, 1, 2).contiguous()


def dummy_noise_deter():
    n = 4 * 3 * 8 * 8
    return torch.arange(n, dtype=torch.float32).flip(0).reshape(3, 8, 8, 4).div(n).permute(3, 0, 1, 2).contiguous()


def dummy_model(sample, t):
    t = float(t)
    return sample * np.float32(t) / np.float32(t + 1)


DDIM_CFG = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=True)
EULER_CFG = dict(num_train_timesteps=1100, beta_start=0.0001, beta_end=0.02, beta_schedule="linear")


def test_ddim_golden():
    sch = DDIMScheduler(**dict(DDIM_CFG, steps_offset=1))
    sch.set_timesteps(5)
    assert list(sch.timesteps) == [801, 601, 401, 201, 1]
    sch = DDIMScheduler(**DDIM_CFG)
    for (a, b), v in {(0, 0): 0.0, (420, 400): 0.14771, (980, 960): 0.32460, (487, 486): 0.00979, (999, 998): 0.02}.items():
        assert abs(float(sch._get_variance(a, b)) - v) < 1e-5
    for kw, (gs, gm) in [({}, (172.0067, 0.223967)), ({"prediction_type": "v_prediction"}, (52.5302, 0.0684)),
                         ({"set_alpha_to_one": True, "beta_start": 0.01}, (149.8295, 0.1951)),
                         ({"set_alpha_to_one": False, "beta_start": 0.01}, (149.0784, 0.1941))]:
        sch = DDIMScheduler(**dict(DDIM_CFG, **kw))
        sch.set_timesteps(10)
        x = dummy_sample_deter()
        for t in sch.timesteps:
            x = sch.step(dummy_model(x, t), t, x, 0.0).prev_sample
        assert abs(x.abs().sum().item() - gs) < 1e-2 and abs(x.abs().mean().item() - gm) < 1e-3, kw
    sch = DDIMScheduler(**DDIM_CFG)
    sch.set_timesteps(10)
    ts = sch.timesteps[8:]
    x = sch.add_noise(dummy_sample_deter(), dummy_noise_deter(), torch.as_tensor(ts[:1]))
    for t in ts:
        x = sch.step(dummy_model(x, t), t, x, 0.0).prev_sample
    assert abs(x.abs().sum().item() - 354.5418) < 1e-2 and abs(x.abs().mean().item() - 0.4616) < 1e-3


def test_euler_golden():
    for kw, (gs, gm) in [({}, (10.0807, 0.0131)), ({"prediction_type": "v_prediction"}, (0.0002, 2.2676e-06)),
                         ({"use_karras_sigmas": True}, (124.52299499511719, 0.16213932633399963))]:
        sch = EulerDiscreteScheduler(**dict(EULER_CFG, **kw))
        sch.set_timesteps(10)
        x = dummy_sample_deter() * sch.init_noise_sigma
        for t in sch.timesteps:
            x = sch.scale_model_input(x, t)
            x = sch.step(dummy_model(x, t), t, x, return_dict=False)[0]
        assert abs(x.abs().sum().item() - gs) < 1e-2 and abs(x.abs().mean().item() - gm) < 1e-3, kw


def test_euler_with_noise_golden():
    """ppdiffusers/tests/schedulers/test_scheduler_euler.py:165-195 (known answer in tests/golden/)"""
    sch = EulerDiscreteScheduler(**EULER_CFG)
    sch.set_timesteps(10)
    ts = sch.timesteps[8:]
    x = sch.add_noise(dummy_sample_deter() * sch.init_noise_sigma, dummy_noise_deter(), ts[:1])
    for t in ts:
        x = sch.scale_model_input(x, t)
        x = sch.step(dummy_model(x, t), t, x, return_dict=False)[0]
    assert abs(x.abs().sum().item() - 57062.9023) < 0.1 and abs(x.abs().mean().item() - 74.3007) < 1e-3


def test_tables_match_oracle_and_linear_update():
    # SDXL bench schedule (tests/pipelines/stable_diffusion_xl/test_stable_diffusion_xl.py:84-90 parameters)
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading", steps_offset=1)
    a, b = EulerDiscreteScheduler(**kw), S.EulerRef(**kw)
    a.set_timesteps(30)
    b.set_timesteps(30)
    assert np.array_equal(a.timesteps, b.timesteps) and np.allclose(a.sigmas, b.sigmas, rtol=1e-6)
    assert abs(a.init_noise_sigma - b.init_noise_sigma) < 1e-5
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g) * a.init_noise_sigma
    x_lin = x.clone()
    lin = EulerDiscreteScheduler(**kw)
    lin.set_timesteps(30)
    for t in a.timesteps:
        eps = torch.randn(x.shape, generator=g)
        x = a.step(eps, t, x).prev_sample
        ca, cb = lin.step_coefficients(t)
        x_lin = ca * x_lin + cb * eps
    assert torch.allclose(x, x_lin, rtol=1e-5, atol=1e-5)
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, clip_sample=False,
              set_alpha_to_one=False)
    d, dr = DDIMScheduler(**kw), S.DDIMRef(**kw)
    d.set_timesteps(20)
    dr.set_timesteps(20)
    assert list(d.timesteps) == list(dr.timesteps)
    x = torch.randn(1, 4, 8, 8, generator=g)
    for t in d.timesteps[:5]:
        eps = torch.randn(x.shape, generator=g)
        ref = torch.from_numpy(dr.step(eps.numpy(), t, x.numpy()))
        ca, cb = d.step_coefficients(t)
        assert torch.allclose(d.step(eps, t, x).prev_sample, ref, atol=1e-5)
        assert torch.allclose(ca * x + cb * eps, ref, atol=1e-5)
        x = ref


def test_flow_match_matches_oracle():
    a, b = FlowMatchEulerDiscreteScheduler(shift=3.0), S.FlowMatchEulerRef(shift=3.0)
    a.set_timesteps(28)
    b.set_timesteps(28)
    assert np.allclose(a.timesteps, b.timesteps) and np.allclose(a.sigmas, b.sigmas)
    x = torch.ones(1, 16, 4, 4)
    v = torch.full_like(x, 0.5)
    t = a.timesteps[0]
    assert torch.allclose(a.step(v, t, x).prev_sample, torch.from_numpy(b.step(v.numpy(), t, x.numpy())))


def _pndm_full_loop(cls, step_fn, **kw):
    """tests/schedulers/test_scheduler_pndm.py:107-125"""
    sch = cls(**{**dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear"), **kw})
    sch.set_timesteps(10)
    x = dummy_sample_deter()
    for t in sch.prk_timesteps:
        x = step_fn(sch.step_prk, dummy_model(x, t), t, x)
    for t in sch.plms_timesteps:
        x = step_fn(sch.step_plms, dummy_model(x, t), t, x)
    return sch, x


def test_pndm_golden_and_oracle():
    """PNDMScheduler against the reference's four full-loop known answers (tests/golden/, harvested from
    test_scheduler_pndm.py:224-256) -- product class and oracle restatement"""
    import json
    import os
    from paddlemix_amd.schedulers import PNDMScheduler
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_known_answers.json")) as f:
        g = json.load(f)["pndm"]["tests"]
    cases = {"test_full_loop_no_noise": {}, "test_full_loop_with_v_prediction": {"prediction_type": "v_prediction"},
             "test_full_loop_with_set_alpha_to_one": {"set_alpha_to_one": True, "beta_start": 0.01},
             "test_full_loop_with_no_set_alpha_to_one": {"set_alpha_to_one": False, "beta_start": 0.01}}
    assert set(cases) == set(g)
    for name, kw in cases.items():
        _, x = _pndm_full_loop(PNDMScheduler, lambda f, mo, t, s: f(mo, t, s, return_dict=False)[0], **kw)
        assert abs(x.abs().sum().item() - g[name]["sum"]["value"]) < g[name]["sum"]["tol"], name
        assert abs(x.abs().mean().item() - g[name]["mean"]["value"]) < g[name]["mean"]["tol"], name
        _, xo = _pndm_full_loop(S.PNDMRef, lambda f, mo, t, s: f(mo.numpy() if torch.is_tensor(mo) else mo, int(t),
                                                               s.numpy() if torch.is_tensor(s) else s), **kw)
        xo = torch.as_tensor(xo)
        assert abs(xo.abs().sum().item() - g[name]["sum"]["value"]) < g[name]["sum"]["tol"], name
    # Stable Diffusion's configuration: skip_prk_steps, scaled_linear betas, steps_offset 1 -> 51 PLMS calls for 50 steps,
    # the second one re-visiting the first timestep (scheduling_pndm.py:216-223)
    sd = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1)
    sd.set_timesteps(50)
    assert len(sd.timesteps) == 51 and sd.timesteps[0] == sd.timesteps[1] + 20 == 981 and sd.timesteps[-1] == 1
    ref = S.PNDMRef(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1)
    ref.set_timesteps(50)
    x = dummy_sample_deter()
    xr = x.numpy().copy()
    for t in sd.timesteps:
        x = sd.step(dummy_model(x, t), t, x).prev_sample
        xr = ref.step(dummy_model(torch.as_tensor(xr), t).numpy(), int(t), xr)
    assert torch.allclose(x, torch.as_tensor(xr), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        PNDMScheduler().step_plms(x, 1, x)


def test_dpm_multistep_golden_and_oracle():
    """DPMSolverMultistepScheduler (dpmsolver++, order 2, midpoint, lower_order_final=False: the reference test's config,
    test_scheduler_dpm_multi.py:29-50) against its full-loop known answers -- product class and oracle restatement"""
    import json
    import os
    from paddlemix_amd.schedulers import DPMSolverMultistepScheduler
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_known_answers.json")) as f:
        g = json.load(f)["dpm_multi"]["tests"]
    base = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", solver_order=2,
                prediction_type="epsilon", algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=False)
    cases = {"test_full_loop_no_noise": {}, "test_full_loop_with_v_prediction": {"prediction_type": "v_prediction"},
             "test_full_loop_with_karras_and_v_prediction": {"prediction_type": "v_prediction", "use_karras_sigmas": True}}
    for name, kw in cases.items():
        sch = DPMSolverMultistepScheduler(**{**base, **kw})
        ref = S.DPMSolverMultistepRef(**{**base, **kw})
        sch.set_timesteps(10)
        ref.set_timesteps(10)
        assert list(sch.timesteps) == list(ref.timesteps) and np.allclose(sch.sigmas, ref.sigmas)
        x = dummy_sample_deter()
        xr = x.numpy().copy()
        for t in sch.timesteps:
            x = sch.step(dummy_model(x, t), t, x).prev_sample
            xr = ref.step(dummy_model(torch.as_tensor(xr), t).numpy(), int(t), xr)
        assert abs(x.abs().mean().item() - g[name]["mean"]["value"]) < g[name]["mean"]["tol"], (name, x.abs().mean().item())
        assert abs(float(np.abs(xr).mean()) - g[name]["mean"]["value"]) < g[name]["mean"]["tol"], name
    # the other deterministic variants agree with the oracle restatement (no known answers in the reference)
    for kw in ({"algorithm_type": "dpmsolver"}, {"solver_type": "heun"}, {"solver_order": 1}, {"lower_order_final": True},
               {"algorithm_type": "dpmsolver", "solver_type": "heun", "prediction_type": "sample"},
               {"timestep_spacing": "leading", "steps_offset": 1}, {"timestep_spacing": "trailing"}):
        sch, ref = DPMSolverMultistepScheduler(**{**base, **kw}), S.DPMSolverMultistepRef(**{**base, **kw})
        sch.set_timesteps(8)
        ref.set_timesteps(8)
        x = dummy_sample_deter()
        xr = x.numpy().copy()
        for t in sch.timesteps:
            x = sch.step(dummy_model(x, t), t, x, return_dict=False)[0]
            xr = ref.step(dummy_model(torch.as_tensor(xr), t).numpy(), int(t), xr)
        assert torch.allclose(x, torch.as_tensor(xr), rtol=2e-5, atol=2e-6), kw
    with pytest.raises(NotImplementedError):
        DPMSolverMultistepScheduler(algorithm_type="sde-dpmsolver++")
    with pytest.raises(NotImplementedError):
        DPMSolverMultistepScheduler(solver_order=3)


def test_lcm_golden_and_oracle():
    """LCMScheduler: the reference's one-step full loop is RNG-free (no re-noising on the last step) and pins product and
    oracle (test_scheduler_lcm.py:239-247, via tests/golden/); the multistep loop is checked product-vs-oracle with the
    re-noising draws supplied (the reference's multistep known answer needs Paddle's generator)."""
    import json
    import os
    from paddlemix_amd.schedulers import LCMScheduler
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_known_answers.json")) as f:
        gold = json.load(f)["lcm"]["tests"]
    cfg = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.0120, beta_schedule="scaled_linear",
               prediction_type="epsilon")
    one = gold["test_full_loop_onestep"]
    sch, ref = LCMScheduler(**cfg), S.LCMRef(**cfg)
    sch.set_timesteps(1)
    ref.set_timesteps(1)
    assert list(sch.timesteps) == [999] == list(ref.timesteps)
    x = dummy_sample_deter()
    out = sch.step(dummy_model(x, 999), 999, x)
    assert torch.equal(out.prev_sample, out.denoised)
    assert abs(out.prev_sample.abs().sum().item() - one["sum"]["value"]) < one["sum"]["tol"]
    assert abs(out.prev_sample.abs().mean().item() - one["mean"]["value"]) < one["mean"]["tol"]
    r, _ = ref.step(dummy_model(x, 999).numpy().astype(np.float64), 999, x.numpy().astype(np.float64))
    assert abs(np.abs(r).sum() - one["sum"]["value"]) < one["sum"]["tol"]
    # schedules: subsets of the 50-step distillation schedule (scheduling_lcm.py:376-379, 458-462)
    for n, want in ((4, [999, 759, 499, 259]), (10, [999, 899, 799, 699, 599, 499, 399, 299, 199, 99]), (2, [999, 499])):
        sch.set_timesteps(n)
        ref.set_timesteps(n)
        assert list(sch.timesteps) == want == list(ref.timesteps)
    sch.set_timesteps(4, strength=0.5)                       # img2img: the schedule is built on original_steps * strength
    assert list(sch.timesteps) == [499, 379, 259, 139]            # indices floor([0, 6.25, 12.5, 18.75]) of 499, 479, ... 19
    sch.set_timesteps(timesteps=[999, 499, 259])
    assert list(sch.timesteps) == [999, 499, 259] and sch.custom_timesteps
    with pytest.raises(ValueError):
        sch.set_timesteps(timesteps=[499, 999])
    with pytest.raises(ValueError):
        sch.set_timesteps(60)                                # more steps than the distillation schedule has
    with pytest.raises(ValueError):
        sch.set_timesteps(4, timesteps=[999])
    # multistep: same draws -> same trajectory as the float64 oracle
    g = torch.Generator().manual_seed(0)
    sch.set_timesteps(10)
    ref.set_timesteps(10)
    x, xr = dummy_sample_deter(), dummy_sample_deter().numpy().astype(np.float64)
    for i, t in enumerate(sch.timesteps):
        nz = torch.randn(x.shape, generator=g)
        out = sch.step(dummy_model(x, t), t, x, noise=nz)
        xr, den = ref.step(dummy_model(torch.from_numpy(xr), t).numpy(), t, xr, None if i == 9 else nz.numpy().astype(np.float64))
        x = out.prev_sample
        assert np.abs(out.denoised.numpy() - den).max() < 1e-4
    assert np.abs(x.numpy() - xr).max() < 1e-4
    # the generator path draws the same noise as an explicit torch.randn with that generator
    sch.set_timesteps(2)
    a = sch.step(dummy_model(x, 999), 999, x, generator=torch.Generator().manual_seed(5), return_dict=False)[0]
    sch.set_timesteps(2)
    b = sch.step(dummy_model(x, 999), 999, x, noise=torch.randn(x.shape, generator=torch.Generator().manual_seed(5))).prev_sample
    assert torch.equal(a, b)
    assert sch.add_noise(x, torch.ones_like(x), torch.tensor([999])).shape == x.shape
