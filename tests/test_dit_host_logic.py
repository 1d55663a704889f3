"""CPU: the class-conditional DiT program (per-block conditioning embedders batched into one GEMM + one gather, adaLN-Zero,
gated residual epilogues, learned-sigma un-patchify) interpreted by the ABI emulator against the oracle."""
import pytest
import torch

from oracle import dit_ref as R
from paddlemix_amd.dit import DiTTransformer2DModel, dit_param_shapes, synth_dit_params
from tests.abi_emulator import Emulator, on_emulator
from tests.configs import DIT_XL2, MINI_DIT


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def test_shapes_and_synth_match_oracle():
    for cfg in (MINI_DIT, DIT_XL2):
        assert list(dit_param_shapes(cfg).items()) == list(R.dit_param_shapes(cfg).items())
    n = sum(torch.Size(s).numel() for s in dit_param_shapes(DIT_XL2).values())
    # DiT-XL/2 as the reference builds it: every block owns its conditioning embedder (28 x (256 x 1152 + 1152^2 + 1001 x 1152))
    assert n == 28 * (256 * 1152 + 1152 + 1152 * 1152 + 1152 + 1001 * 1152 + 1152 * 6912 + 6912 + 4 * (1152 * 1152 + 1152)
                      + 1152 * 4608 + 4608 + 4608 * 1152 + 1152) + (1152 * 16 + 1152) + (1152 * 2304 + 2304) + (1152 * 32 + 32)
    a, b = synth_dit_params(MINI_DIT, 3), R.synth_dit_params(MINI_DIT, 3)
    assert all(torch.equal(a[k], b[k]) for k in b)


@pytest.mark.parametrize("B,side", [(2, 16), (1, 8), (3, 32)])
def test_program_matches_oracle(B, side):
    cfg = MINI_DIT
    P = synth_dit_params(cfg, seed=21)
    Pb = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 4, side, side, generator=g)
    labels = torch.randint(0, 11, (B,), generator=g)        # incl. the CFG null class (index num_embeds_ada_norm)
    t = torch.tensor([999.0, 500.0, 3.0])[:B]
    ref = R.dit_forward(Pb, cfg, x, t, labels)
    model = on_emulator(DiTTransformer2DModel, cfg, P)
    out = model(x, timestep=t, class_labels=labels, return_dict=False)[0]
    assert out.shape == ref.shape == (B, 8, side, side) and out.dtype == torch.float32
    assert _rel(out, ref) < 2e-2, _rel(out, ref)
    assert torch.equal(out, model(x, timestep=t, class_labels=labels).sample)
    # a scalar timestep broadcasts over the batch (pipeline_dit.py:180-190)
    ref1 = R.dit_forward(Pb, cfg, x, 250, labels)
    assert _rel(model(x, timestep=250, class_labels=labels).sample, ref1) < 2e-2


def test_errors():
    P = synth_dit_params(MINI_DIT, seed=1)
    model = on_emulator(DiTTransformer2DModel, MINI_DIT, P)
    x = torch.randn(1, 4, 16, 16)
    with pytest.raises(ValueError, match="timestep"):
        model(x, class_labels=torch.tensor([1]))
    with pytest.raises(ValueError, match="square"):
        model(torch.randn(1, 4, 16, 8), timestep=1, class_labels=torch.tensor([1]))
    with pytest.raises(NotImplementedError):
        on_emulator(DiTTransformer2DModel, dict(MINI_DIT, norm_type="layer_norm"), P)
    with pytest.raises(KeyError):
        on_emulator(DiTTransformer2DModel, MINI_DIT, {k: v for k, v in P.items() if k != "proj_out_2.bias"})


def test_dit_denoiser_loop_matches_reference_semantics():
    """DiTPipeline.__call__ (pipeline_dit.py:158-233): null-class CFG on the duplicated half, epsilon-only guidance,
    learned sigma dropped before the scheduler step -- against the same loop written on the oracle"""
    from oracle import schedulers_ref as S
    from paddlemix_amd.pipeline import DiTDenoiser
    from paddlemix_amd.schedulers import DDIMScheduler
    cfg = MINI_DIT
    P = synth_dit_params(cfg, seed=8)
    Pb = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}
    model = on_emulator(DiTTransformer2DModel, cfg, P)
    kw = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=False)
    lat0 = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(4))
    labels = torch.tensor([3, 7])
    out = DiTDenoiser(model, DDIMScheduler(**kw))(labels, guidance_scale=4.0, num_inference_steps=3, latents=lat0.clone())
    # oracle loop
    sch = S.DDIMRef(**kw)
    sch.set_timesteps(3)
    x = torch.cat([lat0, lat0])
    lab = torch.cat([labels, torch.full((2,), cfg["num_embeds_ada_norm"])])
    for t in sch.timesteps:
        half = x[:2]
        x = torch.cat([half, half])
        n = R.dit_forward(Pb, cfg, x, float(t), lab)
        eps, rest = n[:, :4], n[:, 4:]
        cond, uncond = eps.chunk(2)
        he = uncond + 4.0 * (cond - uncond)
        mo = torch.cat([torch.cat([he, he]), rest], dim=1)[:, :4]
        x = torch.from_numpy(sch.step(mo.numpy(), int(t), x.numpy(), 0.0))
    ref = x[:2]
    assert out.shape == (2, 4, 16, 16) and _rel(out, ref) < 3e-2, _rel(out, ref)
    # guidance_scale <= 1: no batch doubling, no null class
    out1 = DiTDenoiser(model, DDIMScheduler(**kw))(labels, guidance_scale=1.0, num_inference_steps=2, latents=lat0.clone())
    assert out1.shape == (2, 4, 16, 16) and torch.isfinite(out1).all()
