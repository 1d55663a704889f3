"""CPU: the machinery of the full-depth loop parity (tests/parity_cases.py) on a tiny case -- oracle trajectory -> fixture ->
device replay through the C-ABI interpreter -- and the committed fixtures themselves (present, finite, the shapes the cases name)."""
import numpy as np
import pytest
import torch

from tests import parity_cases as PC
from tests.abi_emulator import Emulator, on_emulator
from tests.configs import TINY


def test_dual16_weights_are_exact_in_both_element_types():
    g = torch.Generator().manual_seed(0)
    P = {"a.weight": torch.randn(64, 64, generator=g) * 1e-3, "a.bias": torch.randn(64, generator=g)}
    D = PC.dual16(P)
    w = D["a.weight"]
    assert torch.equal(w.to(torch.bfloat16).float(), w) and torch.equal(w.to(torch.float16).float(), w)
    assert torch.equal(D["a.bias"], P["a.bias"])                  # 1-D parameters stay fp32 on the device too
    assert (w == 0).sum() > 0 and (w != 0).sum() > 0.9 * w.numel()  # only magnitudes below fp16's normal range were dropped


@pytest.mark.parametrize("sched,steps", [("euler", 4), ("ddim", 5)])
def test_loop_replay_through_the_interpreter(tmp_path, monkeypatch, sched, steps):
    from paddlemix_amd.unet import UNet2DConditionModel
    case = dict(kind="unet", cfg=TINY, B=1, C=4, H=16, W=16, L=7, sched=sched, steps=steps, keep=[0, steps - 1])
    monkeypatch.setitem(PC.CASES, "tiny_test", case)
    monkeypatch.setattr(PC, "GOLDEN_DIR", str(tmp_path))
    P = PC.case_params(case)
    x0, enc, extra = PC.case_inputs(case)
    xin, pred = {}, {}

    def on_step(i, x_in, p):
        if i in case["keep"]:
            xin[i], pred[i] = x_in.numpy().copy(), p.float().numpy().copy()

    x_end = PC.run_loop(case, PC.oracle_predictor(case, P, enc, extra), x0, on_step)
    s0_, rows_ = PC.schedule(case)
    np.savez(PC.golden_path("tiny_test"), x_end=x_end.numpy(), kept=np.array(case["keep"]), x_in=np.stack([xin[i] for i in case["keep"]]),
             pred=np.stack([pred[i] for i in case["keep"]]), x0_scale=np.float64(s0_), sched=np.array(rows_, dtype=np.float64))
    r = PC.device_report("tiny_test", lambda c, Pm: on_emulator(UNet2DConditionModel, c["cfg"], Pm), dev="cpu")
    # the interpreter has the device's 16-bit rounding points: ~1e-2 per prediction, less on the latents
    assert 1e-4 < r["pred_rel_teacher_forced_max"] < 2e-2 and r["end_latents_rel"] < 1e-2 and r["steps"] == steps
    # the loop is the reference scheduler's: one step of run_loop == EulerRef / DDIMRef .step on the same prediction
    from oracle import schedulers_ref as S
    s0, rows = PC.schedule(case)
    if sched == "euler":
        sch = S.EulerRef(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading", steps_offset=1)
    else:
        sch = S.DDIMRef(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
                        steps_offset=1)
    sch.set_timesteps(steps)
    x = (x0 * s0).numpy().astype(np.float32)
    e = torch.randn(x0.shape, generator=torch.Generator().manual_seed(1)).numpy()
    t, cin, a, b = rows[0]
    assert np.allclose(sch.scale_model_input(x, t), x * cin, rtol=1e-6)
    assert np.allclose(sch.step(e, t, x), a * x + b * e, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", list(PC.CASES))
def test_committed_trajectories(name):
    case = PC.CASES[name]
    z = PC.load_golden(name)
    shape = (case["B"], case["C"], case["H"], case["W"])
    assert z["x_end"].shape == shape and z["x_end"].dtype == np.float64 and np.isfinite(z["x_end"]).all()
    kept = PC.kept_steps(case)
    assert z["kept"].tolist() == kept and z["x_in"].shape == (len(kept),) + shape and z["pred"].shape == (len(kept),) + shape
    assert np.isfinite(z["x_in"]).all() and np.isfinite(z["pred"]).all() and float(np.abs(z["pred"]).mean()) > 1e-3
    s0, rows = PC.schedule(case)
    assert len(rows) == case["steps"] and all(np.isfinite(r).all() for r in rows)
    # the stored schedule (what the device replay reads) is the oracle scheduler's
    assert float(z["x0_scale"]) == s0 and np.array_equal(z["sched"], np.array(rows, dtype=np.float64))


def test_product_and_oracle_draw_the_same_synthetic_weights():
    """the fixtures were made from the oracle's generator, the device replay uses the product's: same seeded numbers"""
    from oracle.sd3_ref import synth_sd3_params as o_sd3
    from oracle.unet_ref import synth_unet_params as o_unet
    from paddlemix_amd.sd3 import synth_sd3_params as p_sd3
    from paddlemix_amd.unet import synth_unet_params as p_unet
    from tests.configs import MINI_SD3, MINI_XL
    for a, b in ((o_unet(MINI_XL, seed=1234), p_unet(MINI_XL, seed=1234)), (o_sd3(MINI_SD3, seed=1234), p_sd3(MINI_SD3, seed=1234))):
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_sharded_parameter_draw_is_the_serial_draw(tmp_path, monkeypatch):
    """case_params draws the full-size sets as parallel shards from stored generator states (round 6): on a small configuration, with
    states recorded the way scripts/make_param_draw_states.py records them, the shards reproduce the serial draw bit for bit; a state
    that does not belong to the sequence is refused; the committed file matches today's parameter lists and starts at seed 1234"""
    import importlib.util
    import os
    from tests.configs import MINI_SD3, MINI_XL
    spec = importlib.util.spec_from_file_location("make_param_draw_states", os.path.join(os.path.dirname(PC.GOLDEN_DIR), "..", "..", "scripts",
                                                                                          "make_param_draw_states.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    out = {}
    cases = (dict(kind="unet", cfg=MINI_XL), dict(kind="sd3", cfg=MINI_SD3))
    for case in cases:
        fam = PC.family(case)
        out[fam + "_states"], out[fam + "_bounds"] = mk.record(case, 8)
        out[fam + "_sig"] = np.asarray(PC.shapes_sig(PC._synth(case)[1]))
    path = str(tmp_path / "states.npz")
    np.savez(path, **out)
    with np.load(PC.DRAW_STATES) as z:   # the committed file, before the path is patched away
        seed_state = torch.Generator().manual_seed(1234).get_state().numpy()
        for name in ("sdxl_1x4x32x32_euler30", "sd15_1x4x64x64_ddim50", "sd3_1x16x64x64_flow28"):
            c = PC.CASES[name]
            fam, shapes = PC.family(c), PC._synth(c)[1]
            assert str(z[fam + "_sig"]) == PC.shapes_sig(shapes) and np.array_equal(z[fam + "_states"][0], seed_state)
            assert [int(b) for b in z[fam + "_bounds"]] == PC.shard_bounds(shapes, PC.DRAW_SHARDS)
    monkeypatch.setattr(PC, "DRAW_STATES", path)
    for case in cases:
        a, b = PC.case_params(case, threads=3), PC.case_params(case, threads=0)
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    out["sdxl_states"][3] = out["sdxl_states"][2]
    np.savez(path, **out)
    with pytest.raises(RuntimeError, match="did not end on the stored generator state"):
        PC.case_params(cases[0], threads=2)
