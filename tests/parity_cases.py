"""Full-depth denoising-loop parity cases (BASELINE.json configs 2, 3, 5) shared by the fixture generator
(scripts/make_parity_golden.py, run HERE on the CPU: the oracle loops take minutes) and the GPU tests / reports that replay
the same loops on the device and compare with the committed oracle trajectories under tests/golden/parity/.

Weights: the seeded synthetic parameter sets of SURVEY.md 8(d), made representable in BOTH 16-bit element types (bf16 rounding,
then magnitudes below fp16's smallest normal 2^-14 set to zero): the bf16 build, the fp16 build and the fp32 oracle multiply
by identical numbers, so one oracle trajectory serves every device mode.  Latent state and scheduler arithmetic: float64 on
the host for oracle and device alike, so what is compared is the accumulated error of the noise predictions only.

The oracle is the torch-CPU restatement of ppdiffusers, held to the reference's own module code by tests/test_reference_modules.py;
the stored per-step predictions of these fixtures are reproduced bit for bit by the reference's own model classes at these real
architectures (scripts/check_parity_fixtures_against_reference.py, profiles/r03_parity_fixtures_vs_reference.txt)
(Paddle itself cannot be installed here: see oracle/__init__.py for what that leaves unpinned).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from tests.configs import SD15, SD3_MEDIUM, SDXL

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity")

# name -> model family, config, latent geometry, text length, scheduler, steps, which steps' (x_in, eps) pairs are stored for the
# teacher-forced per-step check ("all" or a list), and the reference pipeline settings the loop follows
CASES = {
    # BASELINE config 3 at full depth, small latents: every step stored (north_star's 1e-3 is stated on the end latents)
    "sdxl_1x4x32x32_euler30": dict(kind="unet", cfg=SDXL, B=1, C=4, H=32, W=32, L=77, sched="euler", steps=30, keep="all"),
    # one prompt of the headline geometry, 10 Euler steps
    "sdxl_1x4x128x128_euler10": dict(kind="unet", cfg=SDXL, B=1, C=4, H=128, W=128, L=77, sched="euler", steps=10, keep=[0, 9]),
    # the same prompt over the metric's whole 30-step Euler schedule (round 4: north_star's tolerance is stated on the end latents
    # of the headline geometry; 16 minutes of CPU for the oracle loop)
    "sdxl_1x4x128x128_euler30": dict(kind="unet", cfg=SDXL, B=1, C=4, H=128, W=128, L=77, sched="euler", steps=30, keep=[0, 15, 29]),
    # BASELINE config 2: SD-1.5, 50 DDIM steps
    "sd15_1x4x64x64_ddim50": dict(kind="unet", cfg=SD15, B=1, C=4, H=64, W=64, L=77, sched="ddim", steps=50, keep=[0, 25, 49]),
    # BASELINE config 5: SD3-medium MMDiT, 28 flow-matching Euler steps (512^2 image = 64x64 latents: the CPU oracle at 128x128
    # needs ~1.5 min per step here)
    "sd3_1x16x64x64_flow28": dict(kind="sd3", cfg=SD3_MEDIUM, B=1, C=16, H=64, W=64, L=154, sched="flow", steps=28, keep=[0, 14, 27]),
    # the same loop as the fp8 modes see it (round 4): the oracle multiplies by the SAME quantised operands -- block matrices through
    # the e4m3 + per-channel-scale round trip (fp8w), plus per-token e4m3 activations into the block GEMMs (w8a8,
    # oracle/sd3_ref.py act_quant) -- so that what is compared is the kernels' arithmetic, not the quantisation the mode chose
    "sd3_1x16x64x64_flow28_fp8w": dict(kind="sd3", cfg=SD3_MEDIUM, B=1, C=16, H=64, W=64, L=154, sched="flow", steps=28, keep=[0, 14, 27],
                                       quant="fp8w"),
    "sd3_1x16x64x64_flow28_w8a8": dict(kind="sd3", cfg=SD3_MEDIUM, B=1, C=16, H=64, W=64, L=154, sched="flow", steps=28, keep=[0, 14, 27],
                                       quant="w8a8"),
}


# Single forwards at the LAUNCH SET the metric times (round 5): every loop case above runs one prompt, so its GEMMs see M = 1024 / 4096
# rows and take other tiles (and split-K) than the bs-8 step bench.py measures (M = 8192 / 32768: 256x160 persistent tiles, 256x320
# streaming FF1). One oracle forward of the whole batch -- step `step` of the case's schedule on the seeded inputs -- closes that:
# the device's prediction is compared per prompt (tests/test_gpu_parity_loops.py, bench.py's parity children: pred_rel_bs8).
FWD_CASES = {
    "sdxl_8x4x128x128_fwd": dict(kind="unet", cfg=SDXL, B=8, C=4, H=128, W=128, L=77, sched="euler", steps=30, step=0),
    # BASELINE config 5 at the geometry it is quoted on (round 6): SD3-medium 1024^2 bs 8 -- 33 k-row GEMM tiles, the 4,250-key joint
    # attention, the widened-fp8 and W8A8 launches at M >= 4096. ``prompts``: which rows of the batch the oracle computed (prompts
    # never interact, so the twins on quantised operands store two rows of the eight; the device always runs the whole batch)
    "sd3_8x16x128x128_fwd": dict(kind="sd3", cfg=SD3_MEDIUM, B=8, C=16, H=128, W=128, L=154, sched="flow", steps=28, step=0),
    "sd3_8x16x128x128_fwd_fp8w": dict(kind="sd3", cfg=SD3_MEDIUM, B=8, C=16, H=128, W=128, L=154, sched="flow", steps=28, step=0,
                                      quant="fp8w", prompts=[0, 7]),
    "sd3_8x16x128x128_fwd_w8a8": dict(kind="sd3", cfg=SD3_MEDIUM, B=8, C=16, H=128, W=128, L=154, sched="flow", steps=28, step=0,
                                      quant="w8a8", prompts=[0, 7]),
}


def fp8_roundtrip(P):
    """block matrices -> e4m3 with one fp32 scale per output channel (absmax / 448) and back: the numbers a weight_dtype="fp8" model
    multiplies by (paddlemix_amd/sd3.py quantize_fp8_rows is the device-side definition; restated here for the checker)"""
    out = {}
    for k, v in P.items():
        if k.startswith("transformer_blocks.") and k.endswith(".weight") and ".norm1" not in k:
            w = v.t().contiguous()                                    # [N, K]
            scale = w.abs().amax(dim=1).clamp_min(1e-12) / 448.0
            q = (w / scale[:, None]).to(torch.float8_e4m3fn).to(torch.float32)
            out[k] = (q * scale[:, None]).t().contiguous()
        else:
            out[k] = v
    return out


def dual16(P):
    """matrices / conv kernels -> values exact in bf16 AND fp16; 1-D parameters stay fp32 (the device keeps them in fp32)"""
    out = {}
    for k, v in P.items():
        if v.dim() > 1:
            w = v.to(torch.bfloat16).float()
            w = torch.where(w.abs() < 2.0 ** -14, torch.zeros_like(w), w)
            assert torch.equal(w.to(torch.float16).float(), w)
            out[k] = w
        else:
            out[k] = v.float()
    return out


DRAW_STATES = os.path.join(GOLDEN_DIR, "param_draw_states.npz")   # scripts/make_param_draw_states.py
DRAW_SHARDS = 32


def family(case) -> str:
    return "sd3" if case["kind"] == "sd3" else ("sdxl" if case["cfg"].get("addition_embed_type") else "sd15")


def _synth(case):
    if case["kind"] == "sd3":
        from paddlemix_amd.sd3 import sd3_param_shapes, synth_sd3_params
        return synth_sd3_params, sd3_param_shapes(case["cfg"])
    from paddlemix_amd.unet import synth_unet_params, unet_param_shapes
    return synth_unet_params, unet_param_shapes(case["cfg"])


def shard_bounds(shapes, n=DRAW_SHARDS):
    """construction order cut into n runs of about equal element count: n + 1 indices"""
    numel = np.cumsum([int(np.prod(s)) for s in shapes.values()])
    cuts = [int(np.searchsorted(numel, numel[-1] * i / n)) for i in range(1, n)]
    return [0] + [min(c + 1, len(numel)) for c in cuts] + [len(numel)]


def shapes_sig(shapes) -> str:
    import hashlib
    return hashlib.sha1(repr([(k, tuple(v)) for k, v in shapes.items()]).encode()).hexdigest()


def case_params(case, threads=None):
    """the seeded synthetic parameter set of SURVEY.md 8(d) (the product's and the oracle's generators draw the same numbers:
    tests/test_parity_cases.py), made exact in both 16-bit types.

    One generator, construction order: 2.6 B normals take a minute or more of ONE core (and 160 s when two children draw side by
    side: the GPU suite's longest item through round 5). The committed generator states at DRAW_SHARDS cut points of that very
    sequence (param_draw_states.npz, a few hundred KB) let the shards be drawn by parallel threads -- each checks that it ENDS on the
    next shard's stored state, so the result is the serial draw bit for bit or an error, never a different parameter set."""
    synth, shapes = _synth(case)
    fam = family(case)
    states = None
    if threads != 0 and os.path.exists(DRAW_STATES):
        with np.load(DRAW_STATES) as z:
            if str(z[fam + "_sig"]) == shapes_sig(shapes):   # (the small test configurations have no stored states: serial)
                states, bounds = torch.from_numpy(z[fam + "_states"]), [int(b) for b in z[fam + "_bounds"]]
    if states is None:
        return dual16(synth(case["cfg"], seed=1234))
    assert bounds == shard_bounds(shapes, len(bounds) - 1)

    def shard(i):
        g = torch.Generator()
        g.set_state(states[i].clone())   # (a row VIEW crashes set_state in torch 2.10)
        P = synth(case["cfg"], generator=g, only=range(bounds[i], bounds[i + 1]))
        if not torch.equal(g.get_state(), states[i + 1]):
            raise RuntimeError(f"{fam}: shard {i} did not end on the stored generator state (another torch RNG?): draw serially (threads=0)")
        return dual16(P)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads or min(len(bounds) - 1, os.cpu_count() or 1)) as ex:
        parts = list(ex.map(shard, range(len(bounds) - 1)))
    out = {}
    for p in parts:
        out.update(p)
    assert list(out) == list(shapes)
    return out


def case_inputs(case):
    """(initial noise [B,C,H,W], text states [B,L,D], extra conditioning) -- fp32, seeded"""
    g = torch.Generator().manual_seed(0)
    cfg, B = case["cfg"], case["B"]
    x = torch.randn(B, case["C"], case["H"], case["W"], generator=g)
    if case["kind"] == "sd3":
        enc = torch.randn(B, case["L"], cfg["joint_attention_dim"], generator=g)
        return x, enc, torch.randn(B, cfg["pooled_projection_dim"], generator=g)
    enc = torch.randn(B, case["L"], cfg["cross_attention_dim"], generator=g)
    added = None
    if cfg.get("addition_embed_type") == "text_time":
        td = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
        added = dict(text_embeds=torch.randn(B, td, generator=g),
                     time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(B, 1))
    return x, enc, added


def schedule(case):
    """-> (x0_scale, [(model timestep, input scale, a, b)]): step i feeds the model x * input_scale and updates
    x <- a * x + b * prediction (the three schedulers are linear in (x, prediction) for epsilon models with eta = 0)."""
    from oracle import schedulers_ref as S
    n = case["steps"]
    if case["sched"] == "euler":
        # the reference's SDXL test scheduler (ppdiffusers/tests/pipelines/stable_diffusion_xl/test_stable_diffusion_xl.py:84-90)
        sch = S.EulerRef(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading", steps_offset=1)
        sch.set_timesteps(n)
        sig = sch.sigmas.astype(np.float64)
        rows = [(float(sch.timesteps[i]), float(1.0 / (sig[i] ** 2 + 1.0) ** 0.5), 1.0, float(sig[i + 1] - sig[i])) for i in range(n)]
        return float(sch.init_noise_sigma), rows
    if case["sched"] == "ddim":
        # the SD test scheduler (ppdiffusers/tests/pipelines/stable_diffusion/test_stable_diffusion.py:122-128)
        sch = S.DDIMRef(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                        set_alpha_to_one=False, steps_offset=1)
        sch.set_timesteps(n)
        rows = []
        for t in sch.timesteps:
            t = int(t)
            prev = t - sch.T // n
            a_t = float(sch.alphas_cumprod[t])
            a_p = float(sch.alphas_cumprod[prev]) if prev >= 0 else float(sch.final_alpha_cumprod)
            # x0 = (x - sqrt(1-a_t) e) / sqrt(a_t);  x' = sqrt(a_p) x0 + sqrt(1-a_p) e
            ca = (a_p / a_t) ** 0.5
            rows.append((float(t), 1.0, ca, (1 - a_p) ** 0.5 - ca * (1 - a_t) ** 0.5))
        return 1.0, rows
    if case["sched"] == "flow":
        sch = S.FlowMatchEulerRef(shift=3.0)
        sch.set_timesteps(n)
        sig = sch.sigmas.astype(np.float64)
        return 1.0, [(float(sch.timesteps[i]), 1.0, 1.0, float(sig[i + 1] - sig[i])) for i in range(n)]
    raise ValueError(case["sched"])


def kept_steps(case):
    return list(range(case["steps"])) if case["keep"] == "all" else list(case["keep"])


def run_loop(case, predict, x_init, on_step=None, sched=None):
    """free-running loop: predict(x_in fp32, timestep, step index) -> prediction tensor; float64 state. Returns end latents.
    ``sched``: (x0_scale, rows) as stored in a fixture (the device replay takes the schedule from there, not from oracle/)."""
    s0, rows = sched if sched is not None else schedule(case)
    x = x_init.double() * s0
    for i, (t, cin, a, b) in enumerate(rows):
        x_in = (x * cin).float()
        pred = predict(x_in, t, i).double().cpu()
        if on_step is not None:
            on_step(i, x_in, pred)
        x = a * x + b * pred
    return x


def oracle_predictor(case, P, enc, extra):
    if case["kind"] == "sd3":
        from oracle.sd3_ref import sd3_forward
        if case.get("quant"):
            P = fp8_roundtrip(P)
        aq = case.get("quant") == "w8a8"
        return lambda x_in, t, i: sd3_forward(P, case["cfg"], x_in, enc, extra, float(t), act_quant=aq)
    from oracle.unet_ref import unet_forward
    return lambda x_in, t, i: unet_forward(P, case["cfg"], x_in, int(t), enc, added_cond_kwargs=extra)


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name + ".npz")


def load_golden(name):
    z = np.load(golden_path(name))
    return {k: z[k] for k in z.files}


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / b.norm()).item()


def device_report(case_name, make_model=None, dev="cuda", model=None):
    """Replay one case on the device: free-running end latents and teacher-forced per-step predictions vs the committed oracle
    trajectory. ``make_model(case, P)`` -> object with the product's forward signature (UNet2DConditionModel /
    SD3Transformer2DModel), or ``model``: one already built from ``case_params(case)``."""
    case = CASES[case_name]
    gold = load_golden(case_name)
    x0, enc, extra = case_inputs(case)
    if model is None:
        P = case_params(case)
        model = make_model(case, P)
        del P
    enc_d = enc.to(dev)
    if case["kind"] == "sd3":
        extra_d = extra.to(dev)

        def predict(x_in, t, i):
            return model(x_in.to(dev), enc_d, extra_d, torch.tensor([float(t)], device=dev), return_dict=False)[0].float()
    else:
        extra_d = None if extra is None else {k: v.to(dev) for k, v in extra.items()}

        def predict(x_in, t, i):
            return model(x_in.to(dev), int(t), enc_d, added_cond_kwargs=extra_d, return_dict=False)[0].float()
    sched = (float(gold["x0_scale"]), [tuple(float(v) for v in r) for r in gold["sched"]])
    x_end = run_loop(case, predict, x0, sched=sched)
    out = {"end_latents_rel": rel_l2(x_end, gold["x_end"]), "steps": case["steps"]}
    errs = []
    for j, i in enumerate(gold["kept"].tolist()):
        t = sched[1][i][0]
        e = predict(torch.from_numpy(gold["x_in"][j]), t, i).cpu()
        errs.append(rel_l2(e, gold["pred"][j]))
    out["pred_rel_teacher_forced_max"] = max(errs)
    out["pred_rel_teacher_forced_first_last"] = [errs[0], errs[-1]]
    return out


def fwd_inputs(case):
    """(model input fp32 [B,C,H,W], model timestep, text states, extra conditioning) of a FWD_CASES entry: the seeded noise of
    case_inputs scaled the way step ``step`` of the case's schedule feeds the model"""
    x0, enc, extra = case_inputs(case)
    s0, rows = schedule(case)
    t, cin, _, _ = rows[case["step"]]
    return (x0.double() * s0 * cin).float(), t, enc, extra


def fwd_prompts(case):
    """rows of the batch the oracle forward of a FWD_CASES entry holds"""
    return list(case.get("prompts") or range(case["B"]))


def oracle_fwd(case):
    """the oracle's forward of a FWD_CASES entry (rows fwd_prompts(case) of the batch) -> fp32 [len(prompts), C, H, W]"""
    P = case_params(case)
    x_in, t, enc, extra = fwd_inputs(case)
    rows = fwd_prompts(case)
    if case["kind"] == "sd3":
        from oracle.sd3_ref import sd3_forward
        if case.get("quant"):
            P = fp8_roundtrip(P)
        aq = case.get("quant") == "w8a8"
        # one prompt at a time: the 4,250-key joint attention of the restatement materialises [B, 24, S, S] scores
        return torch.cat([sd3_forward(P, case["cfg"], x_in[b:b + 1], enc[b:b + 1], extra[b:b + 1], float(t), act_quant=aq).float() for b in rows])
    from oracle.unet_ref import unet_forward
    assert rows == list(range(case["B"]))
    return unet_forward(P, case["cfg"], x_in, int(t), enc, added_cond_kwargs=extra).float()


def device_fwd_report(case_name, model, dev="cuda"):
    """One whole-batch forward on the device vs the committed oracle forward: rel-L2 of the batch and of every prompt."""
    case = FWD_CASES[case_name]
    gold = load_golden(case_name)
    x_in, t, enc, extra = fwd_inputs(case)
    assert abs(float(x_in.double().sum()) - float(gold["x_in_sum"])) <= 1e-6 * max(1.0, abs(float(gold["x_in_sum"]))), "seeded inputs differ from the fixture's"
    if case["kind"] == "sd3":
        pred = model(x_in.to(dev), enc.to(dev), extra.to(dev), torch.tensor([float(t)], device=dev), return_dict=False)[0].float().cpu()
    else:
        extra_d = None if extra is None else {k: v.to(dev) for k, v in extra.items()}
        pred = model(x_in.to(dev), int(t), enc.to(dev), added_cond_kwargs=extra_d, return_dict=False)[0].float().cpu()
    rows = fwd_prompts(case)
    per = [rel_l2(pred[b], gold["pred"][j]) for j, b in enumerate(rows)]
    return {"pred_rel_bs8": rel_l2(pred[rows], gold["pred"]), "pred_rel_bs8_per_prompt_max": max(per), "pred": pred}
