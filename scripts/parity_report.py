"""Parity report of the HIP UNet against the torch-CPU oracle at the geometries BASELINE.json names -- run on the GPU box.

    python scripts/parity_report.py [--quick] [--out gpurun_out/parity.json]

One child process per element type (the library is built twice: bf16 and IEEE-half elements, one type per process);
each child measures, for residual_dtype in {16-bit, fp32}:
  * single forward, rel-L2 of the noise prediction against the oracle on identical (16-bit representable) weights:
    tiny, SDXL-structured mini, the full SD-1.5 parameter set at 1x4x64x64 and the full SDXL parameter set at 1x4x128x128
    (BASELINE.json configs 1-3; batch 1 of the bs-8 headline: prompts do not interact, tests/test_gpu_unet.py proves it
    bit-exactly);
  * 30 Euler steps (timestep_spacing="leading", steps_offset=1, scaled_linear betas: the reference's SDXL test scheduler,
    ppdiffusers/tests/pipelines/stable_diffusion_xl/test_stable_diffusion_xl.py:84-90) on the mini config: per-step rel-L2
    of epsilon with the device fed the ORACLE's latents (teacher forced) and rel-L2 of the final latents of the
    free-running device loop, against a float64 oracle loop;
  * the oracle's own floor: float32 vs float64 oracle on the small configs.
The oracle is a torch-CPU restatement of ppdiffusers (Paddle cannot be installed here): parity is "device vs restatement",
unpinned against Paddle itself (oracle/__init__.py).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(elem: str, quick: bool) -> dict:
    import numpy as np
    import torch

    from oracle import schedulers_ref as S
    from oracle import unet_ref as U
    from paddlemix_amd import _lib
    from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params
    from tests.configs import MINI_XL, SD15, SDXL, TINY
    from tests.test_host_logic import _inputs

    ed = _lib.elem_dtype()
    assert (elem == "fp16") == (ed == torch.float16)
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()  # noqa: E731
    cuda = lambda x: None if x is None else ({k: v.cuda() for k, v in x.items()} if isinstance(x, dict) else x.cuda())  # noqa: E731
    out = {"elem": elem, "threads": torch.get_num_threads(), "forward": {}, "loop": {}}

    def params(cfg):
        P = synth_unet_params(cfg, seed=1234)
        return {k: (v.to(ed).float() if v.dim() > 1 else v) for k, v in P.items()}   # what the device holds

    # (the CPU oracle needs ~40 s per SDXL prompt at 128x128 on the GPU box's 128 threads: keep the list short)
    cases = [("tiny", TINY, 2, 16, 16, 7, True), ("mini_xl", MINI_XL, 2, 32, 32, 77, True)]
    if not quick:
        cases += [("sd15_1x4x64x64", SD15, 1, 64, 64, 77, False), ("sdxl_1x4x128x128", SDXL, 1, 128, 128, 77, False)]
    if os.environ.get("PARITY_SKIP_SD15"):
        cases = [c for c in cases if not c[0].startswith("sd15")]
    if os.environ.get("PARITY_ONLY_SDXL_LOOP"):
        cases = []
    for name, cfg, B, H, W, L, small in cases:
        P = params(cfg)
        sample, enc, added = _inputs(cfg, B, H, W, L)
        t0 = time.time()
        ref = U.unet_forward(P, cfg, sample, 501, enc, added_cond_kwargs=added)
        t_ref = time.time() - t0
        r = {"oracle_seconds": round(t_ref, 2)}
        if small:   # the oracle's own arithmetic floor
            P64 = {k: v.double() for k, v in P.items()}
            ref64 = U.unet_forward(P64, cfg, sample.double(), 501, enc.double(),
                                   added_cond_kwargs=None if added is None else {k: v.double() for k, v in added.items()})
            r["oracle_f32_vs_f64"] = rel(ref, ref64)
            ref = ref64
        for rd in ("16", "fp32"):
            model = UNet2DConditionModel(cfg, P, residual_dtype=rd)
            got = model(cuda(sample), 501, cuda(enc), added_cond_kwargs=cuda(added), return_dict=False)[0]
            torch.cuda.synchronize()
            r["resid_" + rd] = rel(got.cpu(), ref)
            assert torch.isfinite(got).all()
            del model
            torch.cuda.empty_cache()
        out["forward"][name] = r
        print(elem, name, r, flush=True)

    # ---- 30 Euler steps (teacher-forced per-step epsilon error + free-running end latents), float64 oracle loop ----
    # (mini: float64 oracle loop; the full SDXL parameter set at 32x32 latents: the fp32 oracle -- a float64 forward of 2.6 B
    # parameters takes minutes per step on the CPU, and the fp32 oracle sits 1e-6 from the float64 one, see oracle_f32_vs_f64)
    loops = [("mini_xl_2x4x32x32", MINI_XL, 2, 32, 32, 77, torch.float64)]
    if os.environ.get("PARITY_SDXL_LOOP") or os.environ.get("PARITY_ONLY_SDXL_LOOP"):
        loops.append(("sdxl_arch_1x4x32x32", SDXL, 1, 32, 32, 77, torch.float32))
    if os.environ.get("PARITY_ONLY_SDXL_LOOP"):
        loops = loops[1:]
    for name, cfg, B, H, W, L, odt in loops:
        P = params(cfg)
        P64 = {k: v.to(odt) for k, v in P.items()}
        sch = S.EulerRef(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading",
                         steps_offset=1)
        sch.set_timesteps(30)
        sig = sch.sigmas.astype(np.float64)
        ts = sch.timesteps
        sample, enc, added = _inputs(cfg, B, H, W, L)
        added64 = None if added is None else {k: v.to(odt) for k, v in added.items()}
        r = {}
        for rd in ("16", "fp32"):
            model = UNet2DConditionModel(cfg, P, residual_dtype=rd)
            x_ref = sample.double() * float(sch.init_noise_sigma)
            x_dev = x_ref.clone()
            eps_err = []
            for i, t in enumerate(ts):
                s = sig[i]
                xin = x_ref / (s * s + 1.0) ** 0.5
                eps_ref = U.unet_forward(P64, cfg, xin.to(odt), int(t), enc.to(odt), added_cond_kwargs=added64).double()
                # teacher forced: the device sees the oracle's latents of this step
                e_tf = model(cuda(xin.float()), int(t), cuda(enc), added_cond_kwargs=cuda(added), return_dict=False)[0]
                eps_err.append(rel(e_tf.cpu(), eps_ref))
                # free running: the device's own latents
                xin_d = x_dev / (s * s + 1.0) ** 0.5
                e_fr = model(cuda(xin_d.float()), int(t), cuda(enc), added_cond_kwargs=cuda(added), return_dict=False)[0]
                x_ref = x_ref + eps_ref * (sig[i + 1] - s)
                x_dev = x_dev + e_fr.cpu().double() * (sig[i + 1] - s)
            r["oracle_dtype"] = str(odt)
            r["resid_" + rd] = {"eps_rel_per_step_max": max(eps_err), "eps_rel_per_step_mean": sum(eps_err) / len(eps_err),
                                "eps_rel_first_last": [eps_err[0], eps_err[-1]], "end_latents_rel": rel(x_dev, x_ref)}
            del model
            torch.cuda.empty_cache()
        out["loop"][name] = r
        print(elem, name, r, flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity.json"))
    a = ap.parse_args()
    if a.child:
        print("PARITY_JSON " + json.dumps(child(a.child, a.quick)))
        return
    res = {}
    for elem in ("bf16", "fp16"):
        env = dict(os.environ, MI355X_SD_DTYPE=elem)
        env.pop("MI355X_SD_RESID", None)
        # stream the child's lines through (a run cut short by a time limit still leaves its finished cases in the log)
        p = subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "--child", elem] + (["--quick"] if a.quick else []),
                             env=env, stdout=subprocess.PIPE, text=True)
        line = None
        for ln in p.stdout:
            sys.stdout.write(ln)
            sys.stdout.flush()
            if ln.startswith("PARITY_JSON "):
                line = ln
        if p.wait() != 0 or line is None:
            raise SystemExit(f"child {elem} failed")
        res[elem] = json.loads(line[len("PARITY_JSON "):])
        with open(a.out + ".partial", "w") as f:
            json.dump(res, f, indent=1)
    res["note"] = ("rel-L2 vs the torch-CPU oracle (restatement of ppdiffusers; Paddle unavailable -> unpinned) on identical "
                   "16-bit-representable synthetic weights; north_star target 1e-3 on latents")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
