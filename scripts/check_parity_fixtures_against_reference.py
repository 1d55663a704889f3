"""The full-depth parity fixtures (tests/golden/parity/*.npz: oracle trajectories of BASELINE configs 2, 3 and 5 at the REAL
architectures -- SD-1.5, SDXL, SD3-medium -- that tests/test_gpu_parity_loops.py replays on the device) against the reference's own
model code: for every stored (x_in, prediction) pair the reference's unmodified UNet2DConditionModel / SD3Transformer2DModel,
executed over oracle/paddle_shim.py on the case's full parameter set, must reproduce the stored prediction.

    python scripts/check_parity_fixtures_against_reference.py [case ...] [--max-steps K]      (build container, CPU, tens of minutes)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import reference_runner as rr  # noqa: E402
from tests import parity_cases as PC  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cases", nargs="*", default=list(PC.CASES))
ap.add_argument("--max-steps", type=int, default=3, help="stored steps checked per case (first, middle, last ...)")
args = ap.parse_args()
if not rr.available():
    sys.exit("/root/reference is not present")
torch.set_grad_enabled(False)
for name in args.cases:
    case = PC.CASES[name]
    fx = np.load(os.path.join(PC.GOLDEN_DIR, name + ".npz"))
    t0 = time.time()
    P = PC.case_params(case)
    x, enc, extra = PC.case_inputs(case)
    if case["kind"] == "sd3":
        net = rr.ref_module("transformer_sd3").SD3Transformer2DModel(**case["cfg"])
        net.eval()
        rr.load_params(net, P, computed=("pos_embed.pos_embed", "norm_out.norm.bias", "norm1_context.norm.bias"))
        call = lambda xin, t: net(rr.to_shim(xin), encoder_hidden_states=rr.to_shim(enc), pooled_projections=rr.to_shim(extra),  # noqa: E731
                                  timestep=rr.to_shim(torch.tensor([float(t)]))).sample
    else:
        net = rr.build_unet(case["cfg"], P)
        call = lambda xin, t: net(rr.to_shim(xin), rr.to_shim(torch.tensor([float(int(t))])), rr.to_shim(enc),  # noqa: E731
                                  added_cond_kwargs=rr.to_shim(extra)).sample
    print(f"{name}: reference model built ({sum(v.numel() for v in P.values()) / 1e6:.1f} M parameters, {time.time() - t0:.0f} s)", flush=True)
    kept = list(fx["kept"])
    pick = sorted(set(np.linspace(0, len(kept) - 1, min(args.max_steps, len(kept))).round().astype(int).tolist()))
    for k in pick:
        step = int(kept[k])
        t = fx["sched"][step][0]
        t0 = time.time()
        ref = rr.from_shim(call(torch.from_numpy(fx["x_in"][k]), t)).float()
        want = torch.from_numpy(fx["pred"][k]).float()
        rel = float((ref - want).abs().max() / want.abs().max())
        print(f"   step {step:2d} (t = {t:7.2f}): max |reference - stored oracle prediction| / max |prediction| = {rel:.3g}   ({time.time() - t0:.0f} s)", flush=True)
        assert rel < 5e-5, (name, step, rel)
    del net, P
print("all stored predictions are the reference's")
