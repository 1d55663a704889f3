"""Oracle trajectories of the full-depth denoising loops (tests/parity_cases.py) -> tests/golden/parity/<case>.npz.

Run on the CPU (no GPU, no network):   python scripts/make_parity_golden.py [case ...]
Each file holds the float64 end latents of the free-running fp32 oracle loop (x_end), and for the kept steps the model input
(x_in, fp32), the oracle's prediction (pred, fp32) and their indices (kept) for the teacher-forced per-step check, the schedule
the loop followed (x0_scale, sched rows: the device replay reads it from here, not from oracle/), plus the wall time and thread
count of the run.  The GPU tests regenerate the same seeded weights / inputs and replay the loop on the
device (tests/test_gpu_parity_loops.py, scripts/parity_loops.py).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import parity_cases as PC  # noqa: E402


def main():
    names = sys.argv[1:] or list(PC.CASES)
    os.makedirs(PC.GOLDEN_DIR, exist_ok=True)
    for name in [n for n in names if n in PC.FWD_CASES]:
        # one whole-batch oracle forward at the launch set the metric times (tests/parity_cases.py FWD_CASES)
        case = PC.FWD_CASES[name]
        t0 = time.time()
        x_in, t, enc, extra = PC.fwd_inputs(case)
        with torch.no_grad():
            pred = PC.oracle_fwd(case)
        np.savez(PC.golden_path(name), pred=pred.numpy(), timestep=np.float64(t), x_in_sum=np.float64(x_in.double().sum().item()),
                 prompts=np.array(PC.fwd_prompts(case), dtype=np.int64),
                 seconds=np.float64(time.time() - t0), threads=np.int64(torch.get_num_threads()))
        print(f"{name}: |pred| = {pred.norm().item():.4f}; wrote {PC.golden_path(name)} in {time.time() - t0:.0f} s", flush=True)
    names = [n for n in names if n not in PC.FWD_CASES]
    for name in names:
        case = PC.CASES[name]
        t0 = time.time()
        P = PC.case_params(case)
        x0, enc, extra = PC.case_inputs(case)
        predict = PC.oracle_predictor(case, P, enc, extra)
        keep = PC.kept_steps(case)
        xin, pred = {}, {}

        def on_step(i, x_in, p):
            if i in keep:
                xin[i], pred[i] = x_in.numpy().copy(), p.float().numpy().copy()
            print(f"{name}: step {i + 1}/{case['steps']}  |pred| = {p.norm().item():.4f}  ({time.time() - t0:.0f} s)", flush=True)

        with torch.no_grad():
            x_end = PC.run_loop(case, predict, x0, on_step)
        s0, rows = PC.schedule(case)
        np.savez(PC.golden_path(name), x_end=x_end.numpy(), kept=np.array(keep, dtype=np.int64), x0_scale=np.float64(s0),
                 sched=np.array(rows, dtype=np.float64),   # per step: model timestep, input scale, a, b  (x <- a x + b pred)
                 x_in=np.stack([xin[i] for i in keep]), pred=np.stack([pred[i] for i in keep]),
                 seconds=np.float64(time.time() - t0), threads=np.int64(torch.get_num_threads()))
        print(f"{name}: wrote {PC.golden_path(name)} in {time.time() - t0:.0f} s", flush=True)
        del P, predict


if __name__ == "__main__":
    main()
