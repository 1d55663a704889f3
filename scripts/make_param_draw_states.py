"""tests/golden/parity/param_draw_states.npz: the CPU generator's state at the shard cut points of the seeded parameter draws
(tests/parity_cases.py case_params), recorded while drawing each family's set serially -- seed 1234, construction order, the
definition of SURVEY.md 8(d).  Run here on the CPU (about three minutes):

    python scripts/make_param_draw_states.py

What is stored is generator state (5056 bytes per cut point), not parameters; tests/test_parity_cases.py checks on a small
configuration that the sharded draw equals the serial one and that every family's first and last shard reproduce.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import parity_cases as PC  # noqa: E402


def record(case, n=PC.DRAW_SHARDS):
    synth, shapes = PC._synth(case)
    bounds = PC.shard_bounds(shapes, n)
    g = torch.Generator().manual_seed(1234)
    states = []
    for i in range(n):
        states.append(g.get_state().clone())
        synth(case["cfg"], generator=g, only=range(bounds[i], bounds[i + 1]))
    states.append(g.get_state().clone())
    return torch.stack(states).numpy(), np.asarray(bounds, dtype=np.int64)


def main():
    out = {}
    for name in ("sdxl_1x4x32x32_euler30", "sd15_1x4x64x64_ddim50", "sd3_1x16x64x64_flow28"):
        case = PC.CASES[name]
        fam = PC.family(case)
        out[fam + "_states"], out[fam + "_bounds"] = record(case)
        out[fam + "_sig"] = np.asarray(PC.shapes_sig(PC._synth(case)[1]))
        print(fam, out[fam + "_states"].shape, out[fam + "_bounds"].tolist(), flush=True)
    np.savez(PC.DRAW_STATES, **out)
    print("wrote", PC.DRAW_STATES, os.path.getsize(PC.DRAW_STATES), "bytes")


if __name__ == "__main__":
    main()
