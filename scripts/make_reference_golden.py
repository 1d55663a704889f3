"""Run the REFERENCE's own modules (ppdiffusers/ppdiffusers/models/*.py, schedulers/*.py, unmodified, from /root/reference) over
oracle/paddle_shim.py on the cases of tests/reference_cases.py, check the oracle against them, and store the reference's
outputs under tests/golden/reference_modules/<case>.npz (fp32) for the machines where /root/reference does not exist.

    python scripts/make_reference_golden.py [case ...]         (build container only; CPU, about a minute)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import reference_runner  # noqa: E402
from tests import reference_cases as RC  # noqa: E402

if not reference_runner.available():
    sys.exit("/root/reference is not present: the golden vectors can only be regenerated in the build container")
os.makedirs(RC.GOLDEN_DIR, exist_ok=True)
names = sys.argv[1:] or list(RC.CASES)
worst = 0.0
for name in names:
    out = RC.CASES[name](True)
    line = []
    for k, r in out["reference"].items():
        o = out["oracle"][k].float()
        r = r.float()
        assert o.shape == r.shape, (name, k, o.shape, r.shape)
        rel = float((o - r).abs().max() / r.abs().max())
        worst = max(worst, rel)
        assert rel < RC.REL_TOL, f"{name}.{k}: oracle differs from the reference by {rel:.3g} (relative to max |reference|)"
        line.append(f"{k} {rel:.2g}")
    np.savez(RC.golden_path(name), **{k: v.float().numpy() for k, v in out["reference"].items()})
    print(f"{name:34s} max |oracle - reference| / max |reference|:  " + "  ".join(line), flush=True)
print(f"{len(names)} cases, worst relative difference {worst:.3g} (tolerance {RC.REL_TOL:g})")
