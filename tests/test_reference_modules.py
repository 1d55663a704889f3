"""The oracle against the REFERENCE'S OWN CODE.

scripts/make_reference_golden.py ran the reference's unmodified modules (ppdiffusers/ppdiffusers/models/*.py and
schedulers/*.py, loaded from /root/reference, executed over oracle/paddle_shim.py) on the cases of tests/reference_cases.py and
committed their outputs. Here:
  * everywhere (CPU): the oracle reproduces those committed reference outputs;
  * where /root/reference exists (the build container): the reference is run again, live, and compared with both.
Every device parity test (-m gpu) is a comparison with this oracle, so this file is what ties "parity with the oracle" to
"parity with the reference's code". What it cannot tie down is Paddle's own kernels: the array operations under the
reference's code are torch's fp32 CPU ones (oracle/paddle_shim.py).
"""
import os

import numpy as np
import pytest
import torch

from oracle import reference_runner
from tests import reference_cases as RC


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


@pytest.mark.parametrize("name", list(RC.CASES))
def test_oracle_reproduces_the_committed_reference_outputs(name):
    assert os.path.isfile(RC.golden_path(name)), "regenerate with scripts/make_reference_golden.py (build container)"
    gold = np.load(RC.golden_path(name))
    out = RC.CASES[name](False)["oracle"]
    assert sorted(gold.files) == sorted(out)
    for k in gold.files:
        g = torch.from_numpy(gold[k])
        assert tuple(g.shape) == tuple(out[k].shape), (k, g.shape, out[k].shape)
        assert torch.isfinite(g).all()
        assert _rel(out[k], g) < RC.REL_TOL, (k, _rel(out[k], g))


@pytest.mark.skipif(not reference_runner.available(), reason="/root/reference exists only in the build container")
@pytest.mark.parametrize("name", ["unet_tiny", "unet_mini_xl", "unet_tiny_masks", "controlnet_bgr_guess_mode", "dit_mini", "sd3_mini_trained_norm_bias",
                                  "vae_mini", "sched_euler_sdxl", "sched_dpmpp_2m_karras_heun", "sched_lcm"])
def test_live_reference_run_agrees(name):
    out = RC.CASES[name](True)
    gold = np.load(RC.golden_path(name))
    for k, r in out["reference"].items():
        assert _rel(out["oracle"][k], r) < RC.REL_TOL, (k, _rel(out["oracle"][k], r))
        assert _rel(torch.from_numpy(gold[k]), r) < 1e-6, k       # the committed vectors are this very computation


def test_the_shim_is_test_infrastructure_only():
    """nothing under paddlemix_amd/ (the product), bench.py's timed path or __graft_entry__ may touch the shim or the runner"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for d, _, files in os.walk(os.path.join(root, "paddlemix_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if "paddle_shim" in src or "reference_runner" in src:
                    offenders.append(os.path.join(d, f))
    for f in ("bench.py", "__graft_entry__.py"):
        src = open(os.path.join(root, f)).read()
        if "paddle_shim" in src or "reference_runner" in src:
            offenders.append(f)
    assert not offenders, offenders


def test_reference_parameter_names_are_the_oracles():
    """load_params refuses a parameter dictionary whose names differ from the reference layer's state dict"""
    if not reference_runner.available():
        pytest.skip("build container only")
    from oracle import unet_ref as U
    from tests.configs import TINY
    P = U.synth_unet_params(TINY, seed=1)
    bad = dict(P)
    bad["conv_in.weight_typo"] = bad.pop("conv_in.weight")
    with pytest.raises(KeyError, match="parameter names differ"):
        reference_runner.build_unet(TINY, bad)
