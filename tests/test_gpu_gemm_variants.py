"""-m gpu: every selectable variant of the software-pipelined GEMM loop (csrc/gemm_pipe.hip) runs on the hardware in the suite, not
only the default: loader waves off / on for every 256x160 launch / with interleaved fragment reads / chosen by shape (the default).
The variants change WHO issues the LDS-DMA and WHEN fragments are read, never the accumulation order: outputs are bit-identical."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gemm_variant_child.py")], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("VARIANT_JSON ")][-1][len("VARIANT_JSON "):])


def test_loader_wave_variants_are_bit_identical():
    base = _run({"MI355X_SD_GEMM_LOADERS": "0"})
    for k, v in base.items():
        assert v["rel"] < 4e-3, (k, v)
    for mode in ("4", "5", "-1"):
        got = _run({"MI355X_SD_GEMM_LOADERS": mode})
        for k in base:
            assert got[k]["sha"] == base[k]["sha"], (mode, k, got[k], base[k])
    # and the library's default (no variable) is one of them
    dflt = _run({})
    assert all(dflt[k]["sha"] == base[k]["sha"] for k in base)
