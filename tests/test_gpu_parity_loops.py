"""-m gpu: north_star's parity bar on the real depth. "Latents within 1e-3 rel-err of the CPU reference": the full SDXL parameter
set (2.6 B parameters), 30 Euler steps, free-running device loop against the committed oracle trajectory
(tests/golden/parity/sdxl_1x4x32x32_euler30.npz, made by scripts/make_parity_golden.py; oracle = torch-CPU restatement of
ppdiffusers, pinned to the reference's own module code by tests/test_reference_modules.py), in the four device modes {bf16, fp16 elements} x {16-bit, fp32 residual stream}.

The ABSOLUTE target is asserted where it is met (fp16 elements; must pass) and recorded as a strict xfail where it is not (the bf16
headline: the 2^-9 operand rounding of ~230 sequential contractions) -- so the suite states the gap instead of
hiding it behind "1.5 x whatever was measured". All four numbers, the other geometries (128x128 latents, SD-1.5 / 50 DDIM steps,
SD3 / 28 flow-matching steps with 16-bit and fp8 weights) and the per-step prediction errors are in profiles/r03_parity.json
(scripts/parity_loops.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = "sdxl_1x4x32x32_euler30"
TARGET = 1e-3


@pytest.fixture(scope="module")
def loops(tmp_path_factory):
    """one child process per library build (one element type per process); the seeded weights are drawn once and shared"""
    cache = str(tmp_path_factory.mktemp("parity_params"))
    out = {}
    for elem in ("bf16", "fp16"):
        env = dict(os.environ, MI355X_SD_DTYPE=elem)
        env.pop("MI355X_SD_RESID", None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_loops.py"), "--child", elem, "--cases", CASE,
                            "--cache-dir", cache], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("PARITY_JSON ")][-1]
        out[elem] = json.loads(line[len("PARITY_JSON "):])[CASE]
    print("full SDXL, 30 Euler steps, rel-L2 vs the oracle trajectory:", json.dumps(out))
    return out


def test_fp16_fp32_stream_meets_the_latents_target(loops):
    r = loops["fp16"]["resid_fp32"]
    assert r["end_latents_rel"] < TARGET, r            # the configuration bench.py reports as "parity_mode"
    assert r["pred_rel_teacher_forced_max"] < 2.5e-3, r


def test_fp16_16bit_stream_meets_the_latents_target(loops):
    r = loops["fp16"]["resid_16"]
    assert r["end_latents_rel"] < TARGET, r
    assert r["pred_rel_teacher_forced_max"] < 4e-3, r


@pytest.mark.xfail(strict=True, reason="bf16 elements miss north_star's 1e-3 on the end latents: 2.6e-3 (16-bit stream) measured on "
                                       "the full SDXL depth in round 2, profiles/r02_parity.json -- operand rounding 2^-9 x ~230 "
                                       "sequential contractions; the fp16 build meets it")
def test_bf16_16bit_stream_latents_target(loops):
    assert loops["bf16"]["resid_16"]["end_latents_rel"] < TARGET


@pytest.mark.xfail(strict=True, reason="bf16 elements + fp32 residual stream: 1.6e-3 on the end latents (round 2), still above 1e-3")
def test_bf16_fp32_stream_latents_target(loops):
    assert loops["bf16"]["resid_fp32"]["end_latents_rel"] < TARGET


def test_bf16_regression_bars(loops):
    """what the bf16 headline DOES hold (regression guards at 1.5 x the measured values of profiles/r02_parity.json)"""
    assert loops["bf16"]["resid_16"]["end_latents_rel"] < 1.5 * 2.6e-3 and loops["bf16"]["resid_fp32"]["end_latents_rel"] < 1.5 * 1.63e-3
    assert loops["bf16"]["resid_16"]["pred_rel_teacher_forced_max"] < 1.5 * 1.6e-2
    assert loops["bf16"]["resid_fp32"]["end_latents_rel"] < loops["bf16"]["resid_16"]["end_latents_rel"]
