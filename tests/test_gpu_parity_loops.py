"""-m gpu: north_star's parity bar on the real depth, every full-depth fixture of tests/golden/parity in the suite (round 4).
"Latents within 1e-3 rel-err of the CPU reference": free-running device loops against the committed oracle trajectories
(scripts/make_parity_golden.py; oracle = torch-CPU restatement of ppdiffusers, its stored predictions reproduced bit for bit by the
reference's own model code over the paddle shim, profiles/r03_parity_fixtures_vs_reference.txt):

  * full SDXL (2.6 B parameters): 30 Euler steps at 1x4x32x32; 10 AND 30 Euler steps at 1x4x128x128 (one prompt of the headline
    geometry over the metric's whole schedule);
  * full SD-1.5, 50 DDIM steps at 1x4x64x64 (BASELINE config 2); SD3-medium, 28 flow-matching steps at 1x16x64x64 (config 5);
  * the SD3 loop in the fp8 modes against oracle trajectories computed on the SAME quantised operands (weight-only e4m3; W8A8).

The ABSOLUTE target is asserted where it is met -- fp16 elements, with the 16-bit residual stream of the headline as well as with
the fp32 stream: must pass -- and recorded as a strict xfail with the measured number where it is not (bf16 elements: the 2^-9
operand rounding of ~230 sequential contractions), so the suite states the gap instead of hiding it behind
"1.5 x whatever was measured". Measured values: profiles/r04_parity.json (scripts/parity_loops.py, the same children)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGET = 1e-3
# (the 10-step loop at 1x4x128x128 is the first third of the 30-step one: fixture kept for scripts/parity_loops.py, not replayed here)
SDXL_CASES = ("sdxl_1x4x32x32_euler30", "sdxl_1x4x128x128_euler30")
OTHER_CASES = ("sd15_1x4x64x64_ddim50", "sd3_1x16x64x64_flow28")
FP8_CASES = ("sd3_1x16x64x64_flow28_fp8w", "sd3_1x16x64x64_flow28_w8a8")
FWD_CASE = "sdxl_8x4x128x128_fwd"   # one whole-batch forward at the launch set bench.py times (tests/parity_cases.py FWD_CASES)
SD3_FWD_CASES = ("sd3_8x16x128x128_fwd", "sd3_8x16x128x128_fwd_fp8w", "sd3_8x16x128x128_fwd_w8a8")   # config 5 at its own geometry
ALL = SDXL_CASES + (FWD_CASE,) + OTHER_CASES + FP8_CASES + SD3_FWD_CASES


@pytest.fixture(scope="module")
def loops():
    """one child process per library build (one element type per process)"""
    out, procs = {}, {}
    # the two children run SIDE BY SIDE (round 6: the suite's longest fixture): most of a child's wall time is host work -- drawing the
    # seeded weights (parallel shards, tests/parity_cases.py case_params), float64 latent state, model builds -- and the GPU has room
    # for both. The fp16 child walks the cases in reverse (the two then build different model families at any one time).
    for elem in ("bf16", "fp16"):
        env = dict(os.environ, MI355X_SD_DTYPE=elem)
        env.pop("MI355X_SD_RESID", None)
        order = ALL if elem == "bf16" else tuple(reversed(ALL))
        procs[elem] = subprocess.Popen([sys.executable, os.path.join(ROOT, "scripts", "parity_loops.py"), "--child", elem, "--cases", ",".join(order)],
                                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for elem, p in procs.items():
        try:
            so, se = p.communicate(timeout=3000)
        except subprocess.TimeoutExpired:
            for q in procs.values():
                q.kill()
            raise
        assert p.returncode == 0, se[-3000:]
        print("\n".join(ln for ln in so.splitlines() if ln.startswith("PARITY_TIMING ")))
        line = [ln for ln in so.splitlines() if ln.startswith("PARITY_JSON ")][-1]
        out[elem] = json.loads(line[len("PARITY_JSON "):])
    print("full-depth loops, rel-L2 of the end latents vs the oracle trajectories:",
          json.dumps({e: {c: {m: round(r["end_latents_rel"], 6) for m, r in v.items()} for c, v in d.items() if not c.endswith("_fwd") and "_fwd_" not in c} for e, d in out.items()}))
    print("whole-batch forward 8x4x128x128 vs the oracle's:", json.dumps({e: d[FWD_CASE] for e, d in out.items()}))
    print("SD3 whole-batch forward 8x16x128x128 vs the oracle's:", json.dumps({e: {c: d[c] for c in SD3_FWD_CASES if c in d} for e, d in out.items()}))
    return out


def test_whole_batch_forward_at_the_timed_launch_set(loops):
    """Round 5 (VERDICT r4 missing #3): every loop fixture runs ONE prompt -- GEMMs of M = 1024 / 4096 rows, other tiles, split-K --
    while bench.py times bs 8 (M = 8192 / 32768: 256x160 persistent tiles, the 256x320 streaming FF1, no split-K). One oracle forward
    of the whole seeded 8x4x128x128 batch (tests/golden/parity/sdxl_8x4x128x128_fwd.npz, 145 s of CPU) pins those launches:
    per-prediction bars of the loops' teacher-forced check (bf16 1.6e-2, fp16 2.5e-3), per prompt as well as over the batch.
    Batch-size consistency: a row of the bs-8 prediction vs the same prompt alone. NOT bit-identical and not 1e-5 -- the bs-1 launches
    take other tiles and split-K, fp32 sums in another order flip 16-bit roundings of the stored activations, and ~230 layers
    amplify that like any other rounding -- the bar is that the batch effect stays BELOW the distance to the oracle (it is a
    re-draw of the same rounding noise, not an error of its own)."""
    for elem, bar in (("bf16", 1.6e-2), ("fp16", 2.5e-3)):
        for mode, r in loops[elem][FWD_CASE].items():
            assert r["pred_rel_bs8"] < bar and r["pred_rel_bs8_per_prompt_max"] < 1.25 * bar, (elem, mode, r)
            assert r["bs8_row_vs_bs1_forward_rel_max"] < 1.5 * r["pred_rel_bs8_per_prompt_max"], (elem, mode, r)


def test_sd3_whole_batch_forward_at_the_config5_geometry(loops):
    """Round 6 (VERDICT r5 missing #1): BASELINE config 5 is SD3-medium 1024^2 bs 8 -- latents 8x16x128x128, joint sequence 4096 + 154 --
    and every SD3 loop fixture is 1x16x64x64. One oracle forward of the whole seeded batch (tests/golden/parity/sd3_8x16x128x128_fwd.npz)
    and its twins on the fp8 modes' own quantised operands (two prompts of the eight each) pin the launches the sd3-* bench lines time:
    33 k-row GEMM tiles, the 4,250-key joint attention, the widened-fp8 and W8A8 GEMMs at M >= 4096. Per-prediction bars of the loops'
    teacher-forced check: bf16 1.6e-2, fp16 2.5e-3; weight-only fp8 vs the same-operand oracle the bf16 bar; W8A8 2 x (a rounding tie
    decided differently by the bf16 producer and the fp32 oracle moves a whole e4m3 step of an activation)."""
    for elem, bar in (("bf16", 1.6e-2), ("fp16", 2.5e-3)):
        (mode, r), = loops[elem]["sd3_8x16x128x128_fwd"].items()
        assert r["pred_rel_bs8"] < bar and r["pred_rel_bs8_per_prompt_max"] < 1.25 * bar, (elem, mode, r)
    (_, r), = loops["bf16"]["sd3_8x16x128x128_fwd_fp8w"].items()
    assert r["pred_rel_bs8"] < 1.6e-2 and r["pred_rel_bs8_per_prompt_max"] < 1.25 * 1.6e-2, r
    (_, r), = loops["bf16"]["sd3_8x16x128x128_fwd_w8a8"].items()
    assert r["pred_rel_bs8"] < 3.2e-2 and r["pred_rel_bs8_per_prompt_max"] < 1.25 * 3.2e-2, r


@pytest.mark.parametrize("case", SDXL_CASES + OTHER_CASES)
def test_fp16_meets_the_latents_target(loops, case):
    """fp16 elements: both residual-stream types (UNets) / 16-bit weights (SD3) -- the mode bench.py reports as "parity_mode" is
    fp16 + the 16-bit stream, i.e. the headline kernels on the other element type"""
    for mode, r in loops["fp16"][case].items():
        assert r["end_latents_rel"] < TARGET, (case, mode, r)
        assert r["pred_rel_teacher_forced_max"] < 4e-3, (case, mode, r)


# bf16 elements, measured on the MI355X in round 3 (profiles/r03_parity.json): end latents, 16-bit / fp32 residual stream
BF16_R03 = {"sdxl_1x4x32x32_euler30": (2.37e-3, 1.58e-3),
            "sdxl_1x4x128x128_euler30": (None, None), "sd15_1x4x64x64_ddim50": (1.42e-3, 9.3e-4), "sd3_1x16x64x64_flow28": (2.26e-3, None)}


@pytest.mark.parametrize("case", SDXL_CASES + OTHER_CASES)
@pytest.mark.xfail(strict=True, reason="bf16 elements with the 16-bit stream miss north_star's 1e-3 on the end latents at every full-depth "
                                       "geometry (1.4e-3 .. 3.5e-3, profiles/r03_parity.json / r04_parity.json): operand rounding 2^-9 x "
                                       "~230 sequential contractions; the fp16 build meets it")
def test_bf16_16bit_stream_latents_target(loops, case):
    mode = "w16" if case.startswith("sd3") else "resid_16"
    assert loops["bf16"][case][mode]["end_latents_rel"] < TARGET


@pytest.mark.parametrize("case", SDXL_CASES)
@pytest.mark.xfail(strict=True, reason="bf16 elements + fp32 residual stream on the SDXL depth: 1.6e-3 .. 2.2e-3 on the end latents")
def test_bf16_fp32_stream_latents_target_sdxl(loops, case):
    assert loops["bf16"][case]["resid_fp32"]["end_latents_rel"] < TARGET


def test_bf16_fp32_stream_meets_the_target_on_sd15(loops):
    """the one bf16 configuration that does meet the bar: SD-1.5 (half the depth of SDXL), fp32 residual stream (9.3e-4 in round 3)"""
    assert loops["bf16"]["sd15_1x4x64x64_ddim50"]["resid_fp32"]["end_latents_rel"] < TARGET


def test_bf16_regression_bars(loops):
    """what the bf16 headline DOES hold (regression guards at 1.5 x the values measured in round 3)"""
    for case, (r16, r32) in BF16_R03.items():
        got = loops["bf16"][case]
        m16 = got.get("resid_16") or got["w16"]
        assert m16["end_latents_rel"] < 1.5 * (r16 or 3.5e-3), (case, m16)
        assert m16["pred_rel_teacher_forced_max"] < 1.5 * 1.63e-2, (case, m16)
        if "resid_fp32" in got:
            assert got["resid_fp32"]["end_latents_rel"] < min(1.5 * (r32 or 2.2e-3), m16["end_latents_rel"]), (case, got)


@pytest.mark.parametrize("case", FP8_CASES)
def test_fp8_modes_vs_the_oracle_on_the_same_quantised_operands(loops, case):
    """BASELINE config 5: the fp8 modes against an oracle that multiplies by the same e4m3 operands. What remains is what the bf16
    path has on this loop (16-bit activations between the GEMMs; 2.3e-3 in round 3) -- against the unquantised oracle the same
    device loops sit at 1e-2, which is the quantisation the mode chose, not kernel error. Bars: 1.5 x the bf16-weights loop's value
    for weight-only fp8; W8A8 quantises activations at 3 mantissa bits, where a rounding tie decided differently by the device
    (bf16 producer) and the fp32 oracle moves a whole e4m3 step: 3 x."""
    (mode, r), = loops["bf16"][case].items()
    bar = 1.5 * 2.26e-3 if mode == "fp8w" else 3 * 2.26e-3
    assert r["end_latents_rel"] < bar, (case, r)
