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


_CACHE = {}


def _run(env_extra):
    key = tuple(sorted(env_extra.items()))
    if key not in _CACHE:
        _CACHE[key] = _run_child(env_extra)
    return _CACHE[key]


def _run_child(env_extra):
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


def test_epilogue_operand_variants():
    """Round-3 epilogue forms (csrc/gemm_epilogue.h): residual fetched during the last K iterations, a row-tile's operands fetched in
    one batch, bias as the accumulators' initial value. The first two change WHEN operands are loaded, never the arithmetic:
    bit-identical to the group-at-a-time epilogue. The bias moves from the end of the fp32 sum to its start: same tolerance."""
    base = _run({})
    for k, v in base.items():
        assert v["rel"] < 4e-3, (k, v)
    for env in ({"MI355X_SD_GEMM_NO_PRE": "1"}, {"MI355X_SD_GEMM_NO_EPI_BATCH": "1"},
                {"MI355X_SD_GEMM_NO_PRE": "1", "MI355X_SD_GEMM_NO_EPI_BATCH": "1"},
                {"MI355X_SD_GEMM_PERSIST": "0"}):   # one block per tile instead of persistent blocks walking several
        got = _run(env)
        for k in base:
            assert got[k]["sha"] == base[k]["sha"], (env, k, got[k], base[k])
    got = _run({"MI355X_SD_GEMM_NO_BIAS_ACC": "1"})
    for k, v in got.items():
        assert v["rel"] < 4e-3, (k, v)


@pytest.mark.parametrize("tile_map", ["160:129", "257:129,320:129"])
def test_four_wave_tiles(tile_map):
    """The four-wave tiles built for two co-resident blocks per CU (gemm_cfg.h: 128x160) substituted for the
    picker's choice: every output element still accumulates its K products in the same order -> bit-identical."""
    base = _run({})
    got = _run({"MI355X_SD_GEMM_TILE_MAP": tile_map})
    for k in base:
        assert got[k]["rel"] < 4e-3, (tile_map, k, got[k])
        # same loop, other tile: bit-identical. (Launches the picker gives to the phased 256x256 kernel, id 257, walk K in another
        # order: equal within the tolerance only.)
        if "257" not in tile_map:
            assert got[k]["sha"] == base[k]["sha"], (tile_map, k, got[k], base[k])


@pytest.mark.parametrize("tile_env", [{}, {"MI355X_SD_GEMM_TILE": "320"}, {"MI355X_SD_GEMM_TILE": "128"}, {"MI355X_SD_GEMM_TILE": "129"},
                                      {"MI355X_SD_GEMM_TILE": "160"}, {"MI355X_SD_GEMM_TILE": "257", "MI355X_SD_PIPE256": "1"}],
                         ids=["picker", "256x320", "128x128", "128x160", "256x160", "256x256-pipelined"])
def test_interleaved_k_loop_is_bit_identical_to_the_burst_loop(tile_env):
    """Round 4: the interleaved K loop (csrc/gemm_pipe.hip, template IL: one LDS-DMA piece / one fragment read between small MFMA
    groups, the A and W pieces of a tile issued in different half-iterations, unrolled tail, static vmcnt) against the round-3 loop
    (MI355X_SD_GEMM_IL=0) on every tile family: WHEN operands are staged changes, the accumulation order does not -- same bits.
    Covers the register-pipelined and the streaming form, two and three LDS stages, the early-residual kernels, GEGLU, ragged M / N,
    launches of 1 .. 3 K-tiles (the unrolled tail alone) and the implicit-GEMM convs."""
    old = _run(dict(tile_env, MI355X_SD_GEMM_IL="0", MI355X_SD_GEMM_LOADERS="0"))
    new = _run(dict(tile_env, MI355X_SD_GEMM_IL="1", MI355X_SD_GEMM_LOADERS="0"))
    for k, v in new.items():
        assert v["rel"] < 4e-3, (tile_env, k, v)
        assert v["sha"] == old[k]["sha"], (tile_env, k, v, old[k])


def test_interleaved_loop_schedules_are_bit_identical():
    """MI355X_SD_GEMM_IL=2|3|4: other placements of the reads and LDS-DMA pieces among the MFMAs of a half-iteration (256x160 tile;
    csrc/gemm_pipe.hip il_piece_step) -- same bits as the default schedule."""
    base = _run({"MI355X_SD_GEMM_IL": "1", "MI355X_SD_GEMM_LOADERS": "0"})
    for v in ("2", "3", "4"):
        got = _run({"MI355X_SD_GEMM_IL": v, "MI355X_SD_GEMM_LOADERS": "0"})
        for k in base:
            assert got[k]["sha"] == base[k]["sha"], (v, k, got[k], base[k])
