"""-m gpu: every selectable variant of the software-pipelined GEMM loops (csrc/gemm_pipe.hip) runs on the hardware in the suite, not
only the default, and the loops are held to an independent implementation: the generic loop of csrc/gemm.hip
(MI355X_SD_NO_PIPE=1). The variants change WHEN operands are staged and fetched, never the accumulation order: bit-identical."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_CACHE = {}


def _bar(key):   # rel-L2 of a case against its fp32 reference (16-bit stores)
    return 4e-3


def _run(env_extra):
    key = tuple(sorted(env_extra.items()))
    if key not in _CACHE:
        _CACHE[key] = _run_child(env_extra)
    return _CACHE[key]


def _run_child(env_extra):
    env = dict(os.environ, MI355X_SD_LIB="dbg", **env_extra)   # the A/B switches exist in the debug-switch build only
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gemm_variant_child.py")], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("VARIANT_JSON ")][-1][len("VARIANT_JSON "):])


def test_epilogue_operand_variants():
    """Round-3 epilogue forms (csrc/gemm_epilogue.h): residual fetched during the last K iterations, a row-tile's operands fetched in
    one batch, bias as the accumulators' initial value. The first two change WHEN operands are loaded, never the arithmetic:
    bit-identical to the group-at-a-time epilogue. The bias moves from the end of the fp32 sum to its start: same tolerance."""
    base = _run({})
    for k, v in base.items():
        assert v["rel"] < _bar(k), (k, v)
    for env in ({"MI355X_SD_GEMM_NO_PRE": "1"}, {"MI355X_SD_GEMM_NO_EPI_BATCH": "1"},
                {"MI355X_SD_GEMM_NO_PRE": "1", "MI355X_SD_GEMM_NO_EPI_BATCH": "1"},
                {"MI355X_SD_GEMM_PERSIST": "0"}):   # one block per tile instead of persistent blocks walking several
        got = _run(env)
        for k in base:
            assert got[k]["sha"] == base[k]["sha"], (env, k, got[k], base[k])
    got = _run({"MI355X_SD_GEMM_NO_BIAS_ACC": "1"})
    for k, v in got.items():
        assert v["rel"] < _bar(k), (k, v)


def test_small_tile_ring_is_bit_identical_to_the_128_tile_without_slices():
    """csrc/gemm_small.hip (round 6): the 64 x 64 tile with the six-stage ring that the small launches of a batch-1 step take instead
    of split-K slices + a reduce kernel. Against the 128 x 128 tiles walking the same K in one chain (MI355X_SD_NO_SPLITK on both
    sides, MI355X_SD_NO_SMALL on the reference's): another tile, another ring depth, the same products in the same order -> the same
    bits on every case of the child (ragged M / N, one to 35 K-tiles, residual rows inside a wider buffer, GEGLU, LayerNorm-free).
    Against the sliced form it replaces the sums associate differently: equal within the kernels' tolerance (the child's rel)."""
    new = _run({"MI355X_SD_NO_SPLITK": "1"})
    ref = _run({"MI355X_SD_NO_SPLITK": "1", "MI355X_SD_NO_SMALL": "1"})
    for k, v in new.items():
        assert v["rel"] < _bar(k), (k, v)
        assert v["sha"] == ref[k]["sha"], (k, v, ref[k])
    sliced = _run({"MI355X_SD_NO_SMALL": "1"})
    assert any(sliced[k]["sha"] != v["sha"] for k, v in _run({}).items())   # (the picker does take the small tile somewhere)


@pytest.mark.parametrize("tile_map", ["160:129", "257:129,320:129"])
def test_four_wave_tiles(tile_map):
    """The four-wave tiles built for two co-resident blocks per CU (gemm_cfg.h: 128x160) substituted for the
    picker's choice: every output element still accumulates its K products in the same order -> bit-identical."""
    base = _run({})
    got = _run({"MI355X_SD_GEMM_TILE_MAP": tile_map})
    for k in base:
        assert got[k]["rel"] < _bar(k), (tile_map, k, got[k])
        # same loop, other tile: bit-identical. (Launches the picker gives to the phased 256x256 kernel, id 257, walk K in another
        # order: equal within the tolerance only.)
        if "257" not in tile_map:
            assert got[k]["sha"] == base[k]["sha"], (tile_map, k, got[k], base[k])


@pytest.mark.parametrize("tile_env", [{}, {"MI355X_SD_GEMM_TILE": "320"}, {"MI355X_SD_GEMM_TILE": "128"}, {"MI355X_SD_GEMM_TILE": "129"},
                                      {"MI355X_SD_GEMM_TILE": "160"}, {"MI355X_SD_GEMM_TILE": "256"}, {"MI355X_SD_GEMM_TILE": "258"},
                                      {"MI355X_SD_GEMM_TILE": "259"}],
                         ids=["picker", "256x320", "128x128", "128x160", "256x160", "256x256-pipelined", "256x256-four-waves", "256x160-four-waves"])
def test_pipelined_loops_are_bit_identical_to_the_generic_loop(tile_env):
    """The interleaved register-pipelined loop (round 4: one LDS-DMA piece / one fragment read between MFMA pairs, the A and W
    pieces of a tile issued in different half-iterations, unrolled tail or -- early-residual kernels -- a tail loop, static vmcnt)
    and the streaming loop against the plain double-buffered loop of csrc/gemm.hip on every tile family: another schedule, another
    author's-week of code, the same tiles and the same K order -> the same bits. (Bias added in the epilogue on both sides: the
    pipelined kernels otherwise start their accumulators at the bias.) Covers two and three LDS stages, the early-residual
    kernels, GEGLU, ragged M / N, launches of 1 .. 5 K-tiles (the tail alone), split-K slices and the implicit-GEMM convs.
    The same comparison held the interleaved loop to the round-3 burst loop before that was deleted (profiles/r04_s2_tests.txt)."""
    ref = _run(dict(tile_env, MI355X_SD_NO_PIPE="1", MI355X_SD_GEMM_NO_BIAS_ACC="1"))
    new = _run(dict(tile_env, MI355X_SD_GEMM_NO_BIAS_ACC="1"))
    for k, v in new.items():
        assert v["rel"] < _bar(k), (tile_env, k, v)
        assert v["sha"] == ref[k]["sha"], (tile_env, k, v, ref[k])
