"""-m gpu: the A/B switches the library still reads (each selects a code path that is also live under other conditions) run in the
suite: kept => tested. Every switch gives the default's result within the kernels' own tolerance; the ones that only change HOW a
result is stored give the same bits.
  MI355X_SD_ATTN_NO_SHORT  short key sequences on the flash kernel instead of the single-pass kernel (masked launches always do)
  MI355X_SD_ATTN_NO_QT     (with NO_SHORT) one query tile per block instead of two for short keys
  MI355X_SD_ATTN_NO_WIDE   8-byte O stores (the form unaligned outputs take) instead of 16-byte ones: bit-identical
  MI355X_SD_NO_SPLITK      small-M launches without split-K slices (the form a caller without workspace gets)
  MI355X_SD_NO_GN_FUSED    GroupNorm as statistics + apply where the one-launch kernel would run (the form large maps take)
  MI355X_SD_NO_WIDEN_F8    weight-only fp8: e4m3 bytes widened in the generic loop's fragment load (what small-M launches do)
                           instead of once, just in time, in front of the pipelined 16-bit kernels (same products, same K
                           order; the bias enters as the accumulators' initial value there: equal to fp32 rounding)"""
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
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_child.py")], env=dict(os.environ, **env_extra), cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        _CACHE[key] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("SWITCH_JSON ")][-1][len("SWITCH_JSON "):])
    return _CACHE[key]


@pytest.mark.parametrize("env,same_bits", [({"MI355X_SD_ATTN_NO_SHORT": "1"}, False),
                                           ({"MI355X_SD_ATTN_NO_SHORT": "1", "MI355X_SD_ATTN_NO_QT": "1"}, False),
                                           ({"MI355X_SD_ATTN_NO_WIDE": "1"}, True), ({"MI355X_SD_NO_SPLITK": "1"}, False),
                                           ({"MI355X_SD_NO_GN_FUSED": "1"}, False), ({"MI355X_SD_NO_WIDEN_F8": "1"}, False)],
                         ids=["attn-no-short", "attn-no-short-no-qt", "attn-no-wide", "no-splitk", "no-gn-fused", "no-widen-f8"])
def test_switch_gives_the_defaults_result(env, same_bits):
    base = _run({})
    got = _run(env)
    for k, v in base.items():
        assert v["rel"] < 5e-3, (k, v)                     # the default path against fp32 math
        assert got[k]["rel"] < 5e-3, (env, k, got[k])      # the switched path against fp32 math
        if same_bits:
            assert got[k]["sha"] == v["sha"], (env, k, got[k], v)
