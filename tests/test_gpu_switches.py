"""-m gpu: the A/B switches of the launchers (each selects a code path that is also live under other conditions) run in the suite:
kept => tested. Since round 5 they exist in the debug-switch build only (libmi355x_sd_dbg.so, MI355X_SD_LIB=dbg; csrc/common.h
sd_switch) -- the production libraries never read the environment, which the last test checks. Every switch gives the default's
result within the kernels' own tolerance; the ones that only change HOW a result is stored give the same bits.
  MI355X_SD_ATTN_NO_M16    d = 64 unmasked attention on the 32x32x16 MFMA kernel (what masked / unaligned launches take) instead of
                           the 16x16x32 one
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
        env = dict(os.environ)
        env.update(MI355X_SD_LIB="dbg")
        env.update(env_extra)
        if env_extra.get("MI355X_SD_LIB") == "":   # (the production library)
            env.pop("MI355X_SD_LIB")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_child.py")], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        _CACHE[key] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("SWITCH_JSON ")][-1][len("SWITCH_JSON "):])
    return _CACHE[key]


M16 = {"MI355X_SD_ATTN_NO_M16": "1"}


@pytest.mark.parametrize("env,base_env,same_bits",
                         [(M16, {}, False), ({"MI355X_SD_ATTN_NO_SHORT": "1"}, {}, False),
                          ({"MI355X_SD_ATTN_NO_SHORT": "1", "MI355X_SD_ATTN_NO_QT": "1"}, {}, False),
                          # 8-byte O stores: the 32x32x16 kernel's own switch (the 16x16x32 kernel has the 16-byte form only and hands
                          # unaligned outputs to the other kernel), so it is compared on that kernel: same bits
                          (dict(M16, MI355X_SD_ATTN_NO_WIDE="1"), M16, True), ({"MI355X_SD_ATTN_NO_WIDE": "1"}, {}, False),
                          ({"MI355X_SD_NO_SPLITK": "1"}, {}, False), ({"MI355X_SD_NO_GN_FUSED": "1"}, {}, False),
                          ({"MI355X_SD_NO_WIDEN_F8": "1"}, {}, False)],
                         ids=["attn-no-m16", "attn-no-short", "attn-no-short-no-qt", "attn-no-wide-on-the-32x32-kernel", "attn-no-wide",
                              "no-splitk", "no-gn-fused", "no-widen-f8"])
def test_switch_gives_the_defaults_result(env, base_env, same_bits):
    base = _run(base_env)
    got = _run(env)
    for k, v in base.items():
        assert v["rel"] < 5e-3, (k, v)                     # the default path against fp32 math
        assert got[k]["rel"] < 5e-3, (env, k, got[k])      # the switched path against fp32 math
        if same_bits:
            assert got[k]["sha"] == v["sha"], (env, k, got[k], v)


def test_production_library_ignores_the_switches():
    """libmi355x_sd.so with every switch of the list set: bit for bit what it computes with a clean environment, and what the
    debug-switch build computes with a clean environment (same sources, same defaults). In the debug build the same environment
    changes bits (split-K off, the other attention kernel)."""
    every = {"MI355X_SD_ATTN_NO_M16": "1", "MI355X_SD_ATTN_NO_SHORT": "1", "MI355X_SD_ATTN_NO_QT": "1", "MI355X_SD_ATTN_NO_WIDE": "1",
             "MI355X_SD_NO_SPLITK": "1", "MI355X_SD_NO_GN_FUSED": "1", "MI355X_SD_NO_WIDEN_F8": "1", "MI355X_SD_GEMM_TILE": "128",
             "MI355X_SD_NO_PIPE": "1"}
    prod_clean, prod_env, dbg_clean, dbg_env = _run({"MI355X_SD_LIB": ""}), _run(dict(every, MI355X_SD_LIB="")), _run({}), _run(every)
    for k, v in prod_clean.items():
        assert prod_env[k]["sha"] == v["sha"] == dbg_clean[k]["sha"], (k, v, prod_env[k], dbg_clean[k])
    assert any(dbg_env[k]["sha"] != v["sha"] for k, v in dbg_clean.items())
