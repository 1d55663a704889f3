"""CPU: the two parsers of the C library that read what a HOST hands them -- a program file (mi355x_sd_program_load, the exported
step of any model) and the UNet config text (mi355x_sd_unet_create) -- against damaged input: truncations, flipped bytes, extreme
32- and 64-bit fields, wrong JSON types, lists of the wrong length, absurd nesting. Every call must return a status (a refusal with a
message); none may fault. The fuzzing runs in a child process (tests/fuzz_child.py) so that a fault fails the test instead of the
test session. Seeds are fixed: a failure reproduces with the command in the assertion message."""
import os
import subprocess
import sys

import pytest

from tests.abi_emulator import on_emulator
from tests import export_cases as EC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "fuzz_child.py")


def _run(args):
    cmd = [sys.executable, CHILD] + [str(a) for a in args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{' '.join(cmd)}\nexit {r.returncode} (negative: killed by that signal)\n{r.stdout[-400:]}\n{r.stderr[-1500:]}"
    assert "no fault" in r.stdout
    return r.stdout


@pytest.mark.parametrize("name,seed", [("unet_tiny", 1), ("unet_tiny_masked_controlnet", 2), ("sd3_mini", 3)])
def test_program_loader_survives_damaged_files(name, seed, tmp_path):
    from paddlemix_amd.export import export_program
    model, run, outputs = on_emulator(EC.build, name, True)
    run()
    path = str(tmp_path / (name + ".mi3prg"))
    export_program(model, EC.last_plan(model), path, outputs)
    out = _run(["program", path, 400, seed])
    n_ref = int(out.split(" refused")[0].split()[-1])
    assert n_ref > 100, out     # (an integer or float ARGUMENT of a launch is data to the loader: such damage is accepted)


@pytest.mark.parametrize("seed", [11, 12])
def test_config_parser_survives_damaged_text(seed):
    out = _run(["config", 1500, seed])
    n_ref = int(out.split(" refused")[0].split()[-1])
    assert n_ref > 500, out
