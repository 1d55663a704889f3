"""The C-ABI library loads and exports exactly the symbols include/mi355x_sd.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

from paddlemix_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mi355x_sd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355x_sd_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_everything():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.mi355x_sd_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.mi355x_sd_last_error(), bytes)
    # pure host-side helper: no device needed
    assert lib.mi355x_sd_groupnorm_workspace_floats(8, 16384, 320) > 0


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmi355x_sd.so")
    with pytest.raises(_lib.MI355XError, match="no CPU"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "paddlemix_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert not re.search(r"^\s*(from|import)\s+tests\b", txt, flags=re.M), f


def test_early_residual_kernels_keep_their_landing_registers():
    """csrc/gemm_pipe.hip gemm_pipe_pre_kernel: the residual rows land in v216 .. v255 through hand-written loads; nothing the
    compiler generated may name those registers (amdgpu_num_vgpr(216) is a budget, not a reservation -- round 4 hit exactly
    that). Disassembles both built libraries; no GPU needed."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tools = [os.path.join("/opt/rocm/lib/llvm/bin", t) for t in ("llvm-objdump", "llvm-objcopy")]
    libs = [os.path.join(root, "paddlemix_amd", n) for n in ("libmi355x_sd.so", "libmi355x_sd_f16.so", "libmi355x_sd_dbg.so")]
    if not all(os.path.exists(t) for t in tools + libs):
        pytest.skip("needs the ROCm LLVM tools and both built libraries")
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_landing_zone.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    import re
    # the kernels are recognised by the marker instruction their epilogue emits, not by name
    found = re.findall(r"(\d+) kernels with the landing-zone marker, (\d+) compiler-generated uses", p.stdout)
    assert len(found) == 3 and all(int(n) >= 3 and int(bad) == 0 for n, bad in found), p.stdout


def test_groupnorm_act_fits_is_the_launchers_whole_contract():
    """mi355x_sd_groupnorm_act_fits == 1 must mean mi355x_sd_groupnorm_act runs (the planners choose the one-launch form on the
    predicate alone): wide groups (C / groups > 128: the kernel's per-group gamma / beta table) are refused by the predicate itself,
    in the built library and in the C-ABI emulator alike -- a norm_num_groups = 8 model at C = 1280 then plans statistics + apply."""
    from paddlemix_amd import _lib
    from tests.abi_emulator import Emulator
    lib, emu = _lib.load(), Emulator()
    for hw, c, groups, want in ((256, 1280, 32, 1), (256, 1280, 8, 0), (64, 2048, 8, 0), (64, 2048, 16, 1), (1 << 16, 320, 32, 0),
                                (256, 1290, 30, 0)):
        assert lib.mi355x_sd_groupnorm_act_fits(hw, c, groups) == want, (hw, c, groups)
        assert emu.mi355x_sd_groupnorm_act_fits(hw, c, groups) == want, (hw, c, groups)


def test_emulator_answers_the_planners_size_queries_like_the_library():
    """A model planned on the emulator can be exported and replayed on the device (tests/test_gpu_export.py): every size the planner
    asks the backend for must be the library's own answer. (Round 4: the emulator's GroupNorm workspace was 64 floats flat, the
    device kernel writes up to thousands -- a CPU-planned VAE program then failed one run in three on the device.)"""
    from paddlemix_amd import _lib
    from tests.abi_emulator import Emulator
    lib, emu = _lib.load(), Emulator()
    for B, HW, C in ((2, 1024, 32), (2, 256, 64), (2, 64, 64), (8, 16384, 128), (1, 4096, 320), (8, 1024, 1280), (3, 100, 2560), (1, 7, 8)):
        assert emu.mi355x_sd_groupnorm_workspace_floats(B, HW, C) == lib.mi355x_sd_groupnorm_workspace_floats(B, HW, C), (B, HW, C)


@pytest.mark.parametrize("src", ["tests/c/unet_exec_test.c", "tests/c/program_test.c", "scripts/c/step_bench.c", "scripts/c/gemm_probe.c",
                                 "scripts/c/conv_probe.c", "scripts/c/attn_probe.c", "tests/c/comm_test.c"])
def test_plain_c_clients_build_against_the_header_and_fail_loudly_without_a_device(src, tmp_path):
    """Every plain-C client of the ABI (the two GPU-test clients and the torch-free benches / probes) compiles against include/mi355x_sd.h
    as C11, links against the library, and -- on a machine without a GPU -- exits non-zero with the library's own message instead
    of computing anything anywhere else."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no gcc / ROCm headers")
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    exe = str(tmp_path / "client")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "scripts", "c"), os.path.join(ROOT, src), "-L" + libdir, "-lmi355x_sd", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    import torch
    if torch.cuda.is_available():
        return
    cfg = os.path.join(ROOT, "scripts", "c", "sd15_unet_config.json")
    args = {"tests/c/unet_exec_test.c": [cfg, "1", "8", "8", "7", str(tmp_path / "o.bin")], "tests/c/program_test.c": [str(tmp_path / "none.prog")],
            "scripts/c/step_bench.c": [cfg, "1", "8", "8", "7", "1", "0"], "scripts/c/gemm_probe.c": ["1"], "scripts/c/conv_probe.c": ["1"],
            "scripts/c/attn_probe.c": ["1"], "tests/c/comm_test.c": ["0", "1", str(tmp_path / "id.bin")]}[src]
    r = subprocess.run([exe] + args, capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH=libdir), timeout=120)
    assert r.returncode != 0, r.stdout
    assert r.stderr.strip(), "a failing client says why"


def test_c_bench_configs_are_the_bench_workloads():
    """scripts/c/*_unet_config.json (what scripts/c/step_bench.c plans) are the configurations bench.py's workloads name."""
    import json
    from tests.configs import SD15, SDXL
    for name, cfg in (("sdxl", SDXL), ("sd15", SD15)):
        got = json.load(open(os.path.join(ROOT, "scripts", "c", f"{name}_unet_config.json")))
        assert got == {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, name
