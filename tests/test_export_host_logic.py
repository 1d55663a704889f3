"""CPU tests of the exported-program seam (include/mi355x_sd.h mi355x_sd_program_*): every model family is planned on the
host-memory emulator, exported with paddlemix_amd.export, and loaded by the C runtime of libmi355x_sd.so -- which parses the file
and type-checks every launch of it against the real entry points (parameter count; int / float / pointer per parameter; pointers
inside their regions). bind / run need a GPU: tests/test_gpu_export.py."""
import struct

import pytest
import torch

from paddlemix_amd import _lib
from paddlemix_amd.export import ExportedProgram, export_program
from tests import export_cases as EC
from tests.abi_emulator import Emulator, on_emulator


def _export(name, tmp_path):
    model, run, outputs = on_emulator(EC.build, name, True)
    run()
    plan = EC.last_plan(model)
    path = str(tmp_path / (name + ".mi3prg"))
    return model, plan, outputs, path, export_program(model, plan, path, outputs)


@pytest.mark.parametrize("name", EC.NAMES)
def test_every_family_exports_and_type_checks_in_the_c_runtime(name, tmp_path):
    model, plan, outputs, path, summary = _export(name, tmp_path)
    assert summary["launches"] == len(plan.prog) > 0 and summary["outputs"]
    prog = ExportedProgram(path)            # mi355x_sd_program_load: would refuse a launch that does not fit its entry point
    assert prog.num_launches == len(plan.prog)
    info = {io["name"]: io for io in prog.info()}
    assert sorted(k for k, io in info.items() if io["is_output"]) == sorted(summary["outputs"])
    for io in info.values():
        t = dict(EC_named(plan))[io["name"]]
        assert io["shape"] == list(t.shape) and io["bytes"] == t.untyped_storage().nbytes() and io["device_ptr"] is None   # not bound
    # device memory the caller must provide = every region, 256-byte aligned, + the split-K workspace (none on the emulator)
    assert prog.device_bytes() >= sum(summary["device_bytes"].values())
    # weights travel with the file, scratch does not
    assert summary["file_bytes"] < summary["device_bytes"]["weight"] + summary["device_bytes"]["const"] + summary["device_bytes"]["io"] + (1 << 20)
    prog.close()


def EC_named(plan):
    from paddlemix_amd.export import _named_tensors
    return _named_tensors(plan)


def test_plan_constants_travel_with_the_file(tmp_path):
    """buffers a planner fills at plan time (SD3's cropped position table, T5's relative-position bias, CLIP's causal mask ...)
    are `const` regions whose contents the file carries; scratch is size only"""
    for name in ("sd3_mini", "t5_encoder", "clip_text", "clip_vision", "dit_mini"):
        model, plan, outputs, path, summary = _export(name, tmp_path)
        assert plan.consts and summary["device_bytes"]["const"] == sum(t.untyped_storage().nbytes() for t in plan.consts)
        raw = open(path, "rb").read()
        for t in plan.consts:
            assert t.contiguous().view(torch.uint8).numpy().tobytes()[:64] in raw


def test_loader_refuses_what_it_should(tmp_path):
    model, plan, outputs, path, _ = _export("unet_tiny", tmp_path)
    raw = bytearray(open(path, "rb").read())

    def load(mut, why):
        p = str(tmp_path / "bad.mi3prg")
        open(p, "wb").write(bytes(mut))
        with pytest.raises(_lib.MI355XError, match=why):
            ExportedProgram(p)

    bad = bytearray(raw)
    bad[:8] = b"NOTAPROG"
    load(bad, "bad magic")
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 12, _lib.ABI_VERSION + 1)        # exported for another ABI
    load(bad, "ABI")
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 16, 1 - _lib._BUILDS[_lib.ELEM_NAME][1])   # the other 16-bit element build
    load(bad, "element type")
    load(raw[:2000], "truncated|implausible|bad")
    # a launch whose first argument (a pointer parameter) arrives tagged as an integer
    sym = b"mi355x_sd_"
    at = raw.index(sym, 36)                                     # first launch record: u32 len, name, u32 nargs, then (u32 tag, u64, u64)...
    name_len = struct.unpack_from("<I", raw, at - 4)[0]
    first_arg = at + name_len + 4
    bad = bytearray(raw)
    tag = struct.unpack_from("<I", bad, first_arg)[0]
    struct.pack_into("<I", bad, first_arg, 1 if tag != 1 else 0)
    load(bad, "tag does not fit")
    with pytest.raises(_lib.MI355XError, match="cannot open"):
        ExportedProgram(str(tmp_path / "missing.mi3prg"))


def test_export_refuses_a_plan_that_is_not_complete(tmp_path):
    from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params
    from tests.configs import MINI_XL
    m = on_emulator(UNet2DConditionModel, MINI_XL, synth_unet_params(MINI_XL, seed=1))
    plan = m._get_plan(1, 16, 16, 7)          # text_time widths are only known at the first forward
    with pytest.raises(ValueError, match="run the model once"):
        export_program(m, plan, str(tmp_path / "x.mi3prg"))


def _replay_on_host(path, inputs):
    """interpret a program FILE on host memory: regions become byte buffers (weights / constants from the file), every launch goes to
    the emulator's function of the same name with its pointers resolved -- nothing of the exporting model is used"""
    import numpy as np
    raw = open(path, "rb").read()
    assert raw[:8] == b"MI3SDPRG"
    _, _, _, n_regions, n_ops, n_io, _ = struct.unpack_from("<IIIIIIQ", raw, 8)
    pos = 8 + struct.calcsize("<IIIIIIQ")
    regions = []
    for _ in range(n_regions):
        kind, nbytes, off = struct.unpack_from("<IQQ", raw, pos)
        pos += 20
        (n,) = struct.unpack_from("<I", raw, pos)
        pos += 4 + n
        buf = np.zeros(max(nbytes, 1) + 64, dtype=np.uint8)
        if off:
            buf[:nbytes] = np.frombuffer(raw, dtype=np.uint8, count=nbytes, offset=off)
        regions.append(buf)
    ios = []
    for _ in range(n_io):
        region, is_out, dtype, ndim, *shape = struct.unpack_from("<IIII4q", raw, pos)
        pos += struct.calcsize("<IIII4q")
        (n,) = struct.unpack_from("<I", raw, pos)
        name = raw[pos + 4: pos + 4 + n].decode()
        pos += 4 + n
        ios.append((name, region, is_out, dtype, shape[:ndim]))
    tdt = {0: torch.float32, 1: _lib.elem_dtype(), 2: torch.int32, 3: torch.uint8}

    def view(region, dtype, shape):
        n = 1
        for s_ in shape:
            n *= s_
        t = torch.from_numpy(regions[region])[: n * torch.empty(0, dtype=tdt[dtype]).element_size()].view(tdt[dtype])
        return t.reshape(shape)

    for name, region, is_out, dtype, shape in ios:
        if not is_out and name in inputs:
            view(region, dtype, shape).copy_(inputs[name].reshape(shape))
    emu = Emulator()
    for _ in range(n_ops):
        (n,) = struct.unpack_from("<I", raw, pos)
        sym = raw[pos + 4: pos + 4 + n].decode()
        pos += 4 + n
        (nargs,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        args = []
        for _ in range(nargs):
            tag, u, v = struct.unpack_from("<IQQ", raw, pos)
            pos += 20
            if tag == 0:
                args.append(u - (1 << 64) if u >= (1 << 63) else u)
            elif tag == 1:
                args.append(struct.unpack("<d", struct.pack("<Q", u))[0])
            elif tag == 2:
                args.append(regions[u].ctypes.data + v)
            elif tag == 3:
                args.append(None)
            else:
                args.append(0)
        rc = getattr(emu, sym)(*args)
        assert not rc, (sym, rc)
    return {name: view(region, dtype, shape).clone() for name, region, is_out, dtype, shape in ios if is_out}


@pytest.mark.parametrize("name", ["unet_mini_xl", "unet_tiny_masked_controlnet", "unet_ip_adapter", "sd3_mini", "vae_encode", "clip_vision", "t5_encoder"])
def test_program_file_is_self_contained(name, tmp_path):
    from paddlemix_amd.export import _named_tensors
    model, plan, outputs, path, _ = _export(name, tmp_path)
    named = dict(_named_tensors(plan))
    is_out = lambda n: any(n == o or n.startswith(o + ".") for o in outputs)  # noqa: E731
    want = {n: t.clone() for n, t in named.items() if is_out(n)}
    inputs = {n: t.clone() for n, t in named.items() if not is_out(n)}
    del model, plan
    got = _replay_on_host(path, inputs)
    assert sorted(got) == sorted(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
