"""Export a planned model step to a file that libmi355x_sd.so replays without Python (seam B2, include/mi355x_sd.h
``mi355x_sd_program_*``; runtime: csrc/program_exec.hip), and the ctypes wrapper around that runtime.

What the reference does for deployment is the same two steps: export the model once (``deploy/sd15/export_model.py:78-90``: named
inputs, static program + parameters on disk), then run it behind a predictor object that knows nothing about the model's Python
(``PaddleInferRuntimeModel``, models/paddleinfer_runtime.py:47-126). Here the "static program" is the planner's launch list:

    model = SD3Transformer2DModel(cfg, params)          # or UNet2DConditionModel, AutoencoderKL, CLIPTextModel, ...
    model(x, enc, pooled, t)                            # builds (and, for lazily completed plans, finishes) the plan
    export_program(model, model.plan_for(...), "sd3_b2_128.mi3prg")

File layout (little endian): magic "MI3SDPRG", u32 version, ABI version, element type, counts, u64 split-K workspace bytes; the
region table (kind weight | const | scratch | io, bytes, file offset of the initial contents or 0, name); the I/O table (region,
direction, dtype, shape, name); the launch list (entry-point name, arguments tagged int | float | pointer = (region, byte offset)
| null | stream); then the 64-byte aligned contents of the weight / const regions (and of small I/O regions).
"""
from __future__ import annotations

import ctypes
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import _lib

MAGIC, VERSION = b"MI3SDPRG", 1
A_INT, A_FLOAT, A_PTR, A_NULL, A_STREAM = range(5)
R_WEIGHT, R_CONST, R_SCRATCH, R_IO = range(4)
IO_F32, IO_ELEM16, IO_I32, IO_U8 = range(4)
_IO_DATA_LIMIT = 1 << 20     # initial contents of I/O regions travel only when small (they are overwritten by the caller anyway)


def _storage_span(t: torch.Tensor) -> Tuple[int, int]:
    s = t.untyped_storage()
    return s.data_ptr(), s.nbytes()


def _storage_bytes(t: torch.Tensor) -> bytes:
    s = t.untyped_storage()
    flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(s, 0, (s.nbytes(),))
    return flat.cpu().numpy().tobytes()


def _named_tensors(plan) -> List[Tuple[str, torch.Tensor]]:
    out = []
    for k, v in vars(plan).items():
        if k in ("keep", "consts", "prog", "emb_tensors"):
            continue
        if torch.is_tensor(v):
            out.append((k, v))
        elif isinstance(v, (list, tuple)) and v and all(torch.is_tensor(x) for x in v):
            out.extend((f"{k}.{i}", x) for i, x in enumerate(v))
    return out


def _io_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return IO_F32
    if t.dtype == _lib.elem_dtype():
        return IO_ELEM16
    if t.dtype == torch.int32:
        return IO_I32
    if t.dtype == torch.uint8:
        return IO_U8
    raise ValueError(f"I/O tensor of dtype {t.dtype}")


def _pack_str(s: str) -> bytes:
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def export_program(model, plan, path: str, outputs: Iterable[str] = ("out",)) -> dict:
    """Write `plan` of `model` (any DeviceProgram model) to `path`. `outputs`: names of plan attributes the caller reads back
    (everything else that is named is an input); a list attribute `hidden` is addressed as "hidden" (all) or "hidden.3".
    Returns a summary dict (regions, launches, bytes).
    Contents: weights and plan-time constants travel with the file; a named I/O region's plan-time contents travel only up to 1 MiB
    (_IO_DATA_LIMIT) -- a larger INPUT is expected to be written by the caller before every run (all of this library's planners do),
    it replays as whatever the bound buffer holds otherwise."""
    if getattr(plan, "text_dim", 0) is None:
        raise ValueError("this plan is completed at the first forward (text_time widths): run the model once before exporting")
    outputs = tuple(outputs)
    regions: List[dict] = []
    by_ptr: Dict[int, int] = {}

    def add_region(t: torch.Tensor, kind: int, name: str) -> int:
        ptr, nbytes = _storage_span(t)
        if ptr not in by_ptr:                 # first registration wins: weights, then constants, then named I/O, then scratch
            by_ptr[ptr] = len(regions)
            regions.append(dict(kind=kind, ptr=ptr, bytes=nbytes, name=name, tensor=t))
        return by_ptr[ptr]

    for k, t in model.w.items():
        if torch.is_tensor(t):
            add_region(t, R_WEIGHT, "w:" + k)
    for i, t in enumerate(getattr(plan, "consts", [])):    # filled at plan time, read by every run
        add_region(t, R_CONST, f"const{i}")
    named = _named_tensors(plan)
    ios = []
    for name, t in named:
        # an I/O entry IS its region (device address = region base, size = region bytes): a named tensor that is a view into a larger
        # storage, or that shares its storage with a weight / constant / another named tensor registered before it, would be
        # reported with the wrong address, size or kind -- refused here instead of exported wrong
        ptr, nbytes = _storage_span(t)
        if (t.storage_offset() != 0 or not t.is_contiguous() or nbytes != t.numel() * t.element_size() or ptr != t.data_ptr()):
            raise ValueError(f"I/O tensor '{name}' must own its storage (contiguous, offset 0): shape {tuple(t.shape)}, "
                             f"storage offset {t.storage_offset()}, storage bytes {nbytes}")
        if ptr in by_ptr and regions[by_ptr[ptr]]["kind"] != R_IO:   # (two names of one I/O tensor -- the VAE's mean / out -- are fine)
            other = regions[by_ptr[ptr]]
            raise ValueError(f"I/O tensor '{name}' shares its storage with '{other['name']}' (exported earlier as kind {other['kind']})")
        idx = add_region(t, R_IO, name)
        is_out = any(name == o or name.startswith(o + ".") for o in outputs)
        ios.append(dict(region=idx, is_output=int(is_out), dtype=_io_dtype(t), shape=list(t.shape), name=name))
    if not any(io["is_output"] for io in ios):
        raise ValueError(f"none of the outputs {outputs} is a tensor attribute of the plan ({[n for n, _ in named]})")
    for i, t in enumerate(plan.keep):
        add_region(t, R_SCRATCH, f"buf{i}")
    ws = getattr(model, "_workspace", None)
    if torch.is_tensor(ws):   # the model's split-K / widening scratch: an argument of its GEMM-class launches (ABI 12)
        add_region(ws, R_SCRATCH, "gemm_workspace")
    spans = sorted((r["ptr"], r["ptr"] + r["bytes"], i) for i, r in enumerate(regions))

    def locate(v: int, what: str) -> Tuple[int, int]:
        lo, hi = 0, len(spans)
        while lo < hi:
            mid = (lo + hi) // 2
            if spans[mid][0] <= v:
                lo = mid + 1
            else:
                hi = mid
        if lo and spans[lo - 1][0] <= v <= spans[lo - 1][1]:
            return spans[lo - 1][2], v - spans[lo - 1][0]
        raise ValueError(f"{what}: pointer {v:#x} lies in no weight / plan tensor of this model")

    ops = []
    for n, (fn, args, _kind, _fl) in enumerate(plan.prog):
        sym = getattr(fn, "__name__", None) or getattr(fn, "name", None)
        if sym not in _lib.SIGNATURES:
            raise ValueError(f"launch {n}: {fn!r} is not a C entry point of include/mi355x_sd.h")
        argtypes = _lib.SIGNATURES[sym][1]
        if len(argtypes) != len(args):
            raise ValueError(f"launch {n} ({sym}): {len(args)} arguments for {len(argtypes)} parameters")
        packed = []
        for j, (ty, v) in enumerate(zip(argtypes, args)):
            v = getattr(v, "value", v)
            if ty is ctypes.c_void_p:
                if j == len(args) - 1:
                    packed.append((A_STREAM, 0, 0))          # every launch takes its stream last
                elif v is None or v == 0:
                    packed.append((A_NULL, 0, 0))
                else:
                    r, off = locate(int(v), f"launch {n} ({sym}), argument {j}")
                    packed.append((A_PTR, r, off))
            elif ty is ctypes.c_float:
                packed.append((A_FLOAT, struct.unpack("<Q", struct.pack("<d", float(v)))[0], 0))
            elif ty in (ctypes.c_int, ctypes.c_int64, ctypes.c_size_t):
                packed.append((A_INT, int(v) & 0xFFFFFFFFFFFFFFFF, 0))
            else:
                raise ValueError(f"{sym}: parameter type {ty} has no tag")
        ops.append((sym, packed))

    workspace_bytes = 0   # (reserved: until ABI 11 the size of a process-wide split-K binding; the scratch is a region now)

    def wants_data(r) -> bool:
        return r["kind"] in (R_WEIGHT, R_CONST) or (r["kind"] == R_IO and r["bytes"] <= _IO_DATA_LIMIT)

    def tables(offsets: Optional[List[int]]) -> bytes:
        b = MAGIC + struct.pack("<IIIIIIQ", VERSION, _lib.ABI_VERSION, _lib._BUILDS[_lib.ELEM_NAME][1], len(regions), len(ops), len(ios),
                                workspace_bytes)
        for i, r in enumerate(regions):
            b += struct.pack("<IQQ", r["kind"], r["bytes"], offsets[i] if offsets else 0) + _pack_str(r["name"])
        for io in ios:
            shape = (io["shape"] + [0, 0, 0, 0])[:4]
            if len(io["shape"]) > 4:
                raise ValueError(f"I/O tensor {io['name']} has more than 4 dimensions")
            b += struct.pack("<IIII4q", io["region"], io["is_output"], io["dtype"], len(io["shape"]), *shape) + _pack_str(io["name"])
        for sym, packed in ops:
            b += _pack_str(sym) + struct.pack("<I", len(packed))
            for tag, u, v in packed:
                b += struct.pack("<IQQ", tag, u, v)
        return b

    head = len(tables(None))
    offsets, pos = [], (head + 63) // 64 * 64
    for r in regions:
        if wants_data(r) and r["bytes"]:
            offsets.append(pos)
            pos = (pos + r["bytes"] + 63) // 64 * 64
        else:
            offsets.append(0)
    if torch.cuda.is_available() and any(r["tensor"].is_cuda for r in regions):
        torch.cuda.synchronize()
    with open(path, "wb") as fh:
        fh.write(tables(offsets))
        for r, off in zip(regions, offsets):
            if off:
                fh.seek(off)
                fh.write(_storage_bytes(r["tensor"]))
        fh.truncate(max(pos, fh.tell()))
    kinds = {R_WEIGHT: "weight", R_CONST: "const", R_SCRATCH: "scratch", R_IO: "io"}
    return dict(launches=len(ops), regions=len(regions), file_bytes=pos,
                device_bytes={kinds[k]: sum(r["bytes"] for r in regions if r["kind"] == k) for k in kinds},
                inputs=[io["name"] for io in ios if not io["is_output"]], outputs=[io["name"] for io in ios if io["is_output"]])


_TORCH_OF_IO = {IO_F32: torch.float32, IO_I32: torch.int32, IO_U8: torch.uint8}


class ExportedProgram:
    """ctypes wrapper of ``mi355x_sd_program_*``: what a C host does, with torch only as the owner of the device buffer.

    ``info`` works without a GPU (load parses and type-checks on the host); ``bind`` / ``run`` need one."""

    def __init__(self, path: str):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.mi355x_sd_program_load(path.encode(), ctypes.byref(self._h)))
        self._buf = None
        self.tensors: Dict[str, torch.Tensor] = {}

    def close(self) -> None:
        if self._h:
            self._lib.mi355x_sd_program_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_launches(self) -> int:
        return self._lib.mi355x_sd_program_num_launches(self._h)

    def device_bytes(self) -> int:
        n = ctypes.c_size_t()
        _lib.check(self._lib.mi355x_sd_program_device_bytes(self._h, ctypes.byref(n)))
        return n.value

    def info(self) -> List[dict]:
        out = []
        for i in range(self._lib.mi355x_sd_program_num_io(self._h)):
            name, is_out, dt, nd = ctypes.c_char_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            shape, nbytes, ptr = (ctypes.c_int64 * 4)(), ctypes.c_size_t(), ctypes.c_void_p()
            _lib.check(self._lib.mi355x_sd_program_io_info(self._h, i, ctypes.byref(name), ctypes.byref(is_out), ctypes.byref(dt), shape,
                                                           ctypes.byref(nd), ctypes.byref(nbytes), ctypes.byref(ptr)))
            out.append(dict(name=name.value.decode(), is_output=bool(is_out.value), dtype=dt.value, shape=list(shape[:nd.value]),
                            bytes=nbytes.value, device_ptr=ptr.value))
        return out

    def bind(self, device="cuda", use_graph: bool = False) -> "ExportedProgram":
        dev = torch.device(device)
        self._buf = torch.empty(self.device_bytes() + 256, dtype=torch.uint8, device=dev)
        pad = (-self._buf.data_ptr()) % 256
        base = self._buf[pad:]
        self._stream = torch.cuda.Stream(device=dev)
        _lib.check(self._lib.mi355x_sd_program_set_option(self._h, b"use_graph", int(use_graph)))
        _lib.check(self._lib.mi355x_sd_program_bind(self._h, base.data_ptr(), base.numel(), self._stream.cuda_stream))
        for io in self.info():
            off = io["device_ptr"] - base.data_ptr()
            dt = _TORCH_OF_IO.get(io["dtype"], _lib.elem_dtype())
            n = 1
            for s in io["shape"]:
                n *= s
            self.tensors[io["name"]] = base[off: off + n * torch.empty(0, dtype=dt).element_size()].view(dt).reshape(io["shape"])
        self._outputs = [io["name"] for io in self.info() if io["is_output"]]
        return self

    def run(self, **inputs) -> Dict[str, torch.Tensor]:
        cur = torch.cuda.current_stream(self._buf.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            for k, v in inputs.items():
                dst = self.tensors[k]
                dst.copy_(v.reshape(dst.shape).to(dst.dtype), non_blocking=True)
            _lib.check(self._lib.mi355x_sd_program_run(self._h, self._stream.cuda_stream))
            out = {k: self.tensors[k].clone() for k in self._outputs}
        cur.wait_stream(self._stream)
        return out
