"""Shared host machinery of the MI355X models: symbolic addresses, row views, and the static program of C-ABI
launches (eager run with optional per-launch HIP-event timing, hipGraph capture / replay on the model's stream)."""
from __future__ import annotations

import contextvars

import ctypes
from typing import Dict

import torch

from . import _lib

WORKSPACE_BYTES = 32 << 20   # split-K partial sums (the planner shrinks the split count to fit) AND the just-in-time widened copy of a
                             # weight-only-fp8 matrix (2 * N * K bytes, <= ~19 MB for SD3-medium): the `ws, ws_bytes` arguments of the
                             # GEMM-class entry points (include/mi355x_sd.h, ABI 12)


class _Ref:
    """Symbolic device address inside a named scratch buffer (resolved after all sizes are known)."""
    __slots__ = ("buf", "off")

    def __init__(self, buf: str, off: int = 0):
        self.buf, self.off = buf, off

    def __add__(self, nbytes: int) -> "_Ref":
        return _Ref(self.buf, self.off + nbytes)


class _V:
    """Row view: `rows` rows of `C` channels of `es` bytes each (2 = the build's 16-bit element, 4 = fp32: the residual
    stream of the fp32-residual mode), row stride `ld` elements, at address `p` (int or _Ref)."""
    __slots__ = ("p", "rows", "C", "ld", "es")

    def __init__(self, p, rows, C, ld=None, es=2):
        self.p, self.rows, self.C, self.ld, self.es = p, rows, C, (C if ld is None else ld), es

    def cols(self, off: int, C: int) -> "_V":
        return _V(self.p + self.es * off, self.rows, C, self.ld, self.es)


class _Plan:
    pass


# Test plumbing, ONE place: while a backend object is pushed here (tests/abi_emulator.py ``on_emulator`` does that around a
# constructor call), models built meanwhile send their C-ABI calls to it instead of the HIP library -- the host logic of every
# model then runs on a machine without a GPU. Product code never pushes anything: the library is the only backend it selects, and
# a missing library or GPU is an error, not a fallback.
# A context variable, not a module global: the override is seen by the thread / task that set it and by nobody else (a model built
# meanwhile on another thread gets the HIP library), and it must be an object that SAYS it is a test backend.
_BACKEND_OVERRIDE: contextvars.ContextVar = contextvars.ContextVar("mi355x_sd_test_backend", default=None)


class DeviceProgram:
    """Backend selection + execution of a plan (``plan.prog``: list of (cfunc, args, kind, flops))."""

    def _init_backend(self, device, use_graph: bool, profile: bool):
        _test_backend = _BACKEND_OVERRIDE.get()
        if _test_backend is not None and not getattr(_test_backend, "IS_TEST_BACKEND", False):
            raise TypeError("paddlemix_amd.program._BACKEND_OVERRIDE holds something that is not a test backend")
        self._emulated = _test_backend is not None
        if self._emulated:
            self._lib = _test_backend
            self.device = torch.device("cpu")
            self._stream = None
            self._stream_ptr = 0
            self._gemm_ws = (None, 0)
            use_graph = False
        else:
            self._lib = _lib.load()  # hard failure if the HIP library is not built
            if not torch.cuda.is_available():
                raise _lib.MI355XError(f"{type(self).__name__}(mi355x) needs a GPU; there is no CPU fallback")
            self.device = torch.device(device)
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
            _lib.check(self._lib.mi355x_sd_init(self.device.index))
            self._stream = torch.cuda.Stream(device=self.device)
            self._stream_ptr = self._stream.cuda_stream
            # split-K / widening scratch of this model's GEMM-class launches: owned here, an ARGUMENT of every such call the planners
            # emit (`*self._gemm_ws` in front of the stream; ABI 12 -- nothing process-wide is bound), baked into the graphs
            self._workspace = torch.empty(WORKSPACE_BYTES, device=self.device, dtype=torch.uint8)
            self._gemm_ws = (self._workspace.data_ptr(), self._workspace.numel())
        self.dtype = _lib.elem_dtype()
        self.use_graph = use_graph
        self.profile = profile
        self._plans: Dict[tuple, _Plan] = {}
        self.w: Dict[str, torch.Tensor] = {}
        self.kernel_times: Dict[str, list] = {}

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.w.values())

    def _run_eager(self, plan: _Plan) -> None:
        if not self.profile or self._emulated:
            for fn, args, _, _ in plan.prog:
                rc = fn(*args)
                if rc:
                    _lib.check(rc)
            return
        evs = []
        for fn, args, kind, fl in plan.prog:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self._stream)
            rc = fn(*args)
            e1.record(self._stream)
            if rc:
                _lib.check(rc)
            evs.append((kind, fl, e0, e1))
        self._stream.synchronize()
        for kind, fl, e0, e1 in evs:
            self.kernel_times.setdefault(kind, []).append((e0.elapsed_time(e1) * 1e-3, fl))

    def _capture(self, plan: _Plan) -> None:
        lib = self._lib
        sp = self._stream_ptr
        _lib.check(lib.mi355x_sd_graph_begin(sp))
        try:
            for fn, args, _, _ in plan.prog:
                rc = fn(*args)
                if rc:
                    _lib.check(rc)
        finally:
            exe = ctypes.c_void_p()
            rc = lib.mi355x_sd_graph_end(sp, ctypes.byref(exe))
        _lib.check(rc)
        plan.graph = exe

    def run(self, plan: _Plan) -> torch.Tensor:
        """Launch the step on the model's stream (inputs already staged); returns the static output buffer."""
        if self.use_graph and not self.profile:
            if plan.graph is None:
                self._run_eager(plan)  # warm-up outside capture (lazy module loading)
                self._capture(plan)
            _lib.check(self._lib.mi355x_sd_graph_launch(plan.graph, self._stream_ptr))
        else:
            self._run_eager(plan)
        return plan.out
