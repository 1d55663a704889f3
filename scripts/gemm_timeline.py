"""Where a GEMM launch's time goes, per block: entry -> first K-tile landed -> K loop done -> stores issued -> stores written.

The software-pipelined kernels (csrc/gemm_pipe.hip) stamp the 100-MHz wall clock at those points when MI355X_SD_GEMM_TSTAMP holds
the address of a device buffer (6 x u64 per block; diagnostics only, NULL in production). This script runs the shapes that
dominate the SDXL bs-8 step back to back (each launch preceded by a cache-disturbing launch of another shape, like inside the
step), copies the stamps back and prints, per shape: the launch span seen by HIP events, the spread of block start times (launch
ramp), and the median / p90 / max per-block prologue, K-loop and epilogue durations, plus the idle tail (last block's end - each
block's end).      python scripts/gemm_timeline.py [--iters 5]
"""
import argparse
import os

os.environ.setdefault("MI355X_SD_LIB", "dbg")   # the time-stamp hook is an A/B switch: debug-switch build only (csrc/common.h sd_switch)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()

MAXB = 1 << 16
ts = torch.zeros(MAXB * 6, dtype=torch.int64, device="cuda")
os.environ["MI355X_SD_GEMM_TSTAMP"] = hex(ts.data_ptr())   # (needs the debug-switch build: MI355X_SD_LIB=dbg, set at the top)   # read once, at the first GEMM launch

from paddlemix_amd import ops  # noqa: E402

ops.init(0)
torch.manual_seed(0)

SHAPES = [  # M, N, K, geglu, residual          (SDXL bs 8: launches per step)
    (8192, 1280, 1280, False, True),     # to_out / to_q: 192
    (8192, 3840, 1280, False, False),    # fused QKV: 60
    (8192, 10240, 1280, True, False),    # FF1 GEGLU: 60
    (8192, 1280, 5120, False, True),     # FF2: 60
    (32768, 640, 640, False, True),      # 640-level projections: 40
]


def q(x, f):
    k = max(0, min(len(x) - 1, int(f * (len(x) - 1))))
    return x[k]


for M, N, K, geglu, resid in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    n_out = N // 2 if geglu else N
    r = torch.randn(M, n_out, device="cuda").to(torch.bfloat16) if resid else None
    out = torch.empty(M, n_out, device="cuda", dtype=torch.bfloat16)
    # a different launch in between (evicts this shape's operands from L2 like the step's neighbouring kernels do)
    xa = torch.randn(8192, 2560, device="cuda").to(torch.bfloat16)
    xw = (torch.randn(1280, 2560, device="cuda") / 50).to(torch.bfloat16)
    xo = torch.empty(8192, 1280, device="cuda", dtype=torch.bfloat16)
    rows = []
    for it in range(args.iters + 1):
        ops.linear(xa, xw, None, out=xo)
        ts.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear(a, w, b, out=out, geglu=geglu, residual=r)
        e1.record()
        torch.cuda.synchronize()
        if it == 0:
            continue
        t = ts.view(-1, 6).cpu()
        t = t[t[:, 0] != 0]
        nb = t.shape[0]
        if nb == 0:   # this shape runs on a kernel without stamps (the phased 256x256 kernel, gemm256.hip)
            rows.append(dict(ev=e0.elapsed_time(e1) * 1e3, nb=0))
            continue
        T0 = t[:, 0].min().item()
        us = lambda c: sorted(((c - 0) * 0.01).tolist())   # noqa: E731  (100 MHz -> us)
        start = us(t[:, 0] - T0)
        pro, loop, epi, drain = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 4] - t[:, 3])
        end = (t[:, 4] - T0) * 0.01
        span = end.max().item()
        tail = sorted((span - end).tolist())
        xcc = ((t[:, 5] >> 32) & 15)
        rows.append(dict(ev=e0.elapsed_time(e1) * 1e3, nb=nb, span=span, start=start, pro=pro, loop=loop, epi=epi, drain=drain, tail=tail,
                         xcds=sorted(set(xcc.tolist()))))
    f = lambda v: f"{q(v, 0.5):6.2f} / {q(v, 0.9):6.2f} / {v[-1]:6.2f}"   # noqa: E731
    best = min(rows, key=lambda d: d["ev"])
    if best["nb"] == 0:
        print(f"gemm {M}x{N}x{K}: no stamps (not a gemm_pipe launch); event time min {best['ev']:.1f} us", flush=True)
        continue
    evs = " ".join("%.1f" % d_["ev"] for d_ in rows)
    print(f"gemm {M}x{N}x{K}{'g' if geglu else ''}{'+R' if resid else ''}: {best['nb']} blocks, event time min {best['ev']:.1f} us "
          f"(all: {evs}), {2.0 * M * N * K / best['ev'] / 1e6:.0f} TFLOP/s; stamps: first entry -> last store "
          f"written {best['span']:.2f} us      [median / p90 / max over blocks, us]")
    print(f"   block entry after the first block's  {f(best['start'])}")
    print(f"   prologue (entry -> first tile landed) {f(best['pro'])}")
    print(f"   K loop                                {f(best['loop'])}")
    print(f"   epilogue (operand loads, math, issue) {f(best['epi'])}")
    print(f"   store drain (issue -> written back)   {f(best['drain'])}")
    print(f"   idle until the launch's last store    {f(best['tail'])}")
    print(f"   event time - stamp span = {best['ev'] - best['span']:.2f} us (launch + completion signalling)", flush=True)
