"""Time per launch of the GEMM / conv shapes that dominate the SDXL bs-8 step, under the MI355X_SD_GEMM_* switches of the environment
(read once per process: run once per variant, scripts/gpu_r03_s3.sh). Four operand sets per shape are rotated so that no launch finds
its own operands hot in L2; time = HIP events around `reps` launches / reps, best of `rounds`.

    python scripts/gemm_variants.py [--label NAME] [--rounds 3] [--reps 8]        prints one line per shape + a JSON line
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--label", default="default")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--only", default="")
args = ap.parse_args()
ops.init(0)
torch.manual_seed(0)

#        name, M, N, K, geglu, residual, bias, launches per SDXL bs-8 step
GEMMS = [("to_out", 8192, 1280, 1280, False, True, True, 192),
         ("qkv", 8192, 3840, 1280, False, False, False, 60),
         ("ff1", 8192, 10240, 1280, True, False, True, 60),
         ("ff2", 8192, 1280, 5120, False, True, True, 60),
         ("out640", 32768, 640, 640, False, True, True, 40),
         ("ff1_640", 32768, 5120, 640, True, False, True, 10),
         ("ff2_640", 32768, 640, 2560, False, True, True, 10),
         ("qkv640", 32768, 1920, 640, False, False, False, 10)]
#        name, B, H, W, Cin, Cout, temb, residual, launches
CONVS = [("c1280", 8, 32, 32, 1280, 1280, True, False, 10),
         ("c320", 8, 128, 128, 320, 320, False, True, 7),
         ("c640", 8, 64, 64, 640, 640, True, False, 6)]
if os.environ.get("GEMM_VARIANTS_ALL"):   # every distinct shape of the SDXL step (+ the SD3-medium block GEMMs): the tile sweep of round 3
    GEMMS += [("g1280x2560", 8192, 1280, 2560, False, False, True, 2), ("g320x640", 131072, 320, 640, False, False, True, 2),
              ("g640x1920", 32768, 640, 1920, False, False, True, 1), ("g320x960", 131072, 320, 960, False, False, True, 1),
              ("sd3_qkv", 32768, 4608, 1536, False, False, True, 24), ("sd3_out", 32768, 1536, 1536, False, True, True, 24),
              ("sd3_ff1", 32768, 6144, 1536, False, False, True, 24), ("sd3_ff2", 32768, 1536, 6144, False, True, True, 24)]
    CONVS += [("c320x8640", 8, 128, 128, 960, 320, True, False, 1), ("c640x11520", 8, 64, 64, 1280, 640, True, False, 1),
              ("c1280x23040", 8, 32, 32, 2560, 1280, True, False, 2), ("c640x17280", 8, 64, 64, 1920, 640, True, False, 1)]
NSET = 4
out = {}


def timeit(fns):
    best = 1e30
    for _ in range(args.rounds):
        for f in fns:
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.reps):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / args.reps * 1e3)
    return best


for name, M, N, K, geglu, resid, bias, n in GEMMS:
    if args.only and name not in args.only.split(","):
        continue
    n_out = N // 2 if geglu else N
    sets = []
    for _ in range(NSET):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda") if bias else None
        r = torch.randn(M, n_out, device="cuda").to(torch.bfloat16) if resid else None
        o = torch.empty(M, n_out, device="cuda", dtype=torch.bfloat16)
        sets.append((a, w, b, r, o))
    us = timeit([(lambda s=s: ops.linear(s[0], s[1], s[2], out=s[4], geglu=geglu, residual=s[3])) for s in sets])
    tf = 2.0 * M * N * K / us / 1e6
    out[name] = dict(us=round(us, 2), tf=round(tf, 1), per_step_ms=round(us * n / 1e3, 3))
    print(f"{args.label:24s} gemm {name:8s} {M}x{N}x{K}{'g' if geglu else ''}{'+R' if resid else ''}: {us:8.1f} us {tf:7.0f} TF   x{n} = {us * n / 1e3:6.2f} ms",
          flush=True)
    del sets
for name, B, H, W, Cin, Cout, temb, resid, n in CONVS:
    if args.only and name not in args.only.split(","):
        continue
    sets = []
    for _ in range(NSET):
        x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Cout, 9 * Cin, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16)
        b = torch.randn(Cout, device="cuda")
        rb = torch.randn(B, Cout, device="cuda") if temb else None
        r = torch.randn(B * H * W, Cout, device="cuda").to(torch.bfloat16) if resid else None
        o = torch.empty(B * H * W, Cout, device="cuda", dtype=torch.bfloat16)
        sets.append((x, w, b, rb, r, o))
    us = timeit([(lambda s=s: ops.conv3x3(s[0], s[1], s[2], rowbias=s[3], residual=s[4], out=s[5])) for s in sets])
    tf = 2.0 * B * H * W * Cout * 9 * Cin / us / 1e6
    out[name] = dict(us=round(us, 2), tf=round(tf, 1), per_step_ms=round(us * n / 1e3, 3))
    print(f"{args.label:24s} conv {name:8s} {B}x{H}x{W}x{Cin}->{Cout}: {us:8.1f} us {tf:7.0f} TF   x{n} = {us * n / 1e3:6.2f} ms", flush=True)
    del sets
tot = sum(v["per_step_ms"] for v in out.values())
print(f"{args.label:24s} sum over these shapes: {tot:.2f} ms per step")
print("VARIANT_TIMES " + json.dumps(dict(label=args.label, env={k: v for k, v in os.environ.items() if k.startswith('MI355X_SD_')}, shapes=out, sum_ms=round(tot, 3))))
