#!/usr/bin/env python
"""Per-kernel micro-benchmark at the SDXL bs-8 shapes (HIP-event timing, random data)."""
import math
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*s):
    return torch.randn(*s, device="cuda").to(torch.bfloat16)


def main():
    ops.init(0)
    print("== linear (M,N,K) ==")
    for M, N, K, geglu in [(32768, 1920, 640, 0), (32768, 640, 640, 0), (32768, 5120, 640, 1), (32768, 640, 2560, 0),
                           (8192, 3840, 1280, 0), (8192, 1280, 1280, 0), (8192, 10240, 1280, 1), (8192, 1280, 5120, 0),
                           (616, 2560, 2048, 0), (4096, 4096, 4096, 0), (8192, 8192, 8192, 0)]:
        a, w = rnd(M, K), rnd(N, K)
        ms = timeit(lambda: ops.linear(a, w, geglu=bool(geglu)))
        print(f"linear {M:6d} {N:6d} {K:6d} geglu={geglu}: {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s")
    print("== conv3x3 (B,H,W,Cin,Cout) ==")
    for B, H, W, Cin, Cout in [(8, 128, 128, 320, 320), (8, 64, 64, 640, 640), (8, 32, 32, 1280, 1280),
                               (8, 32, 32, 2560, 1280), (8, 64, 64, 1920, 640), (8, 128, 128, 960, 320)]:
        x, w = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin)
        ms = timeit(lambda: ops.conv3x3(x, w))
        print(f"conv {B} {H}x{W} {Cin}->{Cout}: {ms:8.3f} ms  {2.0 * B * H * W * Cout * 9 * Cin / ms / 1e9:8.1f} TFLOP/s")
    print("== sdpa (B,H,Sq,Skv,D) ==")
    for B, H, Sq, Skv, D in [(8, 10, 4096, 4096, 64), (8, 20, 1024, 1024, 64), (8, 10, 4096, 77, 64),
                             (8, 20, 1024, 77, 64), (1, 8, 4096, 4096, 40), (16, 64, 2048, 2048, 64)]:
        q, k, v = rnd(B, Sq, H, D), rnd(B, Skv, H, D), rnd(B, Skv, H, D)
        ms = timeit(lambda: ops.sdpa(q, k, v))
        print(f"sdpa {B} {H} {Sq} {Skv} {D}: {ms:8.3f} ms  {4.0 * B * H * Sq * Skv * D / ms / 1e9:8.1f} TFLOP/s")
    print("== norms ==")
    for B, HW, C in [(8, 16384, 320), (8, 16384, 960), (8, 4096, 640), (8, 1024, 1280), (8, 1024, 2560)]:
        x = rnd(B, HW, C)
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        ms1 = timeit(lambda: ops.groupnorm_scale_shift(x, g, b, 32, 1e-5))
        ss = ops.groupnorm_scale_shift(x, g, b, 32, 1e-5)
        y = torch.empty_like(x)
        ms2 = timeit(lambda: ops.scale_shift_act(x, ss, True, out=y))
        nb = B * HW * C * 2
        print(f"groupnorm {B} {HW} {C}: stats {ms1:7.3f} ms ({nb / ms1 / 1e9:6.2f} TB/s)  apply {ms2:7.3f} ms ({2 * nb / ms2 / 1e9:6.2f} TB/s)")
    for rows, C in [(32768, 640), (8192, 1280)]:
        x = rnd(rows, C)
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        y = torch.empty_like(x)
        ms = timeit(lambda: ops.layer_norm(x, g, b, out=y))
        print(f"layernorm {rows} {C}: {ms:7.3f} ms ({2 * rows * C * 2 / ms / 1e9:6.2f} TB/s)")


if __name__ == "__main__":
    main()
