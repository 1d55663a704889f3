"""One GEMM / conv shape: parity against fp32 torch and time per launch (hip events; rocprofv3 --pmc via scripts/pmc.sh).
env GEMM_SHAPE=MxNxK [GEGLU=1] [CONV=BxHxWxCinxCout]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops  # noqa: E402

ops.init(0)
torch.manual_seed(0)


def timeit(fn, n=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if os.environ.get("CONV"):
    B, H, W, Cin, Cout = (int(v) for v in os.environ["CONV"].split("x"))
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    b = torch.randn(Cout, device="cuda")
    out = ops.conv3x3(x, wp, b)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    us = timeit(lambda: ops.conv3x3(x, wp, b))
    print(f"conv {os.environ['CONV']}: rel {((out.float() - ref).norm() / ref.norm()).item():.3e}  {us:.1f} us  "
          f"{2.0 * B * H * W * Cout * 9 * Cin / us / 1e6:.0f} TF")
else:
    M, N, K = (int(v) for v in os.environ.get("GEMM_SHAPE", "8192x10240x1280").split("x"))
    geglu = os.environ.get("GEGLU") == "1"
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=torch.bfloat16)
    ops.linear(a, w, b, out=out, geglu=geglu)
    rows = slice(0, 2048)
    y = a[rows].float() @ w.float().t() + b
    if geglu:   # rows of W interleaved [16 value | 16 gate] (include/mi355x_sd.h MI355X_SD_GEGLU)
        y = y.reshape(-1, N // 32, 2, 16)
        y = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(-1, N // 2)
    us = timeit(lambda: ops.linear(a, w, b, out=out, geglu=geglu))
    print(f"gemm {M}x{N}x{K}{'g' if geglu else ''}: rel {((out[rows].float() - y).norm() / y.norm()).item():.3e}  {us:.1f} us  "
          f"{2.0 * M * N * K / us / 1e6:.0f} TF")
