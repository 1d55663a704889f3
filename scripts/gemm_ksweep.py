"""K-sweep of one GEMM output shape: kernel time = a + b*K separates the per-tile fixed cost (launch ramp, prologue,
epilogue, output write) from the main-loop slope. Run under rocprofv3 by scripts/prof_ksweep.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops  # noqa: E402

ops.init(0)
M, N = (int(v) for v in os.environ.get("KSWEEP_MN", "8192x3840").split("x"))
for K in (64, 128, 256, 512, 1024, 1280, 2560, 5120):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.linear(a, w, out=out)
torch.cuda.synchronize()
