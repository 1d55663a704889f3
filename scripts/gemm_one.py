"""One GEMM shape, 6 launches (for rocprofv3 --pmc passes via scripts/pmc.sh). env GEMM_SHAPE=MxNxK [GEGLU=1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops  # noqa: E402

ops.init(0)
M, N, K = (int(v) for v in os.environ.get("GEMM_SHAPE", "8192x10240x1280").split("x"))
geglu = os.environ.get("GEGLU") == "1"
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    ops.linear(a, w, b, out=out, geglu=geglu)
torch.cuda.synchronize()
