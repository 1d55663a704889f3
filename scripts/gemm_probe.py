import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops
ops.init(0)
shapes = [(8192, 8192, 8192), (8192, 1280, 1280), (8192, 3840, 1280), (8192, 10240, 1280), (32768, 640, 640), (8192, 1280, 5120)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.linear(a, w, out=out)
torch.cuda.synchronize()
