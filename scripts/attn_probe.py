import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops
ops.init(0)
for B, H, S, D in [(8, 10, 4096, 64), (8, 20, 1024, 64), (8, 24, 4250, 64)]:
    q, k, v = (torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16) for _ in range(3))
    o = torch.empty_like(q)
    for _ in range(3):
        ops.sdpa(q, k, v, out=o)
torch.cuda.synchronize()
