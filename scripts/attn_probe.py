import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops
ops.init(0)
LOG2 = os.environ.get("ATTN_PROBE_LOG2") == "1"    # the UNet's self-attention form (scale folded into q): MI355X_SD_SDPA_LOG2
for B, H, S, D in [(8, 10, 4096, 64), (8, 20, 1024, 64), (8, 24, 4250, 64)]:
    q, k, v = (torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16) for _ in range(3))
    if LOG2:
        q = (q.float() * (D ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
    o = torch.empty_like(q)
    for _ in range(3):
        ops.sdpa(q, k, v, out=o, log2=LOG2)
torch.cuda.synchronize()
