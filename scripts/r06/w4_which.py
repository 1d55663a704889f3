import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from paddlemix_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K, kw) in ((32768, 6144, 1536, dict(gelu_tanh=True)), (32768, 4608, 1536, {}), (8192, 3840, 1280, {})):
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    for _ in range(5):
        ops.linear_ex(a, w, b, **kw)
    torch.cuda.synchronize()
