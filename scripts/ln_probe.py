import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops
ops.init(0)
for rows, C in [(8192, 1280), (32768, 640)]:
    x = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    y = torch.empty_like(x)
    for _ in range(5):
        ops.layer_norm(x, g, b, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.layer_norm(x, g, b, out=y)
    e1.record()
    torch.cuda.synchronize()
    print(os.environ.get("MI355X_SD_LN_CAP"), os.environ.get("MI355X_SD_LN_R2"), rows, C, round(5 * e0.elapsed_time(e1), 2), "us")
