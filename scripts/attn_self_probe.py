"""Self-attention launches of the SDXL bs-8 step in the UNet's own form (scale * log2 e folded into q, MI355X_SD_SDPA_LOG2): parity
against fp32 math and time per launch under the MI355X_SD_ATTN_* switches of the environment (MI355X_SD_ATTN_NO_WIDE = 8-byte O
stores; the priority / staggered-start experiments of round 3, profiles/r03_s7_attn_self_experiments.txt, were deleted in round 4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops  # noqa: E402

ops.init(0)
torch.manual_seed(0)
label = " ".join(f"{k[10:]}={v}" for k, v in os.environ.items() if k.startswith("MI355X_SD_ATTN")) or "default"
C2 = 64 ** -0.5 * 1.4426950408889634
for B, H, S, n in ((8, 10, 4096, 10), (8, 20, 1024, 60), (8, 24, 4250, 0)):
    sets = []
    for _ in range(3):
        qkv = torch.randn(B, S, 3, H, 64, device="cuda")
        qkv[:, :, 0] *= C2
        sets.append(qkv.to(torch.bfloat16))
    outs = [torch.empty(B, S, H, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    run = lambda i: ops.sdpa(sets[i][:, :, 0], sets[i][:, :, 1], sets[i][:, :, 2], out=outs[i], log2=True)   # noqa: E731
    out = run(0)
    qf, kf, vf = (sets[0][:1, :, j].float().permute(0, 2, 1, 3) for j in range(3))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * 0.6931471805599453, -1) @ vf).permute(0, 2, 1, 3)
    rel = ((out[:1].float() - ref).norm() / ref.norm()).item()
    best = 1e9
    for _ in range(3):
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(6):
            run(i % 3)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 6 * 1e3)
    tf = 4.0 * B * H * S * S * 64 / best / 1e6
    print(f"[{label}] sdpa {B}x{H}x{S}x{S}x64 log2: rel {rel:.2e}  {best:7.1f} us  {tf:6.0f} TF" + (f"  x{n} = {best * n / 1e3:.2f} ms per step" if n else ""), flush=True)
