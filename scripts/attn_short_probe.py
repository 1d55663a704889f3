"""Cross-attention launches of the SDXL bs-8 step (77 text tokens): parity against fp32 math and time per launch, under the
MI355X_SD_ATTN_* switches of the environment (MI355X_SD_ATTN_NO_SHORT=1: the flash kernel; MI355X_SD_ATTN_SHORT_QT=n: query tiles
per block of the single-pass kernel). Q / O live inside [B, S, H*64] rows like the model's projections."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops  # noqa: E402

ops.init(0)
torch.manual_seed(0)
label = " ".join(f"{k[10:]}={v}" for k, v in os.environ.items() if k.startswith("MI355X_SD_ATTN")) or "default"
for B, H, Sq, Skv, n in ((8, 20, 1024, 77, 60), (8, 10, 4096, 77, 10), (2, 8, 4096, 77, 0), (8, 20, 1024, 64, 0), (8, 20, 1024, 128, 0)):
    sets = []
    for _ in range(4):
        q = torch.randn(B, Sq, H, 64, device="cuda").to(torch.bfloat16)
        k = torch.randn(B, Skv, H, 64, device="cuda").to(torch.bfloat16)
        v = torch.randn(B, Skv, H, 64, device="cuda").to(torch.bfloat16)
        sets.append((q, k, v))
    q, k, v = sets[0]
    out = ops.sdpa(q, k, v)
    qf, kf, vf = (t[:2].float().permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) / 8.0, -1) @ vf).permute(0, 2, 1, 3)
    rel = ((out[:2].float() - ref).norm() / ref.norm()).item()
    best = 1e9
    for _ in range(3):
        for s in sets:
            ops.sdpa(*s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(8):
            ops.sdpa(*sets[i % 4])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8 * 1e3)
    mb = (2 * B * Sq * H * 64 * 2 + 2 * B * Skv * H * 64 * 2) / 1e6
    print(f"[{label}] sdpa {B}x{H}x{Sq}x{Skv}x64: rel {rel:.2e}  {best:7.1f} us  {mb / best:6.2f} TB/s"
          + (f"  x{n} = {best * n / 1e3:.2f} ms per step" if n else ""), flush=True)
