"""Debug aid: where do GEMM outputs differ from fp32 math? Run under the MI355X_SD_GEMM_* switches of the environment."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import ops  # noqa: E402

ops.init(0)
label = " ".join(f"{k[10:]}={v}" for k, v in os.environ.items() if k.startswith("MI355X_SD_")) or "default"
for M, N, K, bias_on, resid in ((4100, 700, 640, True, True), (4100, 700, 640, True, False), (4100, 700, 640, False, True), (4100, 704, 640, True, True),
                                (4096, 700, 640, True, True), (8192, 1280, 1280, True, True), (8192, 1280, 5120, True, True), (8200, 1280, 640, True, True),
                                (32768, 640, 640, True, True), (2048, 2560, 1280, True, False)):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, generator=g) * 0.1 if bias_on else None
    res = torch.randn(M, N, generator=g).to(torch.bfloat16) if resid else None
    ad, wd = a.cuda(), w.cuda()
    ref = ad.float() @ wd.float().t()
    if bias is not None:
        ref = ref + bias.cuda()
    if res is not None:
        ref = ref + res.cuda().float()
    out = ops.linear(ad, wd, bias.cuda() if bias is not None else None, residual=res.cuda() if res is not None else None).float()
    err = (out - ref).abs()
    scale = ref.abs().max().item()
    bad = (err > 2 * 2 ** -8 * scale + 1e-6) | ~torch.isfinite(out)
    nb = int(bad.sum())
    msg = f"[{label}] {M}x{N}x{K} bias={bias_on} R={resid}: rel {((out - ref).norm() / ref.norm()).item():.3e} bad {nb}"
    if nb:
        idx = bad.nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        msg += f" rows[{rows.min().item()}..{rows.max().item()}] ({len(rows.unique())} distinct) cols[{cols.min().item()}..{cols.max().item()}] ({len(cols.unique())} distinct)"
        msg += " first: " + " ".join(f"({r},{c}):{out[r, c].item():.3f}/{ref[r, c].item():.3f}" for r, c in idx[:6].tolist())
        msg += f" cols%128: {sorted(set((cols % 128).tolist()))[:24]}"
    print(msg, flush=True)
