"""Child of tests/test_gpu_switches.py: a few launches whose path depends on the remaining A/B switches of the library (read once
per process) -- attention on the short-key / two-query-tile / wide-store paths, a small-M GEMM that takes split-K slices, GroupNorm
on the one-launch path -- and a mini-SDXL UNet forward. Prints one JSON line: sha256 of every output + rel-L2 against fp32 math."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from paddlemix_amd import _lib, ops  # noqa: E402

ops.init(0)
ed = _lib.elem_dtype()
res = {}


def put(name, out, ref):
    res[name] = dict(sha=hashlib.sha256(out.detach().cpu().contiguous().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
                     rel=((out.float().cpu() - ref.float().cpu()).norm() / ref.float().cpu().norm()).item())


# attention: cross-attention geometry (77 keys: short-key kernel; with it off, two query tiles per block), self-attention (wide stores)
for B, H, Sq, Skv in ((8, 20, 1024, 77), (2, 10, 1024, 1024), (1, 5, 200, 77)):
    g = torch.Generator(device="cuda").manual_seed(B + Sq + Skv)
    q, k, v = (torch.randn(B, S, H, 64, device="cuda", generator=g).to(ed) for S in (Sq, Skv, Skv))
    out = ops.sdpa(q, k, v)
    qf, kf, vf = (t[:1].float().permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) / 8.0, -1) @ vf).permute(0, 2, 1, 3)
    put(f"sdpa {B}x{H}x{Sq}x{Skv}", out[:1], ref)
# small-M GEMM: split-K slices when the workspace is set (ops.init does), plain tiles otherwise
g = torch.Generator(device="cuda").manual_seed(5)
a = torch.randn(64, 4096, device="cuda", generator=g).to(ed)
w = (torch.randn(1280, 4096, device="cuda", generator=g) / 64).to(ed)
b = torch.randn(1280, device="cuda", generator=g)
put("gemm 64x1280x4096", ops.linear(a, w, b), a.float() @ w.float().t() + b)
# weight-only fp8 at large M: the matrix is widened once into the workspace and multiplied by the 16-bit kernels
# (MI355X_SD_NO_WIDEN_F8: converted in the fragment load of the generic loop instead)
from paddlemix_amd.sd3 import dequantize_fp8_rows, quantize_fp8_rows  # noqa: E402
for M, N, K, resid in ((4096, 1536, 1536, True), (8200, 4608, 1536, False)):
    a = torch.randn(M, K, device="cuda", generator=g).to(ed)
    w8, ws = quantize_fp8_rows(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).to(ed) if resid else None
    out = ops.linear_ex(a, w8, b, w_scale=ws, residual=r)
    put(f"gemm fp8w {M}x{N}x{K}", out, a.float() @ dequantize_fp8_rows(w8, ws).t() + b + (r.float() if resid else 0))
# GroupNorm + SiLU on a geometry the one-launch kernel takes (32 x 32 x 640, 32 groups)
x = torch.randn(2, 32 * 32, 640, device="cuda", generator=g).to(ed)
gamma, beta = torch.randn(640, device="cuda", generator=g), torch.randn(640, device="cuda", generator=g)
# (the planners' rule: one launch where groupnorm_act_fits says so -- MI355X_SD_NO_GN_FUSED turns that answer off -- else statistics + apply)
y = (ops.groupnorm_act(x, gamma, beta, 32, 1e-5, silu=True) if ops.groupnorm_act_fits(32 * 32, 640, 32)
     else ops.group_norm(x, gamma, beta, 32, 1e-5, silu=True))
xf = x.float().reshape(2, 32 * 32, 32, 20)
mu, var = xf.mean((1, 3), keepdim=True), xf.var((1, 3), unbiased=False, keepdim=True)
yr = ((xf - mu) / (var + 1e-5).sqrt()).reshape(2, 32 * 32, 640) * gamma + beta
put("groupnorm_silu 2x1024x640", y, yr * torch.sigmoid(yr))
print("SWITCH_JSON " + json.dumps(res))
