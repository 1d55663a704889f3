"""Child of tests/test_gpu_gemm_variants.py: runs a few GEMM / conv launches under the MI355X_SD_GEMM_* switches of its environment
(read once per process) and prints one JSON line: sha256 of every output + rel-L2 against fp32 torch math."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from paddlemix_amd import _lib, ops  # noqa: E402

ops.init(0)
ed = _lib.elem_dtype()
res = {}
# the 256x160 three-stage tile (N = 1280 column tiles) with K on both sides of the loader-wave threshold, ragged M, residual
for M, N, K, resid in ((8192, 1280, 5120, True), (1000, 1280, 4096, False), (8192, 1280, 1280, True), (777, 2560, 8192, False)):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(ed)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(ed)
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).to(ed) if resid else None
    out = ops.linear(a, w, b, residual=r)
    ref = a[:512].float() @ w.float().t() + b + (r[:512].float() if resid else 0)
    res[f"gemm {M}x{N}x{K}{'+R' if resid else ''}"] = dict(
        sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
        rel=((out[:512].float() - ref).norm() / ref.norm()).item())
# epilogue forms (round 3): residual rows inside a wider buffer and ragged M / N (early residual fetch and its fall-backs: N % 8 != 0,
# fewer than TM + 2 K-tiles), no bias, out_scale, a 640-wide launch on the streaming 256x320 tile, GEGLU (bias in the accumulators)
for M, N, K, kind in ((8200, 1280, 1280, "wide"), (300, 1288, 640, "wide"), (8200, 1284, 1280, "plain"), (515, 640, 256, "plain"),
                      (32768, 640, 640, "nobias"), (4100, 640, 640, "scale"), (2048, 2560, 1280, "geglu"), (8192, 10240, 1280, "geglu")):
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 1)
    a = torch.randn(M, K, device="cuda", generator=g).to(ed)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(ed)
    b = None if kind == "nobias" else torch.randn(N, device="cuda", generator=g)
    if kind == "geglu":
        out = ops.linear(a, w, b, geglu=True)
        y = (a[:256].float() @ w.float().t() + b).reshape(256, N // 32, 2, 16)
        ref = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(256, N // 2)
        got = out[:256].float()
    else:
        rbig = torch.randn(M, N + 24, device="cuda", generator=g).to(ed)
        r = rbig[:, 8:8 + N] if kind == "wide" else rbig[:, :N].contiguous()
        sc = 0.5 if kind == "scale" else 1.0
        out = ops.linear(a, w, b, residual=r, out_scale=sc)
        rows = slice(M - 256, M)
        ref = (a[rows].float() @ w.float().t() + (b if b is not None else 0) + r[rows].float()) * sc
        got = out[rows].float()
    res[f"gemm {M}x{N}x{K} {kind}"] = dict(sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
                                           rel=((got - ref).norm() / ref.norm()).item())
# the fp32 residual stream (MI355X_SD_R_F32 | MI355X_SD_OUT_F32): fp32 residual rows in, fp32 sum out -- the early fetch holds two
# row-tiles of them and requests the other two from the epilogue; ragged M, a launch too short for it (K = 128), an N it refuses
for M, N, K in ((8192, 1280, 1280), (8200, 1280, 640), (32768, 640, 640), (1000, 1280, 128), (777, 1284, 1280)):
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 2)
    a = torch.randn(M, K, device="cuda", generator=g).to(ed)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(ed)
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g)
    out = ops.linear(a, w, b, residual=r, out_f32=True)
    rows = slice(M - 256, M)
    ref = a[rows].float() @ w.float().t() + b + r[rows]
    res[f"gemm {M}x{N}x{K} fp32 stream"] = dict(sha=hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16],
                                                rel=((out[rows] - ref).norm() / ref.norm()).item())
# resnet convs: time-embedding row bias (conv1) and shortcut residual (conv2), the operands the batched epilogue fetches per row-tile
for B, H, W, Cin, Cout in ((2, 32, 32, 640, 640), (1, 30, 34, 320, 640)):
    g = torch.Generator(device="cuda").manual_seed(B + H + Cin + 7)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(ed)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (9 * Cin) ** 0.5).to(ed)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    b = torch.randn(Cout, device="cuda", generator=g)
    rb = torch.randn(B, Cout, device="cuda", generator=g)
    r = torch.randn(B * H * W, Cout, device="cuda", generator=g).to(ed)
    out = ops.conv3x3(x, wp, b, rowbias=rb, residual=r, out_scale=0.7)
    ref = (F.conv2d(x[:1].float().permute(0, 3, 1, 2), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout) + rb[0] + r[:H * W].float()) * 0.7
    res[f"conv {B}x{H}x{W}x{Cin}->{Cout} +temb+R"] = dict(
        sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
        rel=((out[:H * W].float() - ref).norm() / ref.norm()).item())
# launches of 1, 2, 3 and 5 K-tiles (the interleaved loop's unrolled tail alone; too short for the early residual fetch) and one
# small-M launch that takes split-K slices (uneven last slice)
for M, N, K, resid in ((300, 1280, 64, True), (4100, 2560, 128, False), (300, 640, 192, True), (8192, 1280, 320, True),
                       (256, 1280, 5184, False), (96, 640, 2240, True)):
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 3)
    a = torch.randn(M, K, device="cuda", generator=g).to(ed)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(ed)
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).to(ed) if resid else None
    out = ops.linear(a, w, b, residual=r)
    ref = a.float() @ w.float().t() + b + (r.float() if resid else 0)
    res[f"gemm {M}x{N}x{K}{'+R' if resid else ''} short"] = dict(
        sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
        rel=((out.float() - ref).norm() / ref.norm()).item())
for B, H, W, Cin, Cout in ((8, 32, 32, 1280, 1280), (2, 30, 34, 640, 1280)):      # 11520- / 5760-deep implicit-GEMM convs
    g = torch.Generator(device="cuda").manual_seed(B + H + Cin)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(ed)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (9 * Cin) ** 0.5).to(ed)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    b = torch.randn(Cout, device="cuda", generator=g)
    out = ops.conv3x3(x, wp, b)
    ref = F.conv2d(x[:1].float().permute(0, 3, 1, 2), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    res[f"conv {B}x{H}x{W}x{Cin}->{Cout}"] = dict(
        sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
        rel=((out[:H * W].float() - ref).norm() / ref.norm()).item())
# wide launches without per-row operands (what the four-wave 256 x 256 tile of csrc/gemm_w4.hip takes, round 6): one K-tile (its
# prologue + last iteration alone), two, an odd count; ragged M and N; SiLU; fewer tiles than CUs and several rounds of persistent blocks
for M, N, K, kind in ((1000, 1288, 64, "plain"), (515, 1928, 128, "silu"), (3000, 3840, 1280, "plain"), (8192, 3840, 1280, "nobias"),
                      (33000, 1536, 1536, "plain"), (700, 2048, 320, "plain")):
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 5)
    a = torch.randn(M, K, device="cuda", generator=g).to(ed)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(ed)
    b = None if kind == "nobias" else torch.randn(N, device="cuda", generator=g)
    out = ops.linear(a, w, b, silu=kind == "silu")
    rows = slice(M - 256, M)
    ref = a[rows].float() @ w.float().t() + (b if b is not None else 0)
    if kind == "silu":
        ref = F.silu(ref)
    res[f"gemm {M}x{N}x{K} wide {kind}"] = dict(sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
                                                rel=((out[rows].float() - ref).norm() / ref.norm()).item())
# the MMDiT block forms on 16-bit weights (csrc/gemm_w4.hip EX: gate + residual, tanh-GELU, C rows remapped into a joint buffer, A rows
# read out of one) at row counts where a 256-row tile lies inside one batch, and one where it does not (4100 rows in 4 batches: the
# launcher must leave that to the other kernels)
for M, N, K, kind in ((8192, 1536, 1536, "gate+R"), (8192, 6144, 1536, "gelu"), (8192, 4608, 1536, "remap"), (8192, 1536, 1536, "a-remap"),
                      (4100, 1536, 1536, "gate+R"), (16384, 1536, 6144, "gate+R")):
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 6)
    a = torch.randn(M, K, device="cuda", generator=g).to(ed)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(ed)
    b = torch.randn(N, device="cuda", generator=g)
    rows = slice(M - 256, M)
    ref = a[rows].float() @ w.float().t() + b
    if kind == "gate+R":
        nb = 4
        gt = torch.randn(nb, N, device="cuda", generator=g)
        r = torch.randn(M, N, device="cuda", generator=g).to(ed)
        out = ops.linear_ex(a, w, b, gate=gt, rows_per_batch=(M + nb - 1) // nb, residual=r)
        ref = r[rows].float() + gt[nb - 1] * ref
        got = out[rows].float()
    elif kind == "gelu":
        out = ops.linear_ex(a, w, b, gelu_tanh=True)
        ref = F.gelu(ref, approximate="tanh")
        got = out[rows].float()
    elif kind == "remap":
        rpb, extra = M // 4, 154
        buf = torch.zeros(4 * (rpb + extra) * N, device="cuda", dtype=ed)
        ops.linear_ex(a, w, b, out=buf, c_rows_per_batch=rpb, c_batch_stride=(rpb + extra) * N, M=M)
        out = buf
        got = buf.view(4, rpb + extra, N)[3, rpb - 256:rpb].float()
        assert (buf.view(4, rpb + extra, N)[:, rpb:] == 0).all(), "wrote outside the remapped rows"
    else:   # A rows of 4 batches read out of a buffer with 154 other rows behind each batch
        rpb, extra = M // 4, 154
        abuf = torch.randn(4 * (rpb + extra) * K, device="cuda", generator=g).to(ed)
        out = ops.linear_ex(abuf, w, b, a_rows_per_batch=rpb, a_batch_stride=(rpb + extra) * K, M=M)
        a_last = abuf.view(4, rpb + extra, K)[3, rpb - 256:rpb].float()
        ref = a_last @ w.float().t() + b
        got = out[rows].float()
    res[f"gemm ex {M}x{N}x{K} {kind}"] = dict(sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
                                              rel=((got - ref).norm() / ref.norm()).item())
# weight-only fp8 on a just-in-time widened matrix (round 5: kernel instantiations that keep the per-channel scale in registers,
# csrc/gemm_epilogue.h WS; under MI355X_SD_NO_PIPE the same launches run the read-where-used epilogue of the generic loop): plain,
# tanh-GELU (SD3 FF1), gate + residual (SD3 to_out / FF2), a row-remapped output (the joint QKV buffer), ragged M, ragged N
from paddlemix_amd.sd3 import dequantize_fp8_rows, quantize_fp8_rows  # noqa: E402
for M, N, K, kind in ((8200, 1536, 1536, "gate+R"), (4100, 6144, 1536, "gelu"), (8192, 4608, 1536, "remap"), (4096, 1540, 1536, "plain"),
                      (16384, 1536, 6144, "gate+R"), (8200, 1544, 1536, "plain")):   # (the last: ragged M and N on the four-wave tile's WS form)
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 4)
    a = torch.randn(M, K, device="cuda", generator=g).to(ed)
    w8, ws = quantize_fp8_rows(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    b = torch.randn(N, device="cuda", generator=g)
    rows = slice(M - 256, M)
    ref = a[rows].float() @ dequantize_fp8_rows(w8, ws).t() + b
    if kind == "gate+R":
        nb = 4
        gt = torch.randn(nb, N, device="cuda", generator=g)
        r = torch.randn(M, N, device="cuda", generator=g).to(ed)
        out = ops.linear_ex(a, w8, b, w_scale=ws, gate=gt, rows_per_batch=(M + nb - 1) // nb, residual=r)
        ref = r[rows].float() + gt[nb - 1] * ref
        got = out[rows].float()
    elif kind == "gelu":
        out = ops.linear_ex(a, w8, b, w_scale=ws, gelu_tanh=True)
        ref = F.gelu(ref, approximate="tanh")
        got = out[rows].float()
    elif kind == "remap":   # 4 batches of M/4 rows written behind 154 other rows each (what the MMDiT's joint sequence looks like)
        rpb, extra = M // 4, 154
        buf = torch.zeros(4 * (rpb + extra) * N, device="cuda", dtype=ed)
        ops.linear_ex(a, w8, b, w_scale=ws, out=buf, c_rows_per_batch=rpb, c_batch_stride=(rpb + extra) * N, M=M)
        out = buf
        got = buf.view(4, rpb + extra, N)[3, rpb - 256:rpb].float()
        assert (buf.view(4, rpb + extra, N)[:, rpb:] == 0).all(), "wrote outside the remapped rows"
    else:
        out = ops.linear_ex(a, w8, b, w_scale=ws)
        got = out[rows].float()
    res[f"gemm fp8w {M}x{N}x{K} {kind}"] = dict(sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
                                                rel=((got - ref).norm() / ref.norm()).item())
print("VARIANT_JSON " + json.dumps(res))
