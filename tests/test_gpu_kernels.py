"""-m gpu parity tests: every HIP kernel, called through the C ABI, against the CPU oracle (oracle/unet_ref.py) on
the same seeded inputs.  Tolerances: inputs/outputs are bf16 (8-bit mantissa, ulp 2^-8 = 3.9e-3 relative) with fp32
accumulation, so a single op is held to rel-L2 <= 4e-3 and max-abs <= 2 bf16 ulps of the output scale against the
fp32 oracle evaluated on the same bf16-rounded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from paddlemix_amd import ops as o
    o.init(0)
    return o


def bfr(t):  # bf16-representable fp32
    return t.to(torch.bfloat16).float()


def dev(t, dtype=torch.bfloat16):
    return t.to("cuda", dtype)


def check(out, ref, rel=4e-3, what=""):
    out = out.float().cpu()
    err = (out - ref).norm() / ref.norm().clamp_min(1e-12)
    mx = (out - ref).abs().max()
    scale = ref.abs().max()
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    assert err < rel, f"{what}: rel-L2 {err:.3e} (max abs {mx:.3e}, scale {scale:.3e})"
    assert mx <= 2 * 2 ** -8 * scale + 1e-6, f"{what}: max abs {mx:.3e} vs scale {scale:.3e}"


def test_probe_layouts(ops):
    """The lane->element maps the kernels assume (cdna guide section 3; asymmetric operands so a transpose shows)."""
    raw = ops.probe_layouts().cpu()
    A16 = torch.tensor([[((r * 7 + k * 3) % 11) - 5 for k in range(32)] for r in range(16)], dtype=torch.float32)
    B16 = torch.tensor([[((k * 5 + c * 2) % 13) - 6 for c in range(16)] for k in range(32)], dtype=torch.float32)
    C16 = A16 @ B16
    A32 = torch.tensor([[((r * 7 + k * 3) % 11) - 5 for k in range(16)] for r in range(32)], dtype=torch.float32)
    B32 = torch.tensor([[((k * 5 + c * 2) % 13) - 6 for c in range(32)] for k in range(16)], dtype=torch.float32)
    C32 = A32 @ B32
    for l in range(64):
        for r in range(4):
            assert raw[l, r] == C16[(l >> 4) * 4 + r, l & 15], ("mfma16", l, r)
        for r in range(16):
            assert raw[l, 4 + r] == C32[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31], ("mfma32", l, r)
        for j in range(4):
            assert raw[l, 20 + j] == (l & 15) + j * 16 + (l >> 4) * 64, ("tr16", l, j)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 328, 72), (8, 1280, 320), (1024, 640, 2048), (77, 96, 768)])
def test_linear_plain(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a, w = bfr(torch.randn(M, K, generator=g)), bfr(torch.randn(K, N, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g) * 0.1
    ref = U.linear({"l.weight": w, "l.bias": bias}, "l", a)
    out = ops.linear(dev(a), dev(w.t().contiguous()), dev(bias, torch.float32))
    check(out, ref, what=f"linear {M}x{N}x{K}")


def test_linear_epilogues_and_strides(ops):
    g = torch.Generator().manual_seed(3)
    M, N, K, B = 192, 256, 128, 3
    a_full = bfr(torch.randn(M, K + 64, generator=g))
    a = a_full[:, 32:32 + K]                      # strided A view
    w = bfr(torch.randn(K, N, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g) * 0.1
    rowbias = torch.randn(B, N + 8, generator=g)[:, :N]
    res_full = bfr(torch.randn(M, N + 16, generator=g))
    res = res_full[:, 8:8 + N]
    ref = (a @ w + bias + rowbias.repeat_interleave(M // B, 0) + res) * 0.5
    out_full = torch.zeros(M, N + 24, device="cuda", dtype=torch.bfloat16)
    a_d = dev(a_full)[:, 32:32 + K]
    rb_d = dev(rowbias.contiguous(), torch.float32)
    r_d = dev(res_full)[:, 8:8 + N]
    ops.linear(a_d, dev(w.t().contiguous()), dev(bias, torch.float32), rowbias=rb_d, rows_per_batch=M // B,
               residual=r_d, out=out_full[:, 16:16 + N], out_scale=0.5)
    check(out_full[:, 16:16 + N], ref, what="linear epilogue")
    assert (out_full[:, :16] == 0).all() and (out_full[:, 16 + N:] == 0).all(), "wrote outside the output view"
    # SiLU epilogue + fp32 output
    ref2 = F.silu(a @ w + bias)
    out2 = ops.linear(a_d, dev(w.t().contiguous()), dev(bias, torch.float32), silu=True, out_f32=True)
    assert out2.dtype == torch.float32
    check(out2, ref2, what="linear silu f32")


def test_geglu_ff_matches_reference(ops):
    """FeedForward with GEGLU (attention.py:670-677, activations.py:101-104) through the interleaved-weight epilogue."""
    g = torch.Generator().manual_seed(5)
    M, C = 300, 128
    P = {"ff.net.0.proj.weight": bfr(torch.randn(C, 8 * C, generator=g) / math.sqrt(C)),
         "ff.net.0.proj.bias": torch.randn(8 * C, generator=g) * 0.1,
         "ff.net.2.weight": bfr(torch.randn(4 * C, C, generator=g) / math.sqrt(4 * C)),
         "ff.net.2.bias": torch.randn(C, generator=g) * 0.1}
    x = bfr(torch.randn(M, C, generator=g))
    hg = U.linear(P, "ff.net.0.proj", x)
    h, gate = hg.chunk(2, -1)
    mid_ref = h * F.gelu(gate)
    w1 = P["ff.net.0.proj.weight"].t()
    half = 4 * C
    w1i = torch.stack([w1[:half].reshape(half // 16, 16, -1), w1[half:].reshape(half // 16, 16, -1)], 1).reshape(2 * half, -1)
    b1 = P["ff.net.0.proj.bias"]
    b1i = torch.stack([b1[:half].reshape(half // 16, 16), b1[half:].reshape(half // 16, 16)], 1).reshape(-1)
    mid = ops.linear(dev(x), dev(w1i.contiguous()), dev(b1i.contiguous(), torch.float32), geglu=True)
    assert mid.shape == (M, 4 * C)
    check(mid, mid_ref, what="geglu")
    out = ops.linear(mid, dev(P["ff.net.2.weight"].t().contiguous()), dev(P["ff.net.2.bias"], torch.float32))
    check(out, U.linear(P, "ff.net.2", bfr(mid.float().cpu())), what="ff2")


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(2, 16, 16, 64, 128, 1, False), (1, 9, 13, 32, 64, 1, False),
                                                      (2, 16, 16, 64, 64, 2, False), (1, 8, 8, 64, 96, 1, True),
                                                      (1, 32, 32, 320, 320, 1, False), (1, 7, 5, 40, 64, 2, False)])
def test_conv3x3(ops, B, H, W, Cin, Cout, stride, up):
    g = torch.Generator().manual_seed(B * H + Cin + Cout + stride)
    x = bfr(torch.randn(B, Cin, H, W, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    P = {"c.conv.weight": w, "c.conv.bias": bias}
    if up:
        ref = U.upsample(P, "c", x)
    elif stride == 2:
        ref = U.downsample(P, "c", x)
    else:
        ref = U.conv2d(P, "c.conv", x)
    x_nhwc = dev(x.permute(0, 2, 3, 1).contiguous())
    wk = dev(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous())
    out = ops.conv3x3(x_nhwc, wk, dev(bias, torch.float32), stride=stride, upsample=up)
    Ho, Wo = ref.shape[2], ref.shape[3]
    check(out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref, what=f"conv3x3 {B,H,W,Cin,Cout,stride,up}")


def test_conv3x3_resnet_epilogues(ops):
    """conv1 (+temb broadcast, resnet.py:772-784) and conv2 (+shortcut, /output_scale_factor, :800-806) epilogues on
    channel-strided (concat-by-construction) views."""
    g = torch.Generator().manual_seed(11)
    B, H, W, C1, C2, Cout = 2, 8, 8, 64, 32, 64
    xcat = bfr(torch.randn(B, H, W, C1 + C2, generator=g))
    w = bfr(torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt(9 * (C1 + C2)))
    bias = torch.randn(Cout, generator=g) * 0.1
    temb = torch.randn(B, Cout, generator=g)
    res = bfr(torch.randn(B * H * W, Cout, generator=g))
    ref = F.conv2d(xcat.permute(0, 3, 1, 2), w, bias, padding=1) + temb[:, :, None, None]
    ref = (ref + res.reshape(B, H, W, Cout).permute(0, 3, 1, 2)) / 2.0
    big = torch.zeros(B * H * W, Cout + 64, device="cuda", dtype=torch.bfloat16)
    # rowbias is a column slice of a wider fp32 table (the batched time_emb_proj output)
    temb_d = dev(torch.cat([torch.zeros(B, 8), temb], 1).contiguous(), torch.float32)[:, 8:]
    ops.conv3x3(dev(xcat), dev(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()), dev(bias, torch.float32),
                rowbias=temb_d, residual=dev(res), out=big[:, 32:32 + Cout], out_scale=0.5)
    check(big[:, 32:32 + Cout].reshape(B, H, W, Cout).permute(0, 3, 1, 2), ref, what="conv epilogue")
    assert (big[:, :32] == 0).all() and (big[:, 32 + Cout:] == 0).all()


@pytest.mark.parametrize("M,N,K", [(64, 1280, 11520), (256, 1280, 5120), (1, 1280, 1280), (1000, 328, 2888)])
def test_linear_splitk_epilogues(ops, M, N, K):
    """Few rows against a long K: the launch splits K over blockIdx.y and the slices are reduced in fixed order before
    the fused epilogue (temb row bias + residual + scale); run twice -> bit-identical."""
    g = torch.Generator().manual_seed(M + N + K)
    a = bfr(torch.randn(M, K, generator=g))
    w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g) * 0.1
    rpb = M // 2 if M % 2 == 0 else M
    rb = torch.randn(M // rpb, N, generator=g)
    res = bfr(torch.randn(M, N, generator=g))
    ref = (a @ w.t() + bias + rb.repeat_interleave(rpb, 0) + res) * 0.5
    args = (dev(a), dev(w), dev(bias, torch.float32))
    kw = dict(rowbias=dev(rb, torch.float32), rows_per_batch=rpb, residual=dev(res), out_scale=0.5)
    out = ops.linear(*args, **kw)
    check(out, ref, what=f"split-K linear {M,N,K}")
    assert torch.equal(out, ops.linear(*args, **kw))
    out32 = ops.linear(dev(a), dev(w), dev(bias, torch.float32), out_f32=True)
    r32 = a @ w.t() + bias
    assert out32.dtype == torch.float32 and (out32.cpu() - r32).norm() / r32.norm() < 1e-3


@pytest.mark.parametrize("M,C,N,geglu", [(300, 320, 960, False), (8192, 1280, 1280, False), (1024, 640, 5120, True),
                                          (64, 1280, 10240, True), (77, 64, 64, False)])
def test_layernorm_folded_linear(ops, M, C, N, geglu):
    """row_stats + linear_ln == LayerNorm(eps 1e-5, affine) followed by the projection (attention.py:405-486), with a
    row offset so that mean >> std is exercised (the correction term cancels a large common component)."""
    g = torch.Generator().manual_seed(M + C + N)
    x = bfr(torch.randn(M, C, generator=g) * 2.0 + torch.randn(M, 1, generator=g) * 3.0)
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    w = torch.randn(N, C, generator=g) / math.sqrt(C)
    bias = torch.randn(N, generator=g) * 0.1
    wf = bfr(w * gamma[None, :])
    # oracle on the same folded bf16 weights: LN without affine, then W' and b' = bias + W beta
    xn = F.layer_norm(x, (C,), eps=1e-5)
    ref = xn @ wf.t() + (bias + w @ beta)
    if geglu:
        half = N // 2
        ref = ref[:, :half] * F.gelu(ref[:, half:])
        il = lambda t: torch.stack([t[:half].reshape(half // 16, 16, *t.shape[1:]),   # noqa: E731
                                    t[half:].reshape(half // 16, 16, *t.shape[1:])], 1).reshape(N, *t.shape[1:])
        wf_d, b_d = il(wf), il(bias + w @ beta)
    else:
        wf_d, b_d = wf, bias + w @ beta
    st = ops.row_stats(dev(x))
    mean, var = x.mean(-1), x.var(-1, unbiased=False)
    assert _close(st[:, 0].cpu(), torch.rsqrt(var + 1e-5), 1e-4) and _close(st[:, 1].cpu(), -mean * torch.rsqrt(var + 1e-5), 1e-4)
    out = ops.linear_ln(dev(x), st, dev(wf_d.contiguous()), dev(wf_d.float().sum(1), torch.float32),
                        dev(b_d.contiguous(), torch.float32), geglu=geglu)
    check(out, ref, what=f"ln-folded linear {M,C,N,geglu}")
    # and against the unfused device path (LayerNorm kernel writes bf16, then the GEMM): same tolerance class
    ln = ops.layer_norm(dev(x), dev(gamma, torch.float32), dev(beta, torch.float32))
    assert ln.shape == (M, C)


def _close(a, b, rel):
    return ((a - b).abs() <= rel * b.abs() + 1e-6).all()


@pytest.mark.parametrize("M,N,K,tile", [(777, 200, 1024, 128), (300, 1288, 640, 160), (8200, 1284, 1280, 160), (515, 644, 1280, 320),
                                          (4100, 700, 640, 320), (260, 260, 2048, 257), (9000, 1280, 128, 160)])
def test_pipelined_tiles_ragged_edges(ops, M, N, K, tile):
    """K % 64 == 0 takes the software-pipelined loops (gemm_pipe.hip): rows >= M / N are out-of-range buffer offsets
    that the hardware zero-fills. Ragged M and N against every tile family, forced through MI355X_SD_GEMM_TILE in a
    subprocess-free way (the env var is read once per process, so the picker's own choice is used when it is unset)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = bfr(torch.randn(M, K, generator=g))
    w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g) * 0.1
    res = bfr(torch.randn(M, N, generator=g))
    ref = a @ w.t() + bias + res
    out = ops.linear(dev(a), dev(w), dev(bias, torch.float32), residual=dev(res))
    check(out, ref, what=f"pipelined ragged {M,N,K}")
    # strided A / C views (concat-by-construction): rows live inside wider buffers
    abig = torch.zeros(M, K + 64, device="cuda", dtype=torch.bfloat16)
    abig[:, 32:32 + K] = dev(a)
    cbig = torch.zeros(M, N + 8, device="cuda", dtype=torch.bfloat16)
    ops.linear(abig[:, 32:32 + K], dev(w), dev(bias, torch.float32), residual=dev(res), out=cbig[:, 4:4 + N])
    assert torch.equal(cbig[:, 4:4 + N], out) and (cbig[:, :4] == 0).all() and (cbig[:, 4 + N:] == 0).all()


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(1, 9, 13, 64, 328, 1, False), (2, 17, 15, 128, 160, 2, False),
                                                      (1, 10, 6, 64, 644, 1, True), (3, 31, 33, 64, 64, 1, False)])
def test_conv3x3_pipelined_ragged(ops, B, H, W, Cin, Cout, stride, up):
    """implicit-GEMM conv through the pipelined loops (Cin % 64 == 0 -> K % 64 == 0): padding and ragged pixel / channel
    tiles are hardware zero-fill of the buffer loads."""
    g = torch.Generator().manual_seed(H * W + Cin + Cout)
    x = bfr(torch.randn(B, Cin, H, W, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, w, bias, stride=stride, padding=1)
    out = ops.conv3x3(dev(x.permute(0, 2, 3, 1).contiguous()), dev(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()),
                      dev(bias, torch.float32), stride=stride, upsample=up)
    Ho, Wo = ref.shape[2], ref.shape[3]
    check(out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref, what=f"pipelined conv {B,H,W,Cin,Cout,stride,up}")


def test_pipelined_loops_race_screen(ops):
    """The software-pipelined loops order their LDS-DMA writes and ds_reads only through counted vmcnt waits and one
    barrier per K-tile; a missing wait shows up as run-to-run differences long before it shows up as a wrong mean.
    40 repetitions each of a long-K GEMM on every pipelined tile family (+ a conv), all bit-identical, under load from a
    concurrently allocated second problem so the DMA timing varies."""
    g = torch.Generator().manual_seed(123)
    cases = [(2048, 1280, 8192, False), (4096, 640, 4096, False), (2048, 2560, 4096, True), (700, 328, 6144, False)]
    noise_a = dev(torch.randn(4096, 4096, generator=g))
    noise_w = dev(torch.randn(4096, 4096, generator=g))
    for M, N, K, geglu in cases:
        a = dev(bfr(torch.randn(M, K, generator=g)))
        w = dev(bfr(torch.randn(N, K, generator=g) / math.sqrt(K)))
        bias = dev(torch.randn(N, generator=g) * 0.1, torch.float32)
        first = ops.linear(a, w, bias, geglu=geglu).clone()
        for i in range(40):
            if i % 4 == 0:
                ops.linear(noise_a, noise_w)   # perturb cache / DMA timing between repetitions
            assert torch.equal(ops.linear(a, w, bias, geglu=geglu), first), (M, N, K, geglu, i)
        if not geglu:
            check(first, (a.float() @ w.float().t() + bias).cpu(), what=f"race-screen reference {M,N,K}")
    x = dev(bfr(torch.randn(2, 32, 32, 640, generator=g)))
    wc = dev(bfr(torch.randn(640, 9 * 640, generator=g) / math.sqrt(9 * 640)))
    first = ops.conv3x3(x, wc, None).clone()
    for i in range(40):
        assert torch.equal(ops.conv3x3(x, wc, None), first), ("conv", i)


def test_geglu_splitk(ops):
    g = torch.Generator().manual_seed(5)
    M, C = 64, 1280
    x = bfr(torch.randn(M, C, generator=g))
    w1 = bfr(torch.randn(8 * C, C, generator=g) / math.sqrt(C))
    b1 = torch.randn(8 * C, generator=g) * 0.1
    half = 4 * C
    h = x @ w1.t() + b1
    ref = h[:, :half] * F.gelu(h[:, half:])
    w1i = torch.stack([w1[:half].reshape(half // 16, 16, C), w1[half:].reshape(half // 16, 16, C)], 1).reshape(8 * C, C)
    b1i = torch.stack([b1[:half].reshape(half // 16, 16), b1[half:].reshape(half // 16, 16)], 1).reshape(-1)
    out = ops.linear(dev(x), dev(w1i.contiguous()), dev(b1i.contiguous(), torch.float32), geglu=True)
    check(out, ref, what="split-K geglu")


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(1, 8, 8, 1280, 1280, 1, False), (1, 16, 16, 640, 1280, 2, False),
                                                      (1, 8, 8, 2560, 1280, 1, True), (1, 32, 32, 320, 320, 1, False)])
def test_conv3x3_splitk(ops, B, H, W, Cin, Cout, stride, up):
    """SD-1.5 batch-1 low-resolution convolutions: 64..1024 output pixels against K = 9 Cin up to 23040."""
    g = torch.Generator().manual_seed(H + Cin + Cout + stride)
    x = bfr(torch.randn(B, Cin, H, W, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, w, bias, stride=stride, padding=1)
    out = ops.conv3x3(dev(x.permute(0, 2, 3, 1).contiguous()), dev(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()),
                      dev(bias, torch.float32), stride=stride, upsample=up)
    Ho, Wo = ref.shape[2], ref.shape[3]
    check(out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref, what=f"split-K conv {B,H,W,Cin,Cout,stride,up}")


@pytest.mark.parametrize("B,H,Sq,Skv,D", [(2, 4, 256, 256, 64), (1, 8, 1024, 1024, 64), (2, 5, 200, 77, 64),
                                          (1, 8, 320, 320, 40), (1, 8, 128, 77, 80), (1, 8, 64, 64, 160),
                                          (1, 2, 33, 130, 8), (1, 2, 4096, 4096, 64),
                                          # ragged d = 64 launches of the 16x16x32 kernel: query rows past Sq (clamped reads, skipped
                                          # stores) together with a masked tail key tile; SD3's joint sequence length
                                          (1, 4, 257, 192, 64), (1, 4, 96, 331, 64), (1, 3, 4250, 4250, 64)])
def test_sdpa(ops, B, H, Sq, Skv, D):
    g = torch.Generator().manual_seed(Sq + Skv + D)
    q = bfr(torch.randn(B, Sq, H, D, generator=g))
    k = bfr(torch.randn(B, Skv, H, D, generator=g))
    v = bfr(torch.randn(B, Skv, H, D, generator=g))
    ref = U.sdpa_math(q, k, v)
    out = ops.sdpa(dev(q), dev(k), dev(v))
    check(out, ref, rel=5e-3, what=f"sdpa {B,H,Sq,Skv,D}")
    if D == 64 and Skv > 128:   # the base-2 form the UNet builder uses (scale * log2(e) folded into q)
        q2 = bfr(q * (D ** -0.5 * 1.4426950408889634))
        ref2 = U.sdpa_math(q2, k, v, scale=0.6931471805599453)
        out2 = ops.sdpa(dev(q2), dev(k), dev(v), log2=True)
        check(out2, ref2, rel=5e-3, what=f"sdpa log2 {B,H,Sq,Skv,D}")


@pytest.mark.parametrize("B,H,Sq,Skv,T,D,scale", [(2, 8, 1024, 77, 4, 64, 1.0), (1, 10, 4096, 77, 4, 64, 0.6),
                                                   (2, 8, 256, 77, 16, 40, 0.35), (1, 8, 64, 77, 4, 160, 1.0)])
def test_sdpa_accum_ip_adapter_tokens(ops, B, H, Sq, Skv, T, D, scale):
    """mi355x_sd_sdpa_accum: out = attn(q, k_text, v_text) + scale * attn(q, k_ip, v_ip), the arithmetic of
    IPAdapterAttnProcessor.__call__ (attention_processor.py:1871-1886) with T image tokens as a second, tiny key set."""
    g = torch.Generator().manual_seed(Sq + T + D)
    q = bfr(torch.randn(B, Sq, H, D, generator=g))
    k, v = (bfr(torch.randn(B, Skv, H, D, generator=g)) for _ in range(2))
    ki, vi = (bfr(torch.randn(B, T, H, D, generator=g)) for _ in range(2))
    ref = U.sdpa_math(q, k, v) + scale * U.sdpa_math(q, ki, vi)
    out = ops.sdpa(dev(q), dev(k), dev(v))
    first = out.clone()
    ops.sdpa(dev(q), dev(ki), dev(vi), out=out, accum=scale)
    check(out, ref, rel=6e-3, what=f"sdpa_accum {B,H,Sq,Skv,T,D}")   # two bf16 roundings of the text half
    ops.sdpa(dev(q), dev(ki), dev(vi), out=first, accum=0.0)          # out += 0 * attention leaves it alone
    assert torch.equal(first, ops.sdpa(dev(q), dev(k), dev(v)))


def test_sdpa_fused_qkv_strides_and_spike(ops):
    """q/k/v consumed in place from a fused [rows, 3C] projection buffer; plus a spiked key that forces the
    online-softmax rescale at a late tile (cdna guide 5.4 rule 26)."""
    g = torch.Generator().manual_seed(21)
    B, S, H, D = 2, 384, 4, 64
    C = H * D
    qkv = bfr(torch.randn(B, S, 3 * C, generator=g))
    qkv[0, 300, C:2 * C] *= 12.0  # key row 300 of batch 0: large scores in tile 4
    qkv = bfr(qkv)  # x12 leaves the bf16 grid: re-round so oracle and device see the same keys
    q, k, v = (qkv[..., i * C:(i + 1) * C].reshape(B, S, H, D) for i in range(3))
    ref = U.sdpa_math(q, k, v)
    d = dev(qkv)
    qd, kd, vd = (d[..., i * C:(i + 1) * C].unflatten(-1, (H, D)) for i in range(3))
    out = ops.sdpa(qd, kd, vd)
    check(out, ref, rel=5e-3, what="sdpa fused qkv")


@pytest.mark.parametrize("log2", [False, True])
@pytest.mark.parametrize("tile,half", [(5, 0), (5, 1), (15, 1), (14, 0), (1, 0)])
def test_sdpa_lazy_maximum_guard_fires(ops, tile, half, log2):
    """The unmasked d = 64 kernels exponentiate against a running reference without looking for a tile's maximum and redo a tile the
    exact way when a partial row sum leaves the safe range (csrc/attention.hip, LAZY_PSUM_LIMIT: 2^60, 2^15 in the IEEE-half build).
    Bounded random data never takes that branch (cdna guide 5.4 rule 26): the keys from one half-tile on are scaled up so that the
    scores jump by tens of base-2 units there -- at the first and the second half of a tile, the last and the second-to-last tile,
    and right after the first tile -- in the plain and the folded-scale (log2) form. scripts/c/attn_check.c does the same against a
    host float64 reference without torch."""
    from paddlemix_amd import _lib
    g = torch.Generator().manual_seed(100 * tile + half)
    B, H, Sq, Skv, D = 1, 3, 512, 1024, 64
    f = 6.0 if _lib.elem_dtype() == torch.float16 else 24.0
    q = bfr(torch.randn(B, Sq, H, D, generator=g) * (0.35 if log2 else 1.0))
    k = torch.randn(B, Skv, H, D, generator=g)
    k[:, tile * 64 + half * 32:] *= f
    k = bfr(k)
    v = bfr(torch.randn(B, Skv, H, D, generator=g))
    if log2:   # q.k IS the base-2 exponent: softmax_2(q k^T) v = softmax(q k^T ln 2) v
        ref = U.sdpa_math(q.double(), k.double(), v.double(), scale=math.log(2.0)).float()
        out = ops.sdpa(dev(q), dev(k), dev(v), log2=True)
    else:
        ref = U.sdpa_math(q.double(), k.double(), v.double()).float()
        out = ops.sdpa(dev(q), dev(k), dev(v))
    assert torch.isfinite(out.float()).all()
    check(out, ref, rel=5e-3, what=f"sdpa guard tile {tile} half {half} log2 {log2}")


def test_sdpa_additive_mask(ops):
    """mask semantics of test_model_xattn_mask (tests/models/test_models_unet_2d_condition.py:486-515): keep-all ==
    none; masking the last key == truncating it (bias = (1-m)*-10000, unet_2d_condition.py:921-927)."""
    g = torch.Generator().manual_seed(31)
    B, H, Sq, Skv, D = 2, 4, 96, 77, 64
    q, k, v = (bfr(torch.randn(B, s, H, D, generator=g)) for s in (Sq, Skv, Skv))
    qd, kd, vd = dev(q), dev(k), dev(v)
    none = ops.sdpa(qd, kd, vd)
    keep = ops.sdpa(qd, kd, vd, bias=torch.zeros(B, 1, 1, Skv, device="cuda"))
    # (two kernels since round 3: short key sequences without a mask take the single-pass kernel, a mask the flash kernel with
    # its deferred rescale: the same softmax rounded at different points -- equal within the per-op bar, not bit for bit)
    check(none, keep.float().cpu(), rel=5e-3, what="sdpa keep-all mask == no mask")
    m = torch.ones(B, Skv)
    m[:, -1] = 0
    bias = ((1 - m) * -10000.0)[:, None, None, :]
    masked = ops.sdpa(qd, kd, vd, bias=bias.cuda().contiguous())
    trunc = ops.sdpa(qd, kd[:, :-1], vd[:, :-1])
    check(masked, trunc.float().cpu(), rel=5e-3, what="sdpa masked last key == truncated")
    full = torch.randn(B, H, Sq, Skv, generator=g)
    ref = U.sdpa_math(q, k, v, attn_mask=full)
    check(ops.sdpa(qd, kd, vd, bias=full.cuda()), ref, rel=5e-3, what="sdpa full bias")


@pytest.mark.parametrize("B,HW,C,groups,pad,silu,eps", [(2, 256, 64, 32, 0, True, 1e-5), (1, 1024, 320, 32, 0, True, 1e-5),
                                                        (2, 64, 960, 32, 64, True, 1e-5), (1, 100, 1280, 32, 0, False, 1e-6),
                                                        (1, 4096, 2560, 32, 0, True, 1e-5)])
def test_groupnorm(ops, B, HW, C, groups, pad, silu, eps):
    g = torch.Generator().manual_seed(HW + C)
    x = bfr(torch.randn(B, HW, C + pad, generator=g) * 1.5 + 0.3)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xr = x[..., :C].permute(0, 2, 1).reshape(B, C, HW, 1)
    ref = U.group_norm({"n.weight": gamma, "n.bias": beta}, "n", xr, groups, eps)
    if silu:
        ref = F.silu(ref)
    out = ops.group_norm(dev(x)[..., :C], dev(gamma, torch.float32), dev(beta, torch.float32), groups, eps, silu)
    check(out.permute(0, 2, 1).reshape(B, C, HW, 1), ref, what=f"groupnorm {B,HW,C}")


@pytest.mark.parametrize("B,HW,C,groups,pad,silu,eps", [(1, 4096, 320, 32, 0, True, 1e-5), (2, 1024, 1280, 32, 64, True, 1e-5),
                                                        (8, 1024, 1280, 32, 0, False, 1e-6), (1, 64, 1280, 32, 0, True, 1e-5),
                                                        (2, 256, 1920, 32, 0, True, 1e-5), (3, 100, 64, 32, 8, True, 1e-5), (2, 256, 32, 8, 0, True, 1e-5),
                                                        (1, 1024, 960, 32, 0, True, 1e-5)])
def test_groupnorm_single_launch(ops, B, HW, C, groups, pad, silu, eps):
    """mi355x_sd_groupnorm_act: GroupNorm (+SiLU) with the (batch, group) chunk held in registers, one launch. Against fp32
    math, against the two-launch pair on the same input (statistics summed in another order: equal to the output's last bit or its
    neighbour), and refused loudly where the chunk does not fit."""
    g = torch.Generator().manual_seed(B * HW + C)
    x = bfr(torch.randn(B, HW, C, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.1
    assert ops.groupnorm_act_fits(HW, C, groups)
    xd = torch.zeros(B, HW, C + pad, device="cuda", dtype=torch.bfloat16)
    xd[..., :C] = dev(x)
    out = ops.groupnorm_act(xd[..., :C], dev(gamma, torch.float32), dev(beta, torch.float32), groups, eps, silu=silu)
    ref = F.group_norm(x.permute(0, 2, 1).reshape(B, C, HW), groups, gamma, beta, eps).reshape(B, C, HW).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    check(out, ref, what=f"groupnorm_act {B,HW,C}")
    ss = ops.groupnorm_scale_shift(xd[..., :C], dev(gamma, torch.float32), dev(beta, torch.float32), groups, eps)
    pair = ops.scale_shift_act(xd[..., :C], ss, silu)
    d = (out.float() - pair.float()).abs()
    assert (d <= 2 ** -7 * pair.float().abs() + 1e-3).all(), d.max()
    assert not ops.groupnorm_act_fits(16384, 320, 32)
    big = torch.zeros(1, 16384, 320, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(Exception):
        ops.groupnorm_act(big, dev(torch.ones(320), torch.float32), dev(torch.zeros(320), torch.float32), 32, 1e-5)


@pytest.mark.parametrize("rows,C", [(300, 64), (1000, 640), (77, 1280), (16, 2560)])
def test_layernorm(ops, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = bfr(torch.randn(rows, C, generator=g) * 2 + 0.5)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = U.layer_norm({"n.weight": gamma, "n.bias": beta}, "n", x)
    out = ops.layer_norm(dev(x), dev(gamma, torch.float32), dev(beta, torch.float32))
    check(out, ref, what=f"layernorm {rows,C}")


def test_timestep_embedding_reference_vectors(ops):
    """The reference's own hard-coded values (ppdiffusers/tests/models/test_layers_utils.py:90-115), atol 0.01."""
    t = torch.arange(128, dtype=torch.float32, device="cuda")
    t1 = ops.timestep_embedding(t, 64, downscale_freq_shift=1, flip_sin_to_cos=False).float().cpu()
    t2 = ops.timestep_embedding(t, 64, downscale_freq_shift=0, flip_sin_to_cos=True).float().cpu()
    t3 = ops.timestep_embedding(t, 64, scale=1000).float().cpu()
    g1 = torch.tensor([0.9646, 0.9804, 0.9892, 0.9615, 0.9787, 0.9882, 0.9582, 0.9769, 0.9872])
    g2 = torch.tensor([0.3019, 0.228, 0.1716, 0.3146, 0.2377, 0.179, 0.3272, 0.2474, 0.1864])
    g3 = torch.tensor([-0.9801, -0.9464, -0.9349, -0.3952, 0.8887, -0.9709, 0.5299, -0.2853, -0.9927])
    assert torch.allclose(t1[23:26, 47:50].flatten(), g1, atol=0.01)
    assert torch.allclose(t2[23:26, 47:50].flatten(), g2, atol=0.01)
    assert torch.allclose(t3[23:26, 47:50].flatten(), g3, atol=0.01)
    # the same values from the committed fixture (tests/golden/, harvested from the reference's test source)
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_known_answers.json")) as f:
        g = json.load(f)["sinusoid"]
    for case in g["cases"]:
        e = ops.timestep_embedding(t, 64, **case["kwargs"]).float().cpu()
        assert torch.allclose(e[23:26, 47:50].flatten(), torch.tensor(case["values"]), atol=g["atol"]), case["line"]
    ts = torch.tensor([981.0, 501.0, 1.0])
    ref = U.get_timestep_embedding(ts, 320, True, 0)
    out = ops.timestep_embedding(ts.cuda(), 320, flip_sin_to_cos=True, downscale_freq_shift=0).float().cpu()
    assert (out - ref).abs().max() < 2 ** -8


def test_conv_in_out_and_silu(ops):
    g = torch.Generator().manual_seed(41)
    B, H, W, C = 2, 16, 16, 64
    x = torch.randn(B, 4, H, W, generator=g)
    w = bfr(torch.randn(C, 4, 3, 3, generator=g) / 6)
    b = torch.randn(C, generator=g) * 0.1
    ref = F.conv2d(bfr(x * 0.5), w, b, padding=1)
    out = ops.conv_in3x3(x.cuda(), dev(w.permute(2, 3, 1, 0).reshape(-1, C).contiguous()), dev(b, torch.float32),
                         in_scale=torch.tensor([0.5], device="cuda"))
    check(out.reshape(B, H, W, C).permute(0, 3, 1, 2), ref, what="conv_in")
    y = bfr(torch.randn(B * H * W, C, generator=g))
    w2 = bfr(torch.randn(4, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b2 = torch.randn(4, generator=g) * 0.1
    ref2 = F.conv2d(y.reshape(B, H, W, C).permute(0, 3, 1, 2), w2, b2, padding=1)
    out2 = ops.conv_out3x3(dev(y), dev(w2.permute(0, 2, 3, 1).reshape(4, -1).contiguous()), dev(b2, torch.float32), B, H, W)
    assert out2.dtype == torch.float32
    err = ((out2.cpu() - ref2).norm() / ref2.norm()).item()
    assert err < 1e-5, err
    s = ops.silu(dev(y))
    check(s, F.silu(y), what="silu")
    for v, e in ((-100.0, 0.0), (0.0, 0.0), (20.0, 20.0)):  # tests/models/test_activations.py:24-62
        assert abs(ops.silu(torch.tensor([v], device="cuda"), torch.float32).item() - e) < 1e-4
    c = torch.tensor([0.25, -1.5], device="cuda")
    a_, b_ = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    assert torch.allclose(ops.axpby(a_.cuda(), b_.cuda(), c).cpu(), 0.25 * a_ - 1.5 * b_, atol=1e-6)


@pytest.mark.parametrize("B,C,H,W,pad", [(2, 64, 8, 8, 0), (1, 320, 5, 7, 64), (3, 8, 1, 1, 8)])
def test_add_nchw_residual(ops, B, C, H, W, pad):
    from paddlemix_amd import _lib
    g = torch.Generator().manual_seed(B + C + H)
    x = bfr(torch.randn(B * H * W, C + pad, generator=g))
    r = torch.randn(B, C, H, W, generator=g)
    xd = dev(x)
    view = xd[:, pad // 2: pad // 2 + C] if pad else xd
    _lib.check(_lib.load().mi355x_sd_add_nchw(view.data_ptr(), xd.stride(0), r.cuda().data_ptr(), B, C, H * W,
                                              torch.cuda.current_stream().cuda_stream))
    ref = x.clone()
    sl = slice(pad // 2, pad // 2 + C)
    ref[:, sl] = (x[:, sl] + r.permute(0, 2, 3, 1).reshape(B * H * W, C)).to(torch.bfloat16).float()
    assert torch.equal(xd.float().cpu(), ref)   # in place, the neighbouring channels of the wide rows untouched


def test_errors_are_loud(ops):
    from paddlemix_amd._lib import MI355XError
    a = torch.zeros(8, 12, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(16, 12, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(MI355XError, match="unsupported"):
        ops.linear(a, w)  # K not a multiple of 8
    with pytest.raises(MI355XError):
        ops.linear(torch.zeros(8, 16, dtype=torch.bfloat16), torch.zeros(16, 16, dtype=torch.bfloat16))  # CPU tensors
    # a host buffer as a GEMM call's scratch is refused by the library itself, before anything is launched (until ABI 11 a
    # process-wide binding accepted one and the next split-K launch wrote through it: a page fault on boxes without XNACK)
    from paddlemix_amd import _lib
    lib = _lib.load()
    host = torch.empty(4096 + 16, dtype=torch.uint8)
    a16 = torch.zeros(8, 16, device="cuda", dtype=torch.bfloat16)
    w16 = torch.zeros(16, 16, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(8, 16, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.mi355x_sd_linear(a16.data_ptr(), 16, w16.data_ptr(), out.data_ptr(), 16, 8, 16, 16, None, None, 0, 0, None, 0, 1.0, 0,
                              (host.data_ptr() + 15) & ~15, 1024, st)
    assert rc != 0 and b"device memory" in lib.mi355x_sd_last_error()
    rc = lib.mi355x_sd_linear(a16.data_ptr(), 16, w16.data_ptr(), out.data_ptr(), 16, 8, 16, 16, None, None, 0, 0, None, 0, 1.0, 0,
                              out.data_ptr() + 8, 1024, st)
    assert rc != 0 and b"16-byte aligned" in lib.mi355x_sd_last_error()
    assert not any(d.type == "cpu" for d, _ in ops._workspaces)


@pytest.mark.parametrize("B,H,Sq,Skv,D", [(2, 5, 256, 77, 64), (1, 4, 128, 130, 64), (1, 8, 64, 77, 40)])
def test_sdpa_ragged_kv_tail_ignores_what_follows_the_tensor(ops, B, H, Sq, Skv, D):
    """The ragged last K/V tile must not read its out-of-range rows from memory: K and V live inside one allocation whose
    bytes right behind each batch item's Skv rows (the next batch item / the padding after the last one) are NaN patterns.
    A kernel that fetched them would multiply NaN by P = 0 in the P.V MFMA and return NaN (the tile offset goes
    through the buffer descriptor's soffset, whose range check is not architecturally promised)."""
    g = torch.Generator().manual_seed(Skv + D)
    q = bfr(torch.randn(B, Sq, H, D, generator=g))
    k = bfr(torch.randn(B, Skv, H, D, generator=g))
    v = bfr(torch.randn(B, Skv, H, D, generator=g))
    ref = U.sdpa_math(q, k, v)
    pad = 200   # rows of NaN behind every batch item's keys / values
    kbuf = torch.full((B, Skv + pad, H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    vbuf = torch.full((B, Skv + pad, H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    kbuf[:, :Skv] = dev(k)
    vbuf[:, :Skv] = dev(v)
    out = ops.sdpa(dev(q), kbuf[:, :Skv], vbuf[:, :Skv])      # views: batch stride (Skv + pad) * H * D
    check(out, ref, rel=5e-3, what=f"sdpa NaN-guard {B,H,Sq,Skv,D}")


@pytest.mark.parametrize("rows,C", [(300, 64), (1000, 640), (77, 1280)])
def test_fp32_residual_forms_of_the_norms(ops, rows, C):
    """x_f32 forms (ABI v9): LayerNorm / GroupNorm(+SiLU) reading fp32 rows, the raw16 side output, cast_rows, and the
    R_F32 | OUT_F32 GEMM epilogue -- against fp32 torch math on the same fp32 inputs."""
    import torch.nn.functional as F
    from paddlemix_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 3 + 0.5).cuda()
    gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    y = torch.empty(rows, C, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.mi355x_sd_layernorm_ex(x.data_ptr(), rows, C, C, gam.data_ptr(), bet.data_ptr(), 1e-5, y.data_ptr(), C, 1, st))
    check(y, F.layer_norm(x, (C,), gam, bet, 1e-5).cpu(), what="layernorm_ex fp32 in")
    # GroupNorm over [B=1][HW=rows][C] + SiLU, with the raw 16-bit copy
    ws = torch.empty(max(1, lib.mi355x_sd_groupnorm_workspace_floats(1, rows, C)), device="cuda")
    ss = torch.empty(2 * C, device="cuda")
    raw = torch.empty(rows, C, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.mi355x_sd_groupnorm_stats_ex(x.data_ptr(), 1, rows, C, C, 32, 1e-5, gam.data_ptr(), bet.data_ptr(),
                                                ws.data_ptr(), ss.data_ptr(), 1, st))
    _lib.check(lib.mi355x_sd_scale_shift_act_ex(x.data_ptr(), 1, rows, C, C, ss.data_ptr(), 1, y.data_ptr(), C, 1,
                                                raw.data_ptr(), C, st))
    ref = F.silu(F.group_norm(x.t()[None], 32, gam, bet, 1e-5))[0].t()
    check(y, ref.cpu(), what="groupnorm_ex fp32 in")
    assert torch.equal(raw, x.to(torch.bfloat16))
    c16 = torch.empty_like(raw)
    _lib.check(lib.mi355x_sd_cast_rows(x.data_ptr(), C, c16.data_ptr(), C, rows, C, st))
    assert torch.equal(c16, raw)
    # GEMM: out(fp32) = a @ w^T + bias + R(fp32)
    N = 128
    a = bfr(torch.randn(rows, C, generator=g)).cuda().to(torch.bfloat16)
    w = bfr(torch.randn(N, C, generator=g) / C ** 0.5).cuda().to(torch.bfloat16)
    R = torch.randn(rows, N, generator=g).cuda() * 7
    bias = torch.randn(N, generator=g).cuda()
    out = torch.empty(rows, N, device="cuda")
    _lib.check(lib.mi355x_sd_linear(a.data_ptr(), C, w.data_ptr(), out.data_ptr(), N, rows, N, C, bias.data_ptr(), None, 0, 0,
                                    R.data_ptr(), N, 1.0, _lib.OUT_F32 | _lib.R_F32, *ops._workspace(a.device), st))
    want = a.float() @ w.float().t() + bias + R
    assert (out - want).abs().max() < 2e-4 * want.abs().max()     # fp32 accumulation order only: no 16-bit rounding anywhere


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(2, 16, 16, 64, 128, 1, False), (1, 32, 32, 320, 320, 1, False),
                                                      (2, 17, 15, 128, 160, 2, False), (1, 8, 8, 640, 96, 1, True),
                                                      (1, 8, 8, 1280, 1280, 1, False), (8, 32, 32, 640, 640, 1, False)])
def test_conv3x3_kb64_weight_order(ops, B, H, W, Cin, Cout, stride, up):
    """MI355X_SD_CONV_KB64: weights packed [O][Cin/64][3][3][64], K loop tap-innermost per 64-channel block -- same result as
    the [O][3][3][Cin] packing bit for bit? No: the fp32 accumulation order over K differs, so to rounding; both against the
    oracle. Covers every conv kernel (generic, pipelined 128 / 256x160, phased 256, split-K)."""
    g = torch.Generator().manual_seed(B * H + Cin + Cout + stride + 1)
    x = bfr(torch.randn(B, Cin, H, W, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g) * 0.1
    P = {"c.conv.weight": w, "c.conv.bias": bias}
    ref = U.upsample(P, "c", x) if up else (U.downsample(P, "c", x) if stride == 2 else U.conv2d(P, "c.conv", x))
    x_nhwc = dev(x.permute(0, 2, 3, 1).contiguous())
    wk = dev(w.reshape(Cout, Cin // 64, 64, 3, 3).permute(0, 1, 3, 4, 2).reshape(Cout, -1).contiguous())
    out = ops.conv3x3(x_nhwc, wk, dev(bias, torch.float32), stride=stride, upsample=up, kb64=True)
    Ho, Wo = ref.shape[2], ref.shape[3]
    check(out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref, what=f"conv3x3 kb64 {B,H,W,Cin,Cout,stride,up}")
    w0 = dev(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous())
    out0 = ops.conv3x3(x_nhwc, w0, dev(bias, torch.float32), stride=stride, upsample=up)
    assert (out.float() - out0.float()).abs().max() <= 2 * 2 ** -8 * ref.abs().max()
    with pytest.raises(Exception):
        ops.conv3x3(dev(torch.zeros(1, 4, 4, 40)), dev(torch.zeros(8, 360)), None, kb64=True)      # Cin % 64 != 0
