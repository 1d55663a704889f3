"""-m gpu: AutoencoderKL decode path through the C ABI against the CPU oracle (oracle/vae_ref.py).
Tolerances as in test_gpu_kernels.py / test_gpu_unet.py: single ops rel-L2 <= 4e-3, whole decoder <= 2e-2 (bf16
activations between ~60 kernels; the oracle sees the same bf16-rounded weights)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_ref as R
from tests.configs import MINI_VAE, SD_VAE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from paddlemix_amd import ops as o
    o.init(0)
    return o


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def test_conv1x1_nchw(ops):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 9, 7, generator=g)
    w = (torch.randn(4, 4, generator=g) * 0.5).to(torch.bfloat16)
    b = torch.randn(4, generator=g)
    ref = F.conv2d((x / 0.18215).to(torch.bfloat16).float(), w.float()[:, :, None, None], b)
    out = ops.conv1x1_nchw(x.cuda(), w.cuda(), b.cuda(), in_scale=1 / 0.18215)
    assert _rel(out.cpu(), ref) < 1e-5
    w16 = torch.randn(16, 16, generator=g).to(torch.bfloat16)
    x16 = torch.randn(1, 16, 5, 5, generator=g)
    out = ops.conv1x1_nchw(x16.cuda(), w16.cuda())
    assert _rel(out.cpu(), F.conv2d(x16.to(torch.bfloat16).float(), w16.float()[:, :, None, None])) < 1e-5
    from paddlemix_amd._lib import MI355XError
    with pytest.raises(MI355XError):
        ops.conv1x1_nchw(torch.zeros(1, 17, 2, 2, device="cuda"), torch.zeros(4, 17, device="cuda", dtype=torch.bfloat16))


@pytest.mark.parametrize("rows,n", [(64, 64), (300, 1000), (16, 16384), (5, 4)])
def test_softmax_rows(ops, rows, n):
    g = torch.Generator().manual_seed(rows + n)
    x = torch.randn(rows, n, generator=g) * 6.0
    x[0, n // 2] = 80.0   # one dominant key
    ref = torch.softmax(x, -1)
    out = ops.softmax_rows(x.cuda())
    assert out.dtype == torch.bfloat16
    assert (out.float().cpu() - ref).abs().max() <= 2 ** -8 * ref.max() + 1e-7
    assert torch.equal(out, ops.softmax_rows(x.cuda()))


def _decode(cfg, P, z, **kw):
    from paddlemix_amd.vae import AutoencoderKL
    vae = AutoencoderKL(cfg, P, **kw)
    out = vae.decode(z.cuda()).sample
    assert torch.equal(out, vae.decode(z.cuda()).sample)   # graph replay is deterministic
    return vae, out


def test_mini_vae_decode_vs_oracle():
    from paddlemix_amd.vae import synth_decoder_params
    cfg = MINI_VAE
    P = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in synth_decoder_params(cfg, 7).items()}
    z = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(1))
    vae, out = _decode(cfg, P, z)
    ref = R.decode(P, cfg, z)
    r = _rel(out.cpu(), ref)
    print(f"mini-vae decode: rel-L2 vs oracle {r:.3e}")
    assert out.shape == ref.shape and r < 2e-2, r
    _, eager = _decode(cfg, P, z, use_graph=False)
    assert torch.equal(eager, out)
    scaled = vae.decode(z.cuda(), in_scale=1 / cfg["scaling_factor"]).sample
    assert _rel(scaled.cpu(), R.decode(P, cfg, z, scaled=True)) < 2e-2
    vae.enable_slicing()
    assert _rel(vae.decode(z.cuda()).sample, out) < 2e-2


def test_sd_vae_decoder_256px():
    """The full SD VAE decoder (49.5 M parameters, 512-wide single-head mid attention over 1024 tokens) on 32x32
    latents -> 256x256 images."""
    from paddlemix_amd.vae import synth_decoder_params
    cfg = SD_VAE
    P = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in synth_decoder_params(cfg, 11).items()}
    z = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(3))
    _, out = _decode(cfg, P, z)
    ref = R.decode(P, cfg, z)
    r = _rel(out.cpu(), ref)
    print(f"sd-vae decode 256px: rel-L2 vs oracle {r:.3e}")
    assert out.shape == (1, 3, 256, 256) and torch.isfinite(out).all() and r < 2e-2, r


# ---------------------------------------------------------------- encode path
@pytest.mark.parametrize("B,H,W,C,Cout", [(2, 16, 16, 32, 32), (1, 9, 7, 64, 16), (1, 64, 64, 128, 128)])
def test_conv3x3_stride2_bottom_right_padding(ops, B, H, W, C, Cout):
    """MI355X_SD_PAD_BR = Downsample2D(padding=0): F.pad (0, 1, 0, 1) then an unpadded stride-2 conv (resnet.py:277-279)"""
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(torch.bfloat16)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1)), w.float(), b, stride=2)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    out = ops.conv3x3(x.cuda(), wk.cuda(), b.cuda(), stride=2, pad_br=True)
    Ho, Wo = ref.shape[2], ref.shape[3]
    assert out.shape == (B * Ho * Wo, Cout)
    assert _rel(out.float().cpu().reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref) < 4e-3
    sym = ops.conv3x3(x.cuda(), wk.cuda(), b.cuda(), stride=2)            # the UNet's symmetric padding is a different conv
    ref_sym = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=2, padding=1)
    assert _rel(sym.float().cpu().reshape(B, ref_sym.shape[2], ref_sym.shape[3], Cout).permute(0, 3, 1, 2), ref_sym) < 4e-3
    from paddlemix_amd._lib import MI355XError
    with pytest.raises(MI355XError):
        ops.conv3x3(x.cuda(), wk.cuda(), b.cuda(), stride=1, pad_br=True)


def test_latent_dist_kernel(ops):
    g = torch.Generator().manual_seed(0)
    B, L, HW = 3, 4, 77
    m = torch.randn(B * HW, 2 * L, generator=g) * 20.0      # logvars beyond both clip bounds
    noise = torch.randn(B, L, HW, generator=g)
    mean, logvar, std = R.posterior(m.reshape(B, HW, 2 * L).permute(0, 2, 1))
    gm, gl, gs = ops.latent_dist(m.cuda(), B, L, noise.cuda(), out_scale=0.18215)
    assert torch.equal(gm.cpu(), mean) and torch.equal(gl.cpu(), logvar)
    assert (logvar == 20).any() and (logvar == -30).any()
    assert torch.allclose(gs.cpu(), (mean + std * noise) * 0.18215, rtol=1e-5, atol=1e-6)
    _, _, mode = ops.latent_dist(m.cuda(), B, L, None, out_scale=2.0)
    assert torch.equal(mode.cpu(), mean * 2.0)
    # wider rows than 2L (a channel slice of a larger buffer)
    wide = torch.zeros(B * HW, 16)
    wide[:, :8] = m
    assert torch.equal(ops.latent_dist(wide.cuda(), B, L, want_sample=False)[0].cpu(), mean)


def _bf(P):
    return {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}


def test_mini_vae_encode_vs_oracle():
    from paddlemix_amd.vae import AutoencoderKL, synth_vae_params
    cfg = MINI_VAE
    P = _bf(synth_vae_params(cfg, 9))
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    vae = AutoencoderKL(cfg, P)
    post = vae.encode(x.cuda()).latent_dist
    noise = torch.randn(post.mean.shape, generator=g)
    mean, logvar, sample = R.encode(P, cfg, x, noise)
    r = _rel(post.mean.cpu(), mean)
    print(f"mini-vae encode: rel-L2 vs oracle mean {r:.3e} logvar {_rel(post.logvar.cpu(), logvar):.3e}")
    assert post.mean.shape == mean.shape and r < 2e-2, r
    assert _rel(post.logvar.cpu(), logvar) < 2e-2
    got = post.sample(noise=noise.cuda())
    assert _rel(got.cpu(), sample) < 2e-2
    assert torch.equal(vae.encode(x.cuda()).latent_dist.mean, post.mean)          # graph replay is deterministic
    eager = AutoencoderKL(cfg, P, use_graph=False).encode(x.cuda()).latent_dist
    assert torch.equal(eager.mean, post.mean) and torch.equal(eager.logvar, post.logvar)
    a = post.sample(generator=torch.Generator(device="cuda").manual_seed(3))
    assert torch.equal(a, post.sample(generator=torch.Generator(device="cuda").manual_seed(3)))
    vae.enable_slicing()
    assert _rel(vae.encode(x.cuda()).latent_dist.mean, post.mean) < 2e-2
    # img2img seam: decode(encode(x).mode()) runs on the same object and keeps the size
    assert vae.decode(post.mode()).sample.shape == x.shape


def test_sd_vae_encoder_256px():
    """The full SD VAE encoder (34.2 M parameters) on a 256x256 image -> 32x32 latents."""
    from paddlemix_amd.vae import AutoencoderKL, synth_vae_params
    cfg = SD_VAE
    P = _bf(synth_vae_params(cfg, 13))
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(6)) * 2 - 1
    post = AutoencoderKL(cfg, P).encode(x.cuda()).latent_dist
    mean, logvar, _ = R.encode(P, cfg, x)
    r = _rel(post.mean.cpu(), mean)
    print(f"sd-vae encode 256px: rel-L2 vs oracle mean {r:.3e} logvar {_rel(post.logvar.cpu(), logvar):.3e}")
    assert post.mean.shape == (1, 4, 32, 32) and torch.isfinite(post.mean).all() and r < 2e-2, r
    assert _rel(post.logvar.cpu(), logvar) < 2e-2
