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
