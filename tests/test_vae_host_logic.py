"""CPU (-m "not gpu"): the AutoencoderKL decode program (weight repacking, V^T by operand swap, value-bias folding,
buffer ping-pong, slicing) executed by the host-memory ABI emulator against the oracle."""
import pytest
import torch

from oracle import vae_ref as R
from paddlemix_amd.vae import AutoencoderKL, decoder_param_shapes, synth_decoder_params
from tests.abi_emulator import Emulator
from tests.configs import MINI_VAE


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def test_param_shapes_match_oracle():
    assert decoder_param_shapes(MINI_VAE) == R.decoder_param_shapes(MINI_VAE)
    full = dict(block_out_channels=(128, 256, 512, 512))
    s = decoder_param_shapes(full)
    assert s == R.decoder_param_shapes(full)
    assert sum(torch.Size(v).numel() for v in s.values()) == 49_490_199   # SD VAE decoder + post_quant_conv


@pytest.mark.parametrize("B,h,w", [(2, 8, 8), (1, 4, 6)])
def test_decode_program_matches_oracle(B, h, w):
    cfg = MINI_VAE
    P = synth_decoder_params(cfg, seed=7)
    z = torch.randn(B, cfg["latent_channels"], h, w, generator=torch.Generator().manual_seed(1))
    if (h * w) % 8:
        with pytest.raises(ValueError):
            AutoencoderKL(cfg, P, _test_backend=Emulator()).decode(z)
        return
    vae = AutoencoderKL(cfg, P, _test_backend=Emulator())
    out = vae.decode(z).sample
    nl = len(cfg["block_out_channels"])
    assert out.shape == (B, 3, h << (nl - 1), w << (nl - 1)) and out.dtype == torch.float32
    Pr = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}
    ref = R.decode(Pr, cfg, z)
    assert _rel(out, ref) < 2e-2, _rel(out, ref)
    # pipelines call decode(latents / scaling_factor): in_scale folds the division into the first kernel
    out_s = vae.decode(z, in_scale=1.0 / cfg["scaling_factor"], return_dict=False)[0]
    assert _rel(out_s, R.decode(Pr, cfg, z, scaled=True)) < 2e-2
    # slicing (autoencoder_kl.py:324-327): same result up to the accumulation order of differently shaped GEMMs
    vae.enable_slicing()
    assert _rel(vae.decode(z).sample, out) < 2e-2   # bf16 rounding flips propagate like in the parity bound


def test_no_post_quant_conv_and_errors():
    cfg = dict(MINI_VAE, use_post_quant_conv=False)
    P = synth_decoder_params(cfg, seed=3)
    assert "post_quant_conv.weight" not in P
    z = torch.randn(1, 4, 4, 4, generator=torch.Generator().manual_seed(2))
    vae = AutoencoderKL(cfg, P, _test_backend=Emulator())
    Pr = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}
    assert _rel(vae.decode(z).sample, R.decode(Pr, cfg, z)) < 2e-2
    with pytest.raises(NotImplementedError):
        vae.encode(z)
    with pytest.raises(ValueError):
        vae.decode(torch.zeros(1, 3, 4, 4))
    bad = dict(P)
    bad.pop("decoder.conv_out.bias")
    with pytest.raises(KeyError):
        AutoencoderKL(cfg, bad, _test_backend=Emulator())
