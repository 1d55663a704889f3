"""CPU (-m "not gpu"): the AutoencoderKL decode program (weight repacking, V^T by operand swap, value-bias folding,
buffer ping-pong, slicing) executed by the host-memory ABI emulator against the oracle."""
import pytest
import torch

from oracle import vae_ref as R
from paddlemix_amd.vae import (AutoencoderKL, decoder_param_shapes, encoder_param_shapes, synth_decoder_params,
                               synth_vae_params)
from tests.abi_emulator import Emulator, on_emulator
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
            on_emulator(AutoencoderKL, cfg, P).decode(z)
        return
    vae = on_emulator(AutoencoderKL, cfg, P)
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
    vae = on_emulator(AutoencoderKL, cfg, P)
    Pr = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}
    assert _rel(vae.decode(z).sample, R.decode(Pr, cfg, z)) < 2e-2
    from paddlemix_amd._lib import MI355XError
    with pytest.raises(MI355XError):          # decode-only parameters: encode refuses
        vae.encode(torch.zeros(1, 3, 16, 16))
    with pytest.raises(NotImplementedError):
        vae.enable_tiling()
    with pytest.raises(ValueError):
        vae.decode(torch.zeros(1, 3, 4, 4))
    bad = dict(P)
    bad.pop("decoder.conv_out.bias")
    with pytest.raises(KeyError):
        on_emulator(AutoencoderKL, cfg, bad)


def _bf(P):
    return {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}


def test_encoder_param_shapes_match_oracle():
    assert encoder_param_shapes(MINI_VAE) == R.encoder_param_shapes(MINI_VAE)
    assert list(encoder_param_shapes(MINI_VAE)) == list(R.encoder_param_shapes(MINI_VAE))
    full = dict(block_out_channels=(128, 256, 512, 512))
    s = encoder_param_shapes(full)
    assert s == R.encoder_param_shapes(full)
    # SD VAE encoder 34 163 592 + quant_conv 72: with the decoder's 49 490 199 the published 83 653 863
    assert sum(torch.Size(v).numel() for v in s.values()) == 83_653_863 - 49_490_199


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 16, 32)])
def test_encode_program_matches_oracle(B, H, W):
    cfg = MINI_VAE
    P = synth_vae_params(cfg, seed=9)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    vae = on_emulator(AutoencoderKL, cfg, P)
    assert vae.has_encoder
    post = vae.encode(x).latent_dist
    n = len(cfg["block_out_channels"])
    assert post.mean.shape == (B, 4, H >> (n - 1), W >> (n - 1)) and post.mean.dtype == torch.float32
    noise = torch.randn(post.mean.shape, generator=g)
    mean, logvar, sample = R.encode(_bf(P), cfg, x, noise)
    assert _rel(post.mode(), mean) < 2e-2, _rel(post.mode(), mean)
    assert _rel(post.logvar, logvar) < 2e-2
    got = post.sample(noise=noise)
    assert _rel(got, sample) < 2e-2
    assert torch.allclose(got, post.mean + post.std * noise, atol=1e-5)
    # the pipelines' `* scaling_factor` folded into the launch; a generator draws the noise otherwise
    assert torch.allclose(post.sample(noise=noise, out_scale=cfg["scaling_factor"]), got * cfg["scaling_factor"], atol=1e-6)
    a = post.sample(generator=torch.Generator().manual_seed(5))
    assert torch.equal(a, post.sample(generator=torch.Generator().manual_seed(5))) and not torch.equal(a, got)
    assert vae.encode(x, return_dict=False)[0].mean.equal(post.mean)
    # slicing: one image per launch sequence (autoencoder_kl.py:271-273)
    vae.enable_slicing()
    assert _rel(vae.encode(x).latent_dist.mean, post.mean) < 2e-2
    # encode -> decode round trip keeps the image size
    assert vae.decode(post.mode()).sample.shape == x.shape


def test_encode_without_quant_conv_logvar_clip_and_errors():
    cfg = dict(MINI_VAE, use_quant_conv=False)
    P = synth_vae_params(cfg, seed=2)
    assert "quant_conv.weight" not in P
    P["encoder.conv_out.bias"] = P["encoder.conv_out.bias"].clone()
    P["encoder.conv_out.bias"][4] = 50.0      # logvar channel 0 far above the clip
    P["encoder.conv_out.bias"][5] = -70.0     # channel 1 far below
    x = torch.rand(1, 3, 16, 16, generator=torch.Generator().manual_seed(1)) * 2 - 1
    vae = on_emulator(AutoencoderKL, cfg, P)
    post = vae.encode(x).latent_dist
    mean, logvar, _ = R.encode(_bf(P), cfg, x)
    assert torch.all(post.logvar[:, 0] == 20.0) and torch.all(post.logvar[:, 1] == -30.0)
    assert _rel(post.logvar, logvar) < 2e-2 and _rel(post.mean, mean) < 2e-2
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(1, 4, 16, 16))          # wrong channel count
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(1, 3, 18, 16))          # not a multiple of 2^(levels-1)
    with pytest.raises(ValueError):
        post.sample(noise=torch.zeros(1, 4, 3, 3))
    bad = dict(P)
    bad.pop("encoder.conv_norm_out.bias")
    with pytest.raises(KeyError):
        on_emulator(AutoencoderKL, cfg, bad)
