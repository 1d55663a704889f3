"""Random SD3 / DiT / AutoencoderKL configurations run three ways -- the reference's own model class (over oracle/paddle_shim.py), the oracle,
the MI355X model on the emulated device -- to look for deviations outside the committed cases. Build container only.

    python scripts/fuzz_against_reference.py <seed> <trials>        (UNet and scheduler fuzz: tests/test_reference_fuzz.py)
"""
import sys, random, math, torch, traceback
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import reference_runner as rr, sd3_ref as S3, dit_ref as D, vae_ref as V
from tests.abi_emulator import Emulator, on_emulator
from paddlemix_amd.sd3 import SD3Transformer2DModel
from paddlemix_amd.dit import DiTTransformer2DModel
from paddlemix_amd.vae import AutoencoderKL
from tests import reference_cases as RC
rr.install()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
bf = lambda P: {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
mx = lambda a, b: float((a - b).abs().max() / b.abs().max())
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    g = torch.Generator().manual_seed(trial)
    try:
        kind = rng.choice(["sd3", "dit", "vae"])
        if kind == "sd3":
            heads, hd = rng.choice([(2, 32), (4, 32), (2, 64)])
            cfg = dict(sample_size=32, patch_size=2, in_channels=rng.choice([4, 16]), num_layers=rng.choice([1, 2, 3]), attention_head_dim=hd, num_attention_heads=heads,
                       caption_projection_dim=heads * hd, joint_attention_dim=rng.choice([32, 64]), pooled_projection_dim=rng.choice([32, 64]), out_channels=None, pos_embed_max_size=48)
            cfg["out_channels"] = cfg["in_channels"]
            P = S3.synth_sd3_params(cfg, seed=trial)
            hw = rng.choice([(16, 16), (8, 24), (12, 20)]); L = rng.choice([5, 9, 20])
            x, enc, pooled, t = torch.randn(1, cfg["in_channels"], *hw, generator=g), torch.randn(1, L, cfg["joint_attention_dim"], generator=g), torch.randn(1, cfg["pooled_projection_dim"], generator=g), torch.tensor([421.0])
            with torch.no_grad():
                ora = S3.sd3_forward(P, cfg, x, enc, pooled, t)
                net = rr.ref_module("transformer_sd3").SD3Transformer2DModel(**cfg); net.eval(); rr.load_params(net, P, computed=RC.SD3_COMPUTED + RC.SD3_OPTIONAL)
                ref = rr.from_shim(net(rr.to_shim(x), encoder_hidden_states=rr.to_shim(enc), pooled_projections=rr.to_shim(pooled), timestep=rr.to_shim(t)).sample)
                orab = S3.sd3_forward(bf(P), cfg, x, enc, pooled, t)
            prod = on_emulator(SD3Transformer2DModel, cfg, P)(x, enc, pooled, 421.0).sample
        elif kind == "dit":
            heads, hd = rng.choice([(2, 32), (4, 32), (2, 64)])
            cfg = dict(sample_size=rng.choice([16, 32]), num_layers=rng.choice([1, 2, 3]), patch_size=2, attention_head_dim=hd, num_attention_heads=heads, in_channels=4,
                       out_channels=rng.choice([4, 8]), num_embeds_ada_norm=rng.choice([10, 100]))
            P = D.synth_dit_params(cfg, seed=trial)
            side = rng.choice([16, 24, 32])
            x, t, y = torch.randn(1, 4, side, side, generator=g), torch.tensor([rng.randrange(1000)]), torch.tensor([rng.randrange(cfg["num_embeds_ada_norm"] + 1)])
            with torch.no_grad():
                ora = D.dit_forward(P, cfg, x, t, y)
                full = D.normalize_config(cfg); full.pop("inner_dim")
                net = rr.ref_module("transformer_2d").Transformer2DModel(**full); net.eval(); rr.load_params(net, P)
                ref = rr.from_shim(net(rr.to_shim(x), timestep=rr.to_shim(t), class_labels=rr.to_shim(y)).sample)
                orab = D.dit_forward(bf(P), cfg, x, t, y)
            prod = on_emulator(DiTTransformer2DModel, cfg, P)(x, timestep=t, class_labels=y).sample
        else:
            nlev = rng.choice([2, 3, 4]); boc = tuple(rng.choice([32, 64]) for _ in range(nlev))
            cfg = dict(in_channels=3, out_channels=3, latent_channels=rng.choice([4, 16]), block_out_channels=boc, layers_per_block=rng.choice([1, 2]), norm_num_groups=32,
                       scaling_factor=0.18215, use_post_quant_conv=rng.choice([True, False]), use_quant_conv=rng.choice([True, False]))
            P = V.synth_decoder_params(cfg, seed=trial); P.update(RC._synth(V.encoder_param_shapes(cfg), trial + 1))
            f = 2 ** (nlev - 1); zh, zw = rng.choice([(8, 8), (4, 8), (8, 12)])
            z, img = torch.randn(1, cfg["latent_channels"], zh, zw, generator=g), torch.randn(1, 3, zh * f, zw * f, generator=g)
            with torch.no_grad():
                ora = V.decode(P, cfg, z); ora_m = V.encode(P, cfg, img)[0]
                full = V.normalize_config(cfg); full.update(down_block_types=("DownEncoderBlock2D",) * nlev, up_block_types=("UpDecoderBlock2D",) * nlev)
                net = rr.ref_module("autoencoder_kl").AutoencoderKL(**{k: v for k, v in full.items()}); net.eval(); rr.load_params(net, P)
                ref = rr.from_shim(net.decode(rr.to_shim(z)).sample); ref_m = rr.from_shim(net.encode(rr.to_shim(img)).latent_dist.mean)
                orab = V.decode(bf(P), cfg, z); orab_m = V.encode(bf(P), cfg, img)[0]
            vae = on_emulator(AutoencoderKL, cfg, P)
            prod = vae.decode(z).sample
            d3 = rel(vae.encode(img).latent_dist.mean, orab_m); d4 = mx(ora_m, ref_m)
            if d3 > 2.5e-2 or d4 > 5e-5: print("   vae encode: product-vs-oracle %.2e oracle-vs-reference %.1e  <<<<" % (d3, d4)); bad += 1
        d1, d2 = mx(ora, ref), rel(prod, orab)
        flag = "" if d1 < 5e-5 and d2 < 2.5e-2 else "   <<<<<<<<"; bad += bool(flag)
        print(f"trial {trial} {kind} {cfg if flag else ''}: oracle-vs-reference {d1:.1e}  product-vs-oracle {d2:.2e}{flag}")
    except Exception as e:
        bad += 1; print(f"trial {trial} {kind} {cfg}: {type(e).__name__}: {str(e)[:300]}")
print("bad:", bad)
