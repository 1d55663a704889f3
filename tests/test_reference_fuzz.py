"""Random UNet configurations inside the supported space -- block types per level, per-level head counts and layer counts, linear /
conv projections, up-cast attention, class / text_time / timestep_cond conditioning, ragged and odd latent sizes -- run three ways: the
reference's own UNet2DConditionModel (over oracle/paddle_shim.py), the oracle, and the MI355X model on the emulated device. The
reference and the oracle must agree to fp32 rounding, the device program to its 16-bit tolerance. Build container only (live
reference); `python tests/test_reference_fuzz.py <seed> <trials>` runs more."""
import random
import sys

import pytest
import torch

from oracle import reference_runner as rr
from oracle import unet_ref as U


def one_trial(trial: int, rng: random.Random):
    from paddlemix_amd.unet import UNet2DConditionModel
    from tests.abi_emulator import Emulator
    n = rng.choice([2, 3])
    boc = tuple(rng.choice([32, 64]) * (2 ** min(i, 1)) for i in range(n))
    down = tuple(rng.choice(["DownBlock2D", "CrossAttnDownBlock2D"]) for _ in range(n))
    up = tuple("CrossAttnUpBlock2D" if d == "CrossAttnDownBlock2D" else "UpBlock2D" for d in reversed(down))
    cfg = dict(block_out_channels=boc, down_block_types=down, up_block_types=up, cross_attention_dim=rng.choice([32, 64]),
               attention_head_dim=rng.choice([4, 8, (4, 8, 8)[:n]]), layers_per_block=rng.choice([1, 2, (1, 2, 1)[:n]]),
               transformer_layers_per_block=rng.choice([1, 2]), use_linear_projection=rng.choice([True, False]),
               upcast_attention=rng.choice([True, False]), norm_num_groups=32, sample_size=16,
               flip_sin_to_cos=rng.choice([True, False]), freq_shift=rng.choice([0, 1]))
    extra = rng.choice([None, "class", "text_time", "tcond"])
    kw = {}
    g = torch.Generator().manual_seed(trial)
    if extra == "class":
        cfg["num_class_embeds"] = 7
        kw["class_labels"] = torch.tensor([2, 5])
    elif extra == "text_time":
        cfg.update(addition_embed_type="text_time", addition_time_embed_dim=16, projection_class_embeddings_input_dim=16 * 6 + 32)
        kw["added_cond_kwargs"] = dict(text_embeds=torch.randn(2, 32, generator=g), time_ids=torch.tensor([[64.0, 64.0, 0.0, 0.0, 64.0, 64.0]]).repeat(2, 1))
    elif extra == "tcond":
        cfg["time_cond_proj_dim"] = 16
        kw["timestep_cond"] = torch.randn(2, 16, generator=g)
    hw = rng.choice([(8, 8), (16, 8), (12, 12), (10, 14)])
    P = U.synth_unet_params(cfg, seed=trial)
    x, enc, t = torch.randn(2, 4, *hw, generator=g), torch.randn(2, 5, cfg["cross_attention_dim"], generator=g), torch.tensor([37.0, 37.0])
    with torch.no_grad():
        ora = U.unet_forward(P, cfg, x, t, enc, **kw)
        ref = rr.from_shim(rr.build_unet(cfg, P)(rr.to_shim(x), rr.to_shim(t), rr.to_shim(enc), **rr.to_shim(kw)).sample)
        orab = U.unet_forward({k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}, cfg, x, t, enc, **kw)
    prod = UNet2DConditionModel(cfg, P, _test_backend=Emulator())(x, 37.0, enc, **kw).sample
    d_ref = float((ora - ref).abs().max() / ref.abs().max())
    d_dev = float((prod - orab).norm() / orab.norm())
    return cfg, extra, hw, d_ref, d_dev


@pytest.mark.skipif(not rr.available(), reason="/root/reference exists only in the build container")
@pytest.mark.parametrize("seed", [0, 1])
def test_random_unet_configurations(seed):
    rng = random.Random(seed)
    for trial in range(6):
        cfg, extra, hw, d_ref, d_dev = one_trial(100 * seed + trial, rng)
        assert d_ref < 5e-5 and d_dev < 2.5e-2, (cfg, extra, hw, d_ref, d_dev)


if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
        cfg, extra, hw, d_ref, d_dev = one_trial(trial, rng)
        print(f"trial {trial}: {len(cfg['block_out_channels'])} levels {[d[0] for d in cfg['down_block_types']]} {extra} {hw}: "
              f"oracle vs reference {d_ref:.1e}, device program vs oracle {d_dev:.2e}")
