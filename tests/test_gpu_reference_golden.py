"""-m gpu: the HIP path against THE REFERENCE'S OWN OUTPUTS.

tests/golden/reference_modules/*.npz hold what the reference's unmodified modules computed (ppdiffusers model files executed over
oracle/paddle_shim.py in the build container, scripts/make_reference_golden.py). Here the MI355X models take the same parameters
and inputs (tests/reference_cases.py regenerates them from their seeds; only the oracle side of a case runs, to rebuild the inputs)
and are compared with those committed reference outputs directly. Stated tolerance: rel-L2 <= 2e-2 for 16-bit weights and
activations with fp32 accumulation (the weights are rounded to the element type on load; the reference ran them in fp32)."""

import numpy as np
import pytest
import torch

from tests import reference_cases as RC

pytestmark = pytest.mark.gpu


def _gold(name):
    g = np.load(RC.golden_path(name))
    return {k: torch.from_numpy(g[k]) for k in g.files}


def _rel(a, b):
    return float((a.float().cpu() - b.float()).norm() / b.float().norm())


def _c(x):
    if isinstance(x, dict):
        return {k: _c(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_c(v) for v in x)
    return x.cuda() if torch.is_tensor(x) else x


UNET_CASES = ["unet_tiny", "unet_mini_xl", "unet_sd2_layout", "unet_tiny_masks", "unet_mini_xl_encoder_mask", "unet_class_embeds",
              "unet_class_projection", "unet_controlnet_residuals", "unet_mini_xl_odd_size"]


def _row(v, b):
    if isinstance(v, dict):
        return {k: _row(x, b) for k, x in v.items()}
    if isinstance(v, (tuple, list)):
        return tuple(_row(x, b) for x in v)
    return v[b:b + 1]


@pytest.mark.parametrize("name", UNET_CASES)
def test_unet_against_the_reference_output(name, backend_kw=None):
    from paddlemix_amd.unet import UNet2DConditionModel
    i = RC.CASES[name](False)["inputs"]
    model = UNet2DConditionModel(i["cfg"], i["P"], **(backend_kw or {}))
    dev = (lambda x: x) if backend_kw else _c
    # the cases put two different timesteps in one batch (the reference broadcasts per sample); the device program takes one: by row
    rows = [model(dev(i["x"][b:b + 1]), float(i["t"][b]), dev(i["enc"][b:b + 1]), **dev(_row(i["kw"], b))).sample for b in range(i["x"].shape[0])]
    r = _rel(torch.cat(rows), _gold(name)["sample"])
    print(f"{name}: rel-L2 vs the reference's output {r:.3e}")
    assert r < 2e-2, r


def test_controlnet_against_the_reference_output():
    from paddlemix_amd.unet import ControlNetModel
    name = "controlnet_bgr_guess_mode"
    i, gold = RC.CASES[name](False)["inputs"], _gold(name)
    out = ControlNetModel(i["cfg"], i["P"])(_c(i["x"]), float(i["t"][0]), _c(i["enc"]), _c(i["cond"]), conditioning_scale=i["scale"],
                                           guess_mode=i["guess"], return_dict=False)
    downs, mid = out[0], out[1]
    for k, d in enumerate(downs):
        assert _rel(d, gold[f"down{k}"]) < 2e-2, (k, _rel(d, gold[f"down{k}"]))
    assert _rel(mid, gold["mid"]) < 2e-2


def test_dit_sd3_against_the_reference_output():
    from paddlemix_amd.dit import DiTTransformer2DModel
    from paddlemix_amd.sd3 import SD3Transformer2DModel
    i = RC.CASES["dit_mini"](False)["inputs"]
    rows = [DiTTransformer2DModel(i["cfg"], i["P"])(_c(i["x"][b:b + 1]), timestep=_c(i["t"][b:b + 1]), class_labels=_c(i["y"][b:b + 1])).sample
            for b in range(2)]
    r = _rel(torch.cat(rows), _gold("dit_mini")["sample"])
    print(f"dit_mini: {r:.3e}")
    assert r < 2e-2, r
    for name in ("sd3_mini", "sd3_mini_trained_norm_bias"):
        i = RC.CASES[name](False)["inputs"]
        m = SD3Transformer2DModel(i["cfg"], i["P"])
        rows = [m(_c(i["x"][b:b + 1]), _c(i["enc"][b:b + 1]), _c(i["pooled"][b:b + 1]), float(i["t"][b])).sample for b in range(2)]
        r = _rel(torch.cat(rows), _gold(name)["sample"])
        print(f"{name}: {r:.3e}")
        assert r < 2e-2, (name, r)


def test_vae_and_text_encoders_against_the_reference_output():
    from paddlemix_amd.clip import CLIPTextModelWithProjection, CLIPVisionModelWithProjection
    from paddlemix_amd.t5 import T5EncoderModel
    from paddlemix_amd.vae import AutoencoderKL
    i, gold = RC.CASES["vae_mini"](False)["inputs"], _gold("vae_mini")
    vae = AutoencoderKL(i["cfg"], i["P"])
    assert _rel(vae.decode(_c(i["z"])).sample, gold["decode"]) < 2e-2
    post = vae.encode(_c(i["img"])).latent_dist
    assert _rel(post.mean, gold["encode_mean"]) < 2e-2 and _rel(post.logvar, gold["encode_logvar"]) < 2e-2
    for name in ("clip_text_quick_gelu", "clip_text_gelu"):
        i, gold = RC.CASES[name](False)["inputs"], _gold(name)
        out = CLIPTextModelWithProjection(i["cfg"], i["P"])(_c(i["ids"]), output_hidden_states=True)
        assert _rel(out.last_hidden_state, gold["last_hidden_state"]) < 1.5e-2 and _rel(out.hidden_states[-2], gold["penultimate"]) < 1.5e-2
        assert _rel(out.text_embeds, gold["text_embeds"]) < 2e-2
    i, gold = RC.CASES["clip_vision"](False)["inputs"], _gold("clip_vision")
    out = CLIPVisionModelWithProjection(i["cfg"], i["P"])(_c(i["px"]), output_hidden_states=True)
    assert _rel(out.image_embeds, gold["image_embeds"]) < 2e-2 and _rel(out.hidden_states[-2], gold["penultimate"]) < 1.5e-2
    i, gold = RC.CASES["t5_encoder"](False)["inputs"], _gold("t5_encoder")
    # T5's unscaled attention logits are the sharpest of the encoders; with weights that are not 16-bit representable (the reference
    # ran them in fp32, the device rounds them on load) the host-memory emulator already sits at 1.8e-2 from the reference here
    assert _rel(T5EncoderModel(i["cfg"], i["P"])(_c(i["ids"])).last_hidden_state, gold["last_hidden_state"]) < 3e-2
