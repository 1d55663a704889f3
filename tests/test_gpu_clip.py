"""-m gpu: CLIP text encoder through the C ABI against the CPU oracle (oracle/clip_ref.py)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import clip_ref as R
from tests.configs import CLIP_L, MINI_CLIP
from tests.test_clip_host_logic import _ids

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from paddlemix_amd import ops as o
    o.init(0)
    return o


def _rel(a, b):
    return ((a.float().cpu() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def test_embed_tokens_and_activation(ops):
    g = torch.Generator().manual_seed(0)
    tok = torch.randn(500, 64, generator=g).to(torch.bfloat16)
    pos = torch.randn(77, 64, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, 500, (3 * 20,), generator=g, dtype=torch.int32)
    out = ops.embed_tokens(ids.cuda(), tok.cuda(), pos.cuda(), 20)
    ref = (tok.float()[ids.long()] + pos.float()[torch.arange(60) % 20]).to(torch.bfloat16)
    assert torch.equal(out.cpu(), ref)
    with pytest.raises(ValueError):
        ops.embed_tokens(torch.full((8,), 500, dtype=torch.int32).cuda(), tok.cuda(), pos.cuda(), 8)
    x = (torch.randn(1000, 64, generator=g) * 3).to(torch.bfloat16)
    xf = x.float()
    for kind, ref in (("quick_gelu", xf * torch.sigmoid(1.702 * xf)), ("gelu", F.gelu(xf)), ("silu", F.silu(xf))):
        y = ops.activation(x.cuda(), kind)
        assert (y.float().cpu() - ref).abs().max() <= 2 ** -8 * ref.abs().max() + 1e-6, kind


def test_mini_clip_vs_oracle():
    from paddlemix_amd.clip import CLIPTextModelWithProjection, synth_clip_params
    cfg = dict(MINI_CLIP, with_projection=True)
    P = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in synth_clip_params(cfg, 5).items()}
    ids = _ids(2, 77, cfg["vocab_size"], 2)
    model = CLIPTextModelWithProjection(cfg, P)
    out = model(ids.cuda(), output_hidden_states=True)
    ref = R.clip_text_forward(P, cfg, ids)
    for name, a, b in (("last", out.last_hidden_state, ref["last_hidden_state"]),
                       ("penultimate", out.hidden_states[-2], ref["hidden_states"][-2]),
                       ("pooled", out.pooler_output, ref["pooler_output"]), ("text_embeds", out.text_embeds, ref["text_embeds"])):
        assert _rel(a, b) < 1.5e-2, (name, _rel(a, b))
    out2 = model(ids.cuda())
    assert torch.equal(out2.last_hidden_state, out.last_hidden_state)
    eager = CLIPTextModelWithProjection(cfg, P, use_graph=False)(ids.cuda())
    assert torch.equal(eager.last_hidden_state, out.last_hidden_state)
    # clip_skip of the single-encoder pipelines (pipeline_stable_diffusion.py:378-391): final LayerNorm of an earlier hidden state
    import torch.nn.functional as F
    from paddlemix_amd.pipeline import StableDiffusionDenoiser
    skip = StableDiffusionDenoiser(None, None, text_encoder=model).encode_prompt(ids.cuda(), clip_skip=1)[0]
    want = F.layer_norm(ref["hidden_states"][-2], (cfg["hidden_size"],), P["text_model.final_layer_norm.weight"], P["text_model.final_layer_norm.bias"], 1e-5)
    assert _rel(skip, want) < 1.5e-2, _rel(skip, want)
    # the last hidden state through the same method is the model's own output
    assert _rel(model.text_model.final_layer_norm(out.hidden_states[-1]), ref["last_hidden_state"]) < 1.5e-2


def test_clip_l_architecture():
    """CLIP ViT-L/14 text encoder geometry (12 layers, 768 wide, 12 heads, 77 tokens): what SD-1.5 conditions on."""
    from paddlemix_amd.clip import CLIPTextModel, synth_clip_params
    cfg = CLIP_L
    P = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in synth_clip_params(cfg, 9).items()}
    ids = _ids(2, 77, cfg["vocab_size"], 2)
    out = CLIPTextModel(cfg, P)(ids.cuda())
    ref = R.clip_text_forward(P, cfg, ids)
    r = _rel(out.last_hidden_state, ref["last_hidden_state"])
    print(f"CLIP-L text encoder: rel-L2 vs oracle {r:.3e}")
    assert out.last_hidden_state.shape == (2, 77, 768) and r < 1.5e-2, r


def _vision_params(cfg, seed):
    from paddlemix_amd.clip import synth_clip_vision_params
    return {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in synth_clip_vision_params(cfg, seed).items()}


def test_mini_clip_vision_vs_oracle():
    """CLIPVisionModelWithProjection: patch GEMM (588 -> 592 padded columns) written in place behind the constant class-token
    row, maskless encoder, post_layernorm of the class rows, visual_projection"""
    from paddlemix_amd.clip import CLIPVisionModelWithProjection
    from tests.configs import MINI_CLIP_VISION
    cfg = MINI_CLIP_VISION
    P = _vision_params(cfg, 4)
    x = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(1))
    model = CLIPVisionModelWithProjection(cfg, P)
    out = model(x.cuda(), output_hidden_states=True)
    ref = R.clip_vision_forward(P, cfg, x)
    for name, a, b in (("embeddings", out.hidden_states[0], ref["hidden_states"][0]), ("last", out.last_hidden_state, ref["last_hidden_state"]),
                       ("image_embeds", out.image_embeds, ref["image_embeds"])):
        assert _rel(a, b) < 1.5e-2, (name, _rel(a, b))
    x2 = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(2))
    assert _rel(model(x2.cuda()).image_embeds, R.clip_vision_forward(P, cfg, x2)["image_embeds"]) < 1.5e-2   # graph replay, new input
    assert torch.equal(model(x.cuda()).image_embeds, out.image_embeds)
    eager = CLIPVisionModelWithProjection(cfg, P, use_graph=False)(x.cuda())
    assert torch.equal(eager.image_embeds, out.image_embeds)


def test_clip_vit_h14_vision_architecture():
    """OpenCLIP ViT-H/14 (32 layers, 1280 wide, 16 heads of 80, 257 tokens, 632 M parameters): the image encoder of the
    IP-Adapter checkpoints; its image_embeds [B, 1024] are the UNet's added_cond_kwargs["image_embeds"]."""
    from paddlemix_amd.clip import CLIPVisionModelWithProjection
    from tests.configs import CLIP_VIT_H14
    cfg = CLIP_VIT_H14
    P = _vision_params(cfg, 11)
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    out = CLIPVisionModelWithProjection(cfg, P)(x.cuda())
    ref = R.clip_vision_forward(P, cfg, x)
    r, rl = _rel(out.image_embeds, ref["image_embeds"]), _rel(out.last_hidden_state, ref["last_hidden_state"])
    print(f"CLIP ViT-H/14 vision tower: rel-L2 vs oracle image_embeds {r:.3e} last_hidden_state {rl:.3e}")
    assert out.image_embeds.shape == (1, 1024) and torch.isfinite(out.image_embeds).all() and r < 2e-2 and rl < 2e-2, (r, rl)
