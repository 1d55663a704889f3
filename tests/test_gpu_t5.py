"""-m gpu: T5 encoder (SD3's third text encoder) through the C ABI against the CPU oracle (oracle/t5_ref.py)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import t5_ref as R
from tests.configs import MINI_T5, T5_XXL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from paddlemix_amd import ops as o
    o.init(0)
    return o


def _rel(a, b):
    return ((a.float().cpu() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("rows,C", [(77, 4096), (300, 64), (1000, 1024), (5, 2048)])
def test_rmsnorm(ops, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 3 + 1).to(torch.bfloat16)
    w = 1 + 0.1 * torch.randn(C, generator=g)
    xf = x.float()
    ref = w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    out = ops.rms_norm(x.cuda(), w.cuda(), 1e-6)
    assert _rel(out, ref) < 4e-3 and (out.float().cpu() - ref).abs().max() <= 2 ** -7 * ref.abs().max()


def test_gated_activation(ops):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(333, 2 * 256, generator=g) * 2).to(torch.bfloat16)
    xf = x.float()
    for kind, act in (("gelu_new", lambda t: F.gelu(t, approximate="tanh")), ("gelu", F.gelu), ("silu", F.silu)):
        ref = act(xf[:, :256]) * xf[:, 256:]
        out = ops.gated_activation(x.cuda(), kind)
        assert (out.float().cpu() - ref).abs().max() <= 2 ** -7 * ref.abs().max() + 1e-6, kind


def test_mini_t5_vs_oracle():
    from paddlemix_amd.t5 import T5EncoderModel, synth_t5_params
    cfg = MINI_T5
    P = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 and "relative_attention_bias" not in k else v)
         for k, v in synth_t5_params(cfg, 11).items()}
    ids = torch.randint(0, cfg["vocab_size"], (2, 77), generator=torch.Generator().manual_seed(2))
    model = T5EncoderModel(cfg, P)
    out = model(ids.cuda()).last_hidden_state
    ref = R.t5_encoder_forward(P, cfg, ids)
    assert _rel(out, ref) < 1.5e-2, _rel(out, ref)
    assert torch.equal(model(ids.cuda()).last_hidden_state, out)
    assert torch.equal(T5EncoderModel(cfg, P, use_graph=False)(ids.cuda()).last_hidden_state, out)


def test_t5_xxl_geometry_two_layers():
    """T5 v1.1 XXL layer geometry (d_model 4096, 64 heads x 64, d_ff 10240) on 77 tokens, 2 of the 24 layers, small vocab."""
    from paddlemix_amd.t5 import T5EncoderModel, synth_t5_params
    cfg = dict(T5_XXL, num_layers=2, vocab_size=1000)
    P = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 and "relative_attention_bias" not in k else v)
         for k, v in synth_t5_params(cfg, 5).items()}
    ids = torch.randint(0, 1000, (2, 77), generator=torch.Generator().manual_seed(3))
    out = T5EncoderModel(cfg, P)(ids.cuda()).last_hidden_state
    ref = R.t5_encoder_forward(P, cfg, ids)
    r = _rel(out, ref)
    print(f"T5-XXL geometry (2 layers): rel-L2 vs oracle {r:.3e}")
    assert out.shape == (2, 77, 4096) and r < 2e-2, r   # whole-model bar (unscaled logits over K = 4096 are sharp)
