"""-m gpu: the reference's inner seams: B3 scaled_dot_product_attention_ and B2 AttnProcessor, MI355X-backed,
against the oracle's restatement of the same call sites."""
import math

import pytest
import torch

from oracle import unet_ref as U
from oracle import vae_ref as V

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float().cpu() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def bfr(t):
    return t.to(torch.bfloat16).float()


def test_sdpa_seam_matches_math_branch():
    """paddle_patch.py:414-529: [B,S,h,d] layout, additive / boolean masks, is_causal, scale, op dispatch errors."""
    from paddlemix_amd.attention import scaled_dot_product_attention_ as sdpa
    g = torch.Generator().manual_seed(0)
    B, Sq, Skv, H, D = 2, 96, 77, 4, 40
    q, k, v = (bfr(torch.randn(B, s, H, D, generator=g)) for s in (Sq, Skv, Skv))
    mask = torch.randn(B, H, Sq, Skv, generator=g)
    out = sdpa(q.cuda(), k.cuda(), v.cuda(), attn_mask=mask.cuda())
    assert out.shape == (B, Sq, H, D) and out.dtype == torch.float32
    assert _rel(out, U.sdpa_math(q, k, v, mask)) < 5e-3
    keep = torch.rand(B, 1, 1, Skv, generator=g) > 0.3
    keep[..., 0] = True
    ref = U.sdpa_math(q, k, v, torch.zeros(B, 1, 1, Skv).masked_fill(~keep, float("-inf")))
    assert _rel(sdpa(q.cuda(), k.cuda(), v.cuda(), attn_mask=keep.cuda(), attention_op="mi355x"), ref) < 5e-3
    qs = bfr(torch.randn(1, 64, 2, 64, generator=g))
    causal = torch.triu(torch.full((64, 64), float("-inf")), 1)[None, None]
    assert _rel(sdpa(qs.cuda(), qs.cuda(), qs.cuda(), is_causal=True, scale=0.2), U.sdpa_math(qs, qs, qs, causal, 0.2)) < 5e-3
    with pytest.raises(ValueError, match="attention_op"):
        sdpa(q.cuda(), k.cuda(), v.cuda(), attention_op="cutlass")
    with pytest.raises(NotImplementedError):
        sdpa(q.cuda(), k.cuda(), v.cuda(), dropout_p=0.1)
    with pytest.raises(ValueError, match="broadcastable"):
        sdpa(q.cuda(), k.cuda(), v.cuda(), attn_mask=torch.zeros(B, H, Sq, Skv + 1).cuda())


def _attn_params(C, cross, g, bias=False, gn=False):
    P = {"a.to_q.weight": torch.randn(C, C, generator=g) / math.sqrt(C),
         "a.to_k.weight": torch.randn(cross, C, generator=g) / math.sqrt(cross),
         "a.to_v.weight": torch.randn(cross, C, generator=g) / math.sqrt(cross),
         "a.to_out.0.weight": torch.randn(C, C, generator=g) / math.sqrt(C),
         "a.to_out.0.bias": torch.randn(C, generator=g) * 0.1}
    if bias:
        for n in ("to_q", "to_k", "to_v"):
            P[f"a.{n}.bias"] = torch.randn(C, generator=g) * 0.1
    if gn:
        P["a.group_norm.weight"] = 1 + 0.1 * torch.randn(C, generator=g)
        P["a.group_norm.bias"] = 0.1 * torch.randn(C, generator=g)
    return {k: (bfr(v) if v.dim() > 1 else v) for k, v in P.items()}


@pytest.mark.parametrize("cross,masked", [(None, False), (96, False), (96, True)])
def test_attn_processor_seam(cross, masked):
    """B2 on a 3-D input, self / cross / masked-cross: attention_processor.py:673-735 (test_special_attn_proc style:
    the processor object is installed on the module and called through it)."""
    from paddlemix_amd.attention import Attention, MI355XAttnProcessor
    g = torch.Generator().manual_seed(3)
    B, S, C, heads, L = 2, 64, 128, 4, 10
    P = _attn_params(C, cross or C, g)
    x = bfr(torch.randn(B, S, C, generator=g))
    enc = bfr(torch.randn(B, L, cross, generator=g)) if cross else None
    mask = None
    if masked:   # additive [B, 1, L] encoder mask as built at unet_2d_condition.py:921-927
        keep = torch.ones(B, L)
        keep[:, -3:] = 0
        mask = ((1 - keep) * -10000.0)[:, None, :]
    attn = Attention({k[2:]: v for k, v in P.items()}, heads=heads)
    calls = []

    class Counting(MI355XAttnProcessor):
        def __call__(self, *a, **k):
            calls.append(1)
            return super().__call__(*a, **k)
    attn.set_processor(Counting())
    out = attn(x.cuda(), encoder_hidden_states=None if enc is None else enc.cuda(),
               attention_mask=None if mask is None else mask.cuda())
    ref = U.attention(P, "a", x, heads, enc, mask)
    assert calls == [1] and out.shape == ref.shape and _rel(out, ref) < 6e-3, _rel(out, ref)


def test_attn_processor_4d_groupnorm_residual():
    """The deprecated-attn-block form used by the VAE mid block: 4-D input, GroupNorm, q/k/v biases, residual."""
    from paddlemix_amd.attention import Attention
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 64, 8, 8
    P = _attn_params(C, C, g, bias=True, gn=True)
    x = bfr(torch.randn(B, C, H, W, generator=g))
    attn = Attention({k[2:]: v for k, v in P.items()}, heads=1, norm_num_groups=32, eps=1e-6, residual_connection=True)
    out = attn(x.cuda())
    ref = V.mid_attention(P, "a", x, 32, 1e-6)
    assert out.shape == x.shape and _rel(out, ref) < 6e-3, _rel(out, ref)


@pytest.mark.parametrize("context_pre_only", [False, True])
def test_joint_attn_processor_seam(context_pre_only):
    """B2, SD3 form (VERDICT r5 missing #4): an object with JointAttnProcessor2_5's contract -- __call__(attn, hidden_states,
    encoder_hidden_states, attention_mask) -> (hidden_states, encoder_hidden_states), attention_processor.py:909-983 -- installed on a
    JointAttention weight holder and called through it; the expected values are that call site's own sequence of operations
    (six projections, concat on the token axis, one attention over the joint sequence, split, two output projections) in fp32 math.
    S_img + S_txt = 218 is ragged against every tile; context_pre_only: the last MMDiT block returns the raw text rows."""
    from paddlemix_amd.attention import JointAttention, MI355XJointAttnProcessor
    g = torch.Generator().manual_seed(9)
    B, S1, S2, C, heads = 2, 64, 154, 128, 2
    names = ["to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0"] + ([] if context_pre_only else ["to_add_out"])
    P = {}
    for n in names:
        P[n + ".weight"] = bfr(torch.randn(C, C, generator=g) / math.sqrt(C))   # Paddle layout [in, out]
        P[n + ".bias"] = torch.randn(C, generator=g) * 0.1
    x = bfr(torch.randn(B, S1, C, generator=g))
    c = bfr(torch.randn(B, S2, C, generator=g))
    attn = JointAttention(P, heads=heads, context_pre_only=context_pre_only)
    calls = []

    class Counting(MI355XJointAttnProcessor):
        def __call__(self, *a, **k):
            calls.append(1)
            return super().__call__(*a, **k)
    attn.set_processor(Counting())
    out, enc = attn(x.cuda(), encoder_hidden_states=c.cuda())

    def lin(t, n):
        return t @ P[n + ".weight"] + P[n + ".bias"]
    q = torch.cat([lin(x, "to_q"), lin(c, "add_q_proj")], 1)
    k = torch.cat([lin(x, "to_k"), lin(c, "add_k_proj")], 1)
    v = torch.cat([lin(x, "to_v"), lin(c, "add_v_proj")], 1)
    d = C // heads
    o = U.sdpa_math(q.reshape(B, -1, heads, d), k.reshape(B, -1, heads, d), v.reshape(B, -1, heads, d), None).reshape(B, S1 + S2, C)
    ref_out = lin(o[:, :S1], "to_out.0")
    ref_enc = o[:, S1:] if context_pre_only else lin(o[:, S1:], "to_add_out")
    assert calls == [1] and out.shape == ref_out.shape and enc.shape == ref_enc.shape
    assert out.dtype == torch.float32 and enc.dtype == torch.float32
    assert _rel(out, ref_out) < 6e-3 and _rel(enc, ref_enc) < 6e-3, (_rel(out, ref_out), _rel(enc, ref_enc))
    with pytest.raises(NotImplementedError):
        attn(x.cuda(), encoder_hidden_states=c.cuda(), attention_mask=torch.zeros(B, 1, S1 + S2).cuda())
    with pytest.raises(ValueError):
        attn(x.cuda())
