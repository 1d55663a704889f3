"""CPU: the T5 encoder program (RMS norm, unscaled attention with the shared relative position bias, gated-gelu FF)
interpreted by the ABI emulator against the oracle."""
import pytest
import torch

from oracle import t5_ref as R
from paddlemix_amd.t5 import T5EncoderModel, relative_position_bucket, synth_t5_params, t5_param_shapes
from tests.abi_emulator import Emulator, on_emulator
from tests.configs import MINI_T5, T5_XXL


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def test_shapes_buckets_and_counts():
    assert t5_param_shapes(MINI_T5) == R.t5_param_shapes(MINI_T5)
    n = sum(torch.Size(s).numel() for s in t5_param_shapes(T5_XXL).values())
    assert n == 4_762_310_656   # T5 v1.1 XXL encoder (the 4.7 B text encoder of SD3)
    rp = torch.arange(-200, 201)
    assert torch.equal(relative_position_bucket(rp), R.relative_position_bucket(rp))
    b = relative_position_bucket(rp)
    assert b.min() == 0 and b.max() == 31 and b[200] == 0 and b[201] == 17 and b[199] == 1


@pytest.mark.parametrize("B,S", [(2, 24), (1, 77)])
def test_program_matches_oracle(B, S):
    cfg = MINI_T5
    P = synth_t5_params(cfg, seed=11)
    Pb = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 and "relative_attention_bias" not in k else v) for k, v in P.items()}
    ids = torch.randint(0, cfg["vocab_size"], (B, S), generator=torch.Generator().manual_seed(2))
    ref = R.t5_encoder_forward(Pb, cfg, ids)
    model = on_emulator(T5EncoderModel, cfg, P)
    out = model(ids)
    assert out.last_hidden_state.shape == (B, S, cfg["d_model"])
    assert _rel(out.last_hidden_state, ref) < 1.5e-2, _rel(out.last_hidden_state, ref)
    assert torch.equal(model(ids, return_dict=False)[0], out[0])
    # tied embedding name is accepted (T5Model._tied_weights_keys)
    P2 = {("encoder.embed_tokens.weight" if k == "shared.weight" else k): v for k, v in P.items()}
    assert torch.equal(on_emulator(T5EncoderModel, cfg, P2)(ids)[0], out[0])


def test_errors():
    P = synth_t5_params(MINI_T5, seed=1)
    m = on_emulator(T5EncoderModel, MINI_T5, P)
    with pytest.raises(ValueError):
        m(torch.full((1, 4), 500))
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 4, dtype=torch.long), attention_mask=torch.ones(1, 4))
    with pytest.raises(NotImplementedError):
        on_emulator(T5EncoderModel, dict(MINI_T5, feed_forward_proj="relu"), P)
    with pytest.raises(KeyError):
        on_emulator(T5EncoderModel, MINI_T5, {k: v for k, v in P.items() if "final_layer_norm" not in k})
