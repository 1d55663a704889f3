"""CPU tests of the host logic: the program paddlemix_amd.unet emits is interpreted on host memory
(tests/abi_emulator.py) and compared with the oracle.  No GPU, no HIP compute."""
import pytest
import torch

from oracle import unet_ref as U
from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params, unet_param_shapes
from tests.abi_emulator import Emulator, on_emulator
from tests.configs import MINI_XL, SD15, SDXL, TINY, UNET_VARIANTS


def _inputs(cfg, B, H, W, L=77, seed=0):
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(B, cfg.get("in_channels", 4), H, W, generator=g)
    cross = cfg["cross_attention_dim"]
    cross = cross[0] if isinstance(cross, (tuple, list)) else cross
    enc = torch.randn(B, L, cross, generator=g)
    added = None
    if cfg.get("addition_embed_type") == "text_time":
        td = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
        added = dict(text_embeds=torch.randn(B, td, generator=g),
                     time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(B, 1))
    return sample, enc, added


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("cfg,B,H,W,L", [(TINY, 2, 16, 16, 7), (MINI_XL, 1, 16, 16, 77), (TINY, 1, 8, 24, 5)])
def test_program_matches_oracle(cfg, B, H, W, L):
    P = synth_unet_params(cfg, seed=1234)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}  # device weights are bf16
    sample, enc, added = _inputs(cfg, B, H, W, L)
    t = 501
    ref = U.unet_forward(Pb, cfg, sample, t, enc, added_cond_kwargs=added)
    emu = Emulator()
    model = on_emulator(UNet2DConditionModel, cfg, P, backend=emu)
    out = model(sample, t, enc, added_cond_kwargs=added, return_dict=False)[0]
    assert out.shape == ref.shape and out.dtype == torch.float32
    # bf16 activations between ops: a few 1e-3 of relative error is the rounding floor
    assert _rel(out, ref) < 2e-2, _rel(out, ref)
    # second call reuses the plan and is deterministic
    out2 = model(sample, t, enc, added_cond_kwargs=added).sample
    assert torch.equal(out, out2)
    assert len(model._plans) == 1
    # optional program variant: LayerNorms folded into their consuming projections (row_stats + linear_ln)
    assert "linear_ln" not in emu.calls
    emu2 = Emulator()
    folded = on_emulator(UNet2DConditionModel, cfg, P, fold_layernorm=True, backend=emu2)
    out3 = folded(sample, t, enc, added_cond_kwargs=added).sample
    assert "linear_ln" in emu2.calls and _rel(out3, ref) < 2e-2


def test_forward_after_staged_steps_takes_the_sample_as_is():
    """A host that drives a geometry through the staged API with scale_model_input folded into conv_in (stage_inputs(in_scale=...),
    what bench.py's timed loop does) leaves its factor in the plan; forward() on the same geometry must not inherit it -- the bs-8
    parity leg of the round-5 bench line read 0.68 through exactly that."""
    cfg = TINY
    P = synth_unet_params(cfg, seed=1234)
    sample, enc, added = _inputs(cfg, 2, 16, 16, 7)
    model = on_emulator(UNet2DConditionModel, cfg, P)
    ref = model(sample, 501, enc, added_cond_kwargs=added).sample.clone()
    plan = model._get_plan(2, 16, 16, 7, False, False, 0)
    assert len(model._plans) == 1                          # the plan forward() built and will use again
    model.stage_inputs(plan, sample, 501, enc, added, in_scale=0.25)
    model._run_eager(plan)
    assert _rel(plan.out, ref) > 1e-2                      # the staged call did scale its input
    assert torch.equal(model(sample, 501, enc, added_cond_kwargs=added).sample, ref)


@pytest.mark.parametrize("cfg,B,H,W,L", [(TINY, 2, 16, 16, 7), (MINI_XL, 1, 16, 16, 77)])
def test_fp32_residual_stream_program(cfg, B, H, W, L):
    """residual_dtype="fp32": resnet outputs / transformer hidden state / skip slots are fp32 rows, 16-bit values exist only
    as MFMA operands. Same program semantics, fewer rounding points: closer to the oracle than the 16-bit stream, and no
    fp32 row is ever handed to a GEMM / conv as an operand (the builder asserts it)."""
    P = synth_unet_params(cfg, seed=1234)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, added = _inputs(cfg, B, H, W, L)
    ref = U.unet_forward(Pb, cfg, sample, 501, enc, added_cond_kwargs=added)
    e16, e32 = Emulator(), Emulator()
    o16 = on_emulator(UNet2DConditionModel, cfg, P, backend=e16)(sample, 501, enc, added_cond_kwargs=added).sample
    o32 = on_emulator(UNet2DConditionModel, cfg, P, residual_dtype="fp32", backend=e32)(sample, 501, enc, added_cond_kwargs=added).sample
    r16, r32 = _rel(o16, ref), _rel(o32, ref)
    assert r32 < 0.8 * r16 and r32 < 1e-2, (r16, r32)
    # same number of contractions; the extra launches are the operand casts in front of the down / upsampling convs
    # (small GroupNorms of the 16-bit stream run as one launch; fp32 rows take the statistics + apply pair)
    pair = [c2 for c in e16.calls for c2 in (("gn_stats", "scale_shift_act") if c == "gn_fused" else (c,))]
    assert [c for c in e32.calls if c != "cast_rows"] == pair and "cast_rows" in e32.calls and "gn_fused" in e16.calls
    with pytest.raises(NotImplementedError):
        on_emulator(UNet2DConditionModel, cfg, P, residual_dtype="fp32", fold_layernorm=True)


@pytest.mark.parametrize("name", sorted(UNET_VARIANTS))
def test_config_variants_match_oracle(name):
    """the configuration switches the reference's model tests flip (tests/configs.py UNET_VARIANTS)"""
    cfg = UNET_VARIANTS[name]
    P = synth_unet_params(cfg, seed=7)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, _ = _inputs(cfg, 2, 16, 16, 7)
    ref = U.unet_forward(Pb, cfg, sample, 333, enc)
    out = on_emulator(UNet2DConditionModel, cfg, P)(sample, 333, enc, return_dict=False)[0]
    assert out.shape == ref.shape and _rel(out, ref) < 2e-2, _rel(out, ref)


CLASS_CASES = {
    "embedding": (dict(num_class_embeds=10), lambda g, B, c: torch.randint(0, 10, (B,), generator=g)),
    "timestep": (dict(class_embed_type="timestep"), lambda g, B, c: torch.rand(B, generator=g) * 900),
    "identity": (dict(class_embed_type="identity"), lambda g, B, c: torch.randn(B, c["block_out_channels"][0] * 4, generator=g)),
    "projection": (dict(class_embed_type="projection", projection_class_embeddings_input_dim=32),
                   lambda g, B, c: torch.randn(B, 32, generator=g)),
    "simple_projection": (dict(class_embed_type="simple_projection", projection_class_embeddings_input_dim=32),
                          lambda g, B, c: torch.randn(B, 32, generator=g)),
}


@pytest.mark.parametrize("kind", sorted(CLASS_CASES))
@pytest.mark.parametrize("concat", [False, True])
def test_class_embeddings_match_oracle(kind, concat):
    """class_labels / class_embed_type / num_class_embeds / class_embeddings_concat (unet_2d_condition.py:354-382, 953-975;
    the reference's test_model_with_simple_projection / test_model_with_class_embeddings_concat)"""
    extra, make = CLASS_CASES[kind]
    cfg = dict(TINY, class_embeddings_concat=concat, **extra)
    P = synth_unet_params(cfg, seed=3)
    assert list(unet_param_shapes(cfg).items()) == list(U.unet_param_shapes(cfg).items())
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, _ = _inputs(cfg, 2, 16, 16, 7)
    labels = make(torch.Generator().manual_seed(5), 2, cfg)
    ref = U.unet_forward(Pb, cfg, sample, 333, enc, class_labels=labels)
    model = on_emulator(UNet2DConditionModel, cfg, P)
    out = model(sample, 333, enc, class_labels=labels, return_dict=False)[0]
    assert _rel(out, ref) < 2e-2, _rel(out, ref)
    # the class embedding matters (different labels -> different output), and is required
    other = make(torch.Generator().manual_seed(6), 2, cfg)
    assert not torch.equal(model(sample, 333, enc, class_labels=other).sample, out)
    with pytest.raises(ValueError, match="class_labels should be provided"):
        model(sample, 333, enc)
    if "num_class_embeds" in extra:   # nn.Embedding's range check on host labels (the gather kernel trusts device indices)
        with pytest.raises(IndexError, match="class_labels must lie in"):
            model(sample, 333, enc, class_labels=torch.tensor([0, cfg["num_class_embeds"]]))


def test_timestep_cond_matches_oracle():
    """time_cond_proj_dim / timestep_cond: the guidance-scale embedding of LCM-distilled UNets (embeddings.py:265-285,
    unet_2d_condition.py:953)"""
    cfg = dict(TINY, time_cond_proj_dim=32)
    P = synth_unet_params(cfg, seed=9)
    assert list(unet_param_shapes(cfg).items()) == list(U.unet_param_shapes(cfg).items())
    assert "time_embedding.cond_proj.bias" not in P and P["time_embedding.cond_proj.weight"].shape == (32, 64)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, _ = _inputs(cfg, 2, 16, 16, 7)
    w = torch.randn(2, 32, generator=torch.Generator().manual_seed(1))
    model = on_emulator(UNet2DConditionModel, cfg, P)
    out = model(sample, 200, enc, timestep_cond=w).sample
    assert _rel(out, U.unet_forward(Pb, cfg, sample, 200, enc, timestep_cond=w)) < 2e-2
    out0 = model(sample, 200, enc).sample     # no condition: the projection adds nothing
    assert _rel(out0, U.unet_forward(Pb, cfg, sample, 200, enc)) < 2e-2 and not torch.equal(out0, out)
    with pytest.raises(ValueError, match="timestep_cond"):
        on_emulator(UNet2DConditionModel, TINY, synth_unet_params(TINY, seed=9))(sample, 200, enc, timestep_cond=w)


def test_ip_adapter_matches_oracle():
    """encoder_hid_dim_type="ip_image_proj": ImageProjection of `image_embeds` (embeddings.py:507-527, unet_2d_condition.py:
    1054-1061) and the image-token half of every cross-attention (IPAdapterAttnProcessor, attention_processor.py:1793-1901)"""
    cfg = dict(TINY, encoder_hid_dim_type="ip_image_proj", encoder_hid_dim=48)
    assert list(unet_param_shapes(cfg).items()) == list(U.unet_param_shapes(cfg).items())
    P = synth_unet_params(cfg, seed=3)
    b = "down_blocks.1.attentions.0.transformer_blocks.0.attn2.processor"
    assert P[b + ".to_k_ip.weight"].shape == P[b.replace(".processor", "") + ".to_k.weight"].shape and (b + ".to_k_ip.bias") not in P       # Paddle [cross, C], no bias
    assert P["encoder_hid_proj.image_embeds.weight"].shape == (48, 4 * 64) and P["encoder_hid_proj.norm.weight"].shape == (64,)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, _ = _inputs(cfg, 2, 16, 16, 7)
    g = torch.Generator().manual_seed(2)
    img, img2 = torch.randn(2, 48, generator=g), torch.randn(2, 48, generator=g)
    model = on_emulator(UNet2DConditionModel, cfg, P)
    outs = {}
    for sc in (1.0, 0.6, 0.0):
        model.set_ip_adapter_scale(sc)
        outs[sc] = model(sample, 10, enc, added_cond_kwargs={"image_embeds": img}).sample
        ref = U.unet_forward(Pb, cfg, sample, 10, enc, added_cond_kwargs={"image_embeds": img}, ip_adapter_scale=sc)
        assert _rel(outs[sc], ref) < 2e-2, (sc, _rel(outs[sc], ref))
    # scale 0 launches no image-token attention: exactly the UNet without the adapter on the same weights
    base = on_emulator(UNet2DConditionModel, TINY, {k: v for k, v in P.items() if "_ip." not in k and "encoder_hid_proj" not in k})
    assert torch.equal(outs[0.0], base(sample, 10, enc).sample)
    assert _rel(outs[1.0], outs[0.0]) > 1e-2                     # the image prompt matters ...
    model.set_ip_adapter_scale(1.0)
    assert not torch.equal(model(sample, 10, enc, added_cond_kwargs={"image_embeds": img2}).sample, outs[1.0])   # ... and which one
    with pytest.raises(ValueError, match="image_embeds"):
        model(sample, 10, enc)
    with pytest.raises(ValueError, match="image_embeds of shape"):
        model(sample, 10, enc, added_cond_kwargs={"image_embeds": img[:, :40]})
    with pytest.raises(NotImplementedError):
        model(sample, 10, enc, added_cond_kwargs={"image_embeds": img}, encoder_attention_mask=torch.ones(2, 7))
    with pytest.raises(ValueError):
        base.set_ip_adapter_scale(0.5)
    with pytest.raises(ValueError, match="encoder_hid_dim"):
        unet_param_shapes(dict(TINY, encoder_hid_dim_type="ip_image_proj"))
    with pytest.raises(NotImplementedError):
        unet_param_shapes(dict(TINY, encoder_hid_dim_type="text_proj", encoder_hid_dim=32))


def test_controlnet_matches_oracle_and_feeds_the_unet():
    """ControlNetModel.forward (controlnet.py:671-877): conditioning embedding + encoder half + zero convolutions, and its
    outputs as the UNet's down_block_additional_residuals / mid_block_additional_residual (unet_2d_condition.py:1121-1155)"""
    from paddlemix_amd.unet import ControlNetModel, controlnet_param_shapes, synth_controlnet_params
    for cfg in (TINY, MINI_XL):
        assert list(controlnet_param_shapes(cfg).items()) == list(U.controlnet_param_shapes(cfg).items())
    # SD-1.5 ControlNet: 361 279 120 parameters (the published sd-controlnet checkpoints)
    assert sum(torch.Size(v).numel() for v in controlnet_param_shapes(SD15).values()) == 361_279_120
    cfg = dict(TINY, controlnet_conditioning_channel_order="bgr")
    P = synth_controlnet_params(cfg, seed=3)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, _ = _inputs(TINY, 2, 16, 16, 7)
    cond = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(4))
    net = on_emulator(ControlNetModel, cfg, P)
    assert net.config.conditioning_embedding_out_channels == (16, 32, 96, 256) and net.config.in_channels == 4
    for sc, gm in ((1.0, False), (0.6, True)):
        out = net(sample, 20, enc, cond, conditioning_scale=sc, guess_mode=gm)
        downs, mid = U.controlnet_forward(Pb, cfg, sample, 20, enc, cond, sc, gm)
        assert len(out.down_block_res_samples) == len(downs) == 6
        for a, b in zip(out.down_block_res_samples + (out.mid_block_res_sample,), downs + (mid,)):
            assert a.shape == b.shape and a.dtype == torch.float32 and _rel(a, b) < 2e-2, _rel(a, b)
    d2, m2 = net(sample, 20, enc, cond, return_dict=False)
    Pu = synth_unet_params(TINY, seed=9)
    unet = on_emulator(UNet2DConditionModel, TINY, Pu)
    got = unet(sample, 20, enc, down_block_additional_residuals=d2, mid_block_additional_residual=m2).sample
    Pub = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in Pu.items()}
    rd, rm = U.controlnet_forward(Pb, cfg, sample, 20, enc, cond)
    ref = U.unet_forward(Pub, TINY, sample, 20, enc, down_block_additional_residuals=rd, mid_block_additional_residual=rm)
    assert _rel(got, ref) < 2e-2 and _rel(got, unet(sample, 20, enc).sample) > 1e-2
    with pytest.raises(ValueError, match="controlnet_cond of shape"):
        net(sample, 20, enc, cond[:, :, :64, :64])
    with pytest.raises(NotImplementedError):
        net(sample, 20, enc, cond, conditioning_scale=[1.0] * 7)
    with pytest.raises(NotImplementedError):
        on_emulator(ControlNetModel, dict(TINY, global_pool_conditions=True), P)
    bad = dict(P)
    bad.pop("controlnet_mid_block.weight")
    with pytest.raises(KeyError):
        on_emulator(ControlNetModel, cfg, bad)


def test_param_inventory_matches_oracle():
    for cfg in (TINY, MINI_XL, SD15, SDXL):
        a, b = unet_param_shapes(cfg), U.unet_param_shapes(cfg)
        assert list(a.items()) == list(b.items())


def test_synth_params_match_oracle_generator():
    a, b = synth_unet_params(TINY, 7), U.synth_unet_params(TINY, 7)
    assert all(torch.equal(a[k], b[k]) for k in b)


def test_errors_mirror_reference():
    P = synth_unet_params(MINI_XL)
    model = on_emulator(UNet2DConditionModel, MINI_XL, P)
    sample, enc, added = _inputs(MINI_XL, 1, 8, 8)
    with pytest.raises(ValueError, match="text_embeds"):
        model(sample, 1, enc, added_cond_kwargs={})
    with pytest.raises(ValueError, match="time_ids"):
        model(sample, 1, enc, added_cond_kwargs={"text_embeds": added["text_embeds"]})
    with pytest.raises(KeyError):
        on_emulator(UNet2DConditionModel, MINI_XL, {k: v for k, v in P.items() if k != "conv_in.weight"})
    bad = dict(P)
    bad["time_embedding.linear_1.weight"] = bad["time_embedding.linear_1.weight"].t()
    with pytest.raises(ValueError, match="Paddle layout"):
        on_emulator(UNet2DConditionModel, MINI_XL, bad)
    with pytest.raises(NotImplementedError):   # config fields outside the implemented path fail loudly at construction
        on_emulator(UNet2DConditionModel, dict(MINI_XL, dual_cross_attention=True), P)
    with pytest.raises(ValueError, match="requires `projection_class_embeddings_input_dim`"):   # unet_2d_condition.py:363-366
        on_emulator(UNet2DConditionModel, dict(TINY, class_embed_type="projection"), P)


def test_no_fallback_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from paddlemix_amd._lib import MI355XError
    with pytest.raises(MI355XError):
        UNet2DConditionModel(TINY, synth_unet_params(TINY))


def test_encoder_attention_mask_semantics():
    """test_model_xattn_mask (ppdiffusers/tests/models/test_models_unet_2d_condition.py:486-515): keep-all == no mask;
    masking the last token == truncating it; and the masked forward matches the oracle."""
    cfg = TINY
    P = synth_unet_params(cfg, seed=1234)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, _ = _inputs(cfg, 2, 8, 8, L=7)
    model = on_emulator(UNet2DConditionModel, cfg, P)
    none = model(sample, 10, enc).sample
    keep = model(sample, 10, enc, encoder_attention_mask=torch.ones(2, 7)).sample
    assert torch.allclose(none, keep, rtol=1e-3, atol=1e-5)
    m = torch.ones(2, 7)
    m[:, -1] = 0
    masked = model(sample, 10, enc, encoder_attention_mask=m).sample
    trunc = model(sample, 10, enc[:, :-1]).sample
    assert torch.allclose(masked, trunc, rtol=1e-3, atol=1e-3)
    ref = U.unet_forward(Pb, cfg, sample, 10, enc, encoder_attention_mask=m)
    assert _rel(masked, ref) < 2e-2


@pytest.mark.parametrize("head_dim64", [False, True], ids=["d16", "d64-folded-scale"])
def test_self_attention_mask_semantics(head_dim64):
    """`attention_mask` of UNet2DConditionModel.forward (unet_2d_condition.py:916-923): a keep/discard mask over the LATENT tokens,
    turned into a -10000 bias and added to the scores of every self-attention (BasicTransformerBlock.attn1). It has to match the
    token count of every attention level (Attention.prepare_attention_mask pads a wrong length by target_length and the add then
    fails on the shapes, attention_processor.py:616-622): the reference's tiny test geometry, where all attention runs at one
    resolution, is the usable case. head_dim 64: the softmax scale is folded into to_q and the mask goes in base-2 units."""
    cfg = dict(TINY, attention_head_dim=2) if head_dim64 else TINY       # 128 channels / 2 heads = 64
    P = synth_unet_params(cfg, seed=1234)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, _ = _inputs(cfg, 2, 16, 16, L=7)                        # attention at 8 x 8 = 64 latent tokens
    model = on_emulator(UNet2DConditionModel, cfg, P)
    if head_dim64:
        assert model._log2_blocks
    none = model(sample, 10, enc).sample
    keep = model(sample, 10, enc, attention_mask=torch.ones(2, 64)).sample
    assert torch.allclose(none, keep, rtol=1e-3, atol=1e-4)
    g = torch.Generator().manual_seed(3)
    m = (torch.rand(2, 64, generator=g) > 0.4).float()
    m[:, 0] = 1
    masked = model(sample, 10, enc, attention_mask=m).sample
    assert _rel(masked, none) > 1e-2                                       # the mask does something
    ref = U.unet_forward(Pb, cfg, sample, 10, enc, attention_mask=m)
    assert _rel(masked, ref) < 2e-2
    both = model(sample, 10, enc, attention_mask=m, encoder_attention_mask=torch.ones(2, 7)).sample   # both masks together
    assert torch.allclose(both, masked, rtol=1e-3, atol=1e-4)
    with pytest.raises(ValueError, match="key tokens"):                   # wrong length: the reference fails on the shapes
        model(sample, 10, enc, attention_mask=torch.ones(2, 63))
    with pytest.raises(ValueError, match="batch"):
        model(sample, 10, enc, attention_mask=torch.ones(64))
    # a UNet with attention at several resolutions cannot take one mask length (SD-1.5 / SDXL: no pipeline passes it)
    multi = on_emulator(UNet2DConditionModel, MINI_XL, synth_unet_params(MINI_XL, seed=1))
    s2, e2, a2 = _inputs(MINI_XL, 1, 16, 16, L=7)
    with pytest.raises(ValueError, match="key tokens"):
        multi(s2, 10, e2, added_cond_kwargs=a2, attention_mask=torch.ones(1, 64))


def test_seam_objects_refuse_cpu_tensors():
    """B2 / B3 seams (paddlemix_amd/attention.py) have no CPU fallback either."""
    from paddlemix_amd.attention import Attention, scaled_dot_product_attention_
    from paddlemix_amd._lib import MI355XError
    q = torch.zeros(1, 8, 2, 16)
    with pytest.raises(ValueError, match="attention_op"):
        scaled_dot_product_attention_(q, q, q, attention_op="flash")
    if not torch.cuda.is_available():
        with pytest.raises((MI355XError, ValueError)):
            scaled_dot_product_attention_(q, q, q)
        P = {f"to_{n}.weight": torch.zeros(32, 32) for n in "qkv"}
        P.update({"to_out.0.weight": torch.zeros(32, 32), "to_out.0.bias": torch.zeros(32)})
        with pytest.raises((MI355XError, ValueError)):
            Attention(P, heads=2, device="cpu")(torch.zeros(1, 8, 32))


def _controlnet_residuals(cfg, B, H, W, seed=3):
    """one residual per skip tensor (conv_in, every down resnet/attention output, every downsampler output) + one for
    the mid block, shaped like a ControlNet's outputs"""
    from paddlemix_amd.unet import normalize_config
    g = torch.Generator().manual_seed(seed)
    c = normalize_config(cfg)
    boc = c["block_out_channels"]
    shapes, h, w = [(boc[0], H, W)], H, W
    for i in range(len(boc)):
        shapes += [(boc[i], h, w)] * c["layers_per_block"][i]
        if i != len(boc) - 1:
            h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
            shapes.append((boc[i], h, w))
    down = [0.3 * torch.randn(B, cc, hh, ww, generator=g) for cc, hh, ww in shapes]
    mid = 0.3 * torch.randn(B, boc[-1], h, w, generator=g)
    return down, mid


@pytest.mark.parametrize("cfg", [TINY, MINI_XL])
def test_controlnet_residual_inputs(cfg):
    """down_block_additional_residuals / mid_block_additional_residual (unet_2d_condition.py:1121-1155)."""
    P = synth_unet_params(cfg, seed=1234)
    Pb = {k: v.to(torch.bfloat16).float() if v.dim() > 1 else v for k, v in P.items()}
    sample, enc, added = _inputs(cfg, 2, 16, 16, 7)
    model = on_emulator(UNet2DConditionModel, cfg, P)
    plain = model(sample, 300, enc, added_cond_kwargs=added).sample
    down, mid = _controlnet_residuals(cfg, 2, 16, 16)
    ref = U.unet_forward(Pb, cfg, sample, 300, enc, added_cond_kwargs=added, down_block_additional_residuals=down,
                         mid_block_additional_residual=mid)
    out = model(sample, 300, enc, added_cond_kwargs=added, down_block_additional_residuals=down,
                mid_block_additional_residual=mid).sample
    assert _rel(out, ref) < 2e-2 and _rel(out, plain) > 5e-2     # matches the oracle and is not a no-op
    assert torch.equal(model(sample, 300, enc, added_cond_kwargs=added).sample, plain)   # the plain plan is untouched
    with pytest.raises(NotImplementedError):
        model(sample, 300, enc, added_cond_kwargs=added, down_block_additional_residuals=down)
    with pytest.raises(ValueError):
        model(sample, 300, enc, added_cond_kwargs=added, down_block_additional_residuals=down[:-1],
              mid_block_additional_residual=mid)


@pytest.mark.parametrize("key,val", [("attention_type", "gated"), ("conv_in_kernel", 5), ("conv_out_kernel", 1), ("dropout", 0.1),
                                     ("mid_block_only_cross_attention", True), ("resnet_out_scale_factor", 2.0)])
def test_semantics_changing_config_keys_are_refused(key, val):
    """ctor arguments of the reference that change the arithmetic (unet_2d_condition.py:172-226) and are only built at their
    default must not be accepted silently"""
    with pytest.raises(NotImplementedError, match=key):
        unet_param_shapes(dict(TINY, **{key: val}))
    unet_param_shapes(dict(TINY, attention_type="default", conv_in_kernel=3, dropout=0.0))   # defaults spelled out are fine


def test_latent_sizes_that_are_not_multiples_of_the_up_factor():
    """forward_upsample_size (unet_2d_condition.py:900-906, :1165-1169; pinned in the oracle by the reference case
    unet_mini_xl_odd_size): a skip of odd size 2h - 1 makes the upsampler interpolate to that size. The Python planner materialises the
    cropped nearest upsample with strided row copies and runs a plain 3x3 conv; the C++ planner (mi355x_sd_unet_plan) refuses such sizes."""
    P = synth_unet_params(TINY, seed=1)
    Pb = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}
    g = torch.Generator().manual_seed(0)
    for hw in ((15, 15), (15, 18), (16, 13)):
        x, enc = torch.randn(2, 4, *hw, generator=g), torch.randn(2, 7, 64, generator=g)
        out = on_emulator(UNet2DConditionModel, TINY, P)(x, 10.0, enc).sample
        ref = U.unet_forward(Pb, TINY, x, torch.tensor([10.0, 10.0]), enc)
        assert out.shape == ref.shape == x.shape and ((out - ref).norm() / ref.norm()).item() < 2e-2, hw
    # three levels: 18 -> 9 -> 5, then 5 -> 9 (cropped) and 9 -> 18 (exact x2, still folded into the conv gather)
    m = on_emulator(UNet2DConditionModel, MINI_XL, synth_unet_params(MINI_XL, seed=1))
    td = MINI_XL["projection_class_embeddings_input_dim"] - 6 * MINI_XL["addition_time_embed_dim"]
    added = dict(text_embeds=torch.randn(1, td, generator=g), time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]))
    x, enc = torch.randn(1, 4, 18, 18, generator=g), torch.randn(1, 7, 128, generator=g)
    out = m(x, 10.0, enc, added_cond_kwargs=added).sample
    ref = U.unet_forward({k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in synth_unet_params(MINI_XL, seed=1).items()}, MINI_XL, x,
                         torch.tensor([10.0]), enc, added_cond_kwargs=added)
    assert ((out - ref).norm() / ref.norm()).item() < 2e-2
    names = [fn.__name__ for fn, *_ in list(m._plans.values())[-1].prog]
    assert names.count("mi355x_sd_copy_rows") == 5 * 4 - 2      # 5 source rows x (dy, dx); destination row 9 (source row 4, dy 1) is cropped


def test_bench_cpu_leg_respects_the_containers_cpu_quota(tmp_path):
    """bench.usable_cpus: the affinity mask capped by the cgroup CPU quota (v2 cpu.max, v1 cfs quota / period); no quota = the mask."""
    import os
    import bench
    mask = len(os.sched_getaffinity(0))
    assert bench.usable_cpus(str(tmp_path)) == mask                       # no cgroup files at all
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench.usable_cpus(str(tmp_path)) == mask
    (tmp_path / "cpu.max").write_text("150000 100000\n")                    # 1.5 CPUs -> 2 threads at most
    assert bench.usable_cpus(str(tmp_path)) == min(mask, 2)
    (tmp_path / "cpu.max").write_text("garbage\n")
    (tmp_path / "cpu").mkdir()
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.usable_cpus(str(tmp_path)) == mask                       # v1, unlimited
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("100000\n")
    assert bench.usable_cpus(str(tmp_path)) == 1


def test_wrappers_called_with_cpu_tensors_do_not_bind_a_host_workspace():
    """ops.linear(CPU tensors) must raise BEFORE anything reaches the library: it used to bind a 32-MB host buffer as the split-K
    workspace on its way to the argument check, and the next launch that took split-K slices without rebinding wrote through it."""
    import pytest
    import torch
    from paddlemix_amd import _lib, ops
    calls = []
    real = _lib.load

    class Spy:
        def __getattr__(self, name):
            calls.append(name)
            return getattr(real(), name)

    ops._lib.load, saved = (lambda: Spy()), ops._lib.load
    try:
        with pytest.raises(_lib.MI355XError):
            ops.linear(torch.zeros(8, 16, dtype=torch.bfloat16), torch.zeros(16, 16, dtype=torch.bfloat16))
        with pytest.raises(_lib.MI355XError):
            ops.linear_ex(torch.zeros(8, 16, dtype=torch.bfloat16), torch.zeros(16, 16, dtype=torch.bfloat16))
    finally:
        ops._lib.load = saved
    assert not any(c.startswith("mi355x_sd_linear") for c in calls) and not any(d.type == "cpu" for d, _ in ops._workspaces)

