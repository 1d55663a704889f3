"""CPU: host logic of seam B1 in the C ABI (csrc/unet_exec.hip) -- config parsing, parameter table, weight packing and the plan
-- against the Python builder (paddlemix_amd/unet.py), without launching anything. The packed image must equal the Python
model's packed tensors bit for bit, the parameter table must equal unet_param_shapes, the program must have the same number of
launches. (The launches themselves are compared on the GPU: tests/test_gpu_cexec.py.)"""
import ctypes

import pytest
import torch

from paddlemix_amd import _lib
from paddlemix_amd.cexec import UNetHandle
from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params, unet_param_shapes
from tests.abi_emulator import Emulator, on_emulator
from tests.configs import MINI_XL, SD15, SDXL, TINY


# the inputs of forward that belong to the config (round 6: the C++ planner builds them too -- unet_2d_condition.py:404-441, 953-975)
CLASS_CONFIGS = {
    "lcm-cond": dict(TINY, time_cond_proj_dim=16),
    "class-table": dict(TINY, num_class_embeds=10),
    "class-timestep": dict(MINI_XL, class_embed_type="timestep"),       # on top of text_time: both add into the embedding
    "class-identity": dict(TINY, class_embed_type="identity"),
    "class-projection": dict(TINY, class_embed_type="projection", projection_class_embeddings_input_dim=24, time_cond_proj_dim=8),
    "class-simple": dict(TINY, class_embed_type="simple_projection", projection_class_embeddings_input_dim=24),
    # class_embeddings_concat: the blocks see [emb | class_emb] (blocks_time_embed_dim = 2 x time_embed_dim)
    "class-table-concat": dict(TINY, num_class_embeds=6, class_embeddings_concat=True),
    "class-identity-concat": dict(TINY, class_embed_type="identity", class_embeddings_concat=True),
    "class-timestep-concat": dict(TINY, class_embed_type="timestep", class_embeddings_concat=True, time_cond_proj_dim=16),
    # IP-Adapter: ImageProjection + to_k_ip / to_v_ip on every cross-attention (unet_2d_condition.py:1054-1061)
    "ip-adapter": dict(TINY, encoder_hid_dim_type="ip_image_proj", encoder_hid_dim=96),
    "ip-adapter-xl": dict(MINI_XL, encoder_hid_dim_type="ip_image_proj", encoder_hid_dim=64, ip_adapter_num_tokens=16),
}


@pytest.mark.parametrize("cfg", [TINY, MINI_XL, SD15, SDXL] + list(CLASS_CONFIGS.values()), ids=["tiny", "mini-xl", "sd15", "sdxl"] + list(CLASS_CONFIGS))
def test_parameter_table_equals_python_table(cfg):
    hd = UNetHandle(cfg)
    got = hd.param_shapes()
    want = unet_param_shapes(cfg)
    assert list(got) == list(want)                      # names AND construction order
    assert all(tuple(got[k]) == tuple(want[k]) for k in want)


@pytest.mark.parametrize("cfg,rd", [(TINY, None), (MINI_XL, None), (MINI_XL, "fp32")] + [(c, None) for c in CLASS_CONFIGS.values()],
                         ids=["tiny", "mini-xl", "mini-xl-f32resid"] + list(CLASS_CONFIGS))
def test_packed_weights_and_plan_equal_python_builder(cfg, rd):
    P = synth_unet_params(cfg, seed=5)
    hd = UNetHandle(cfg, residual_dtype=rd)
    hd.load(P)
    image = hd.pack()
    model = on_emulator(UNet2DConditionModel, cfg, P, residual_dtype=rd)
    checked = 0
    for key, t in model.w.items():
        got = hd.packed_tensor(image, key)
        if t.dtype == torch.float32:
            assert torch.equal(got, t.reshape(-1)), key
        else:
            assert got.shape == t.shape and torch.equal(got.view(torch.int16), t.view(torch.int16)), key
        checked += 1
    assert checked == len(model.w) and checked > 30
    # plan: same number of kernel launches as the Python program for the same geometry (the Python plan inserts the
    # time_ids sinusoid at its first call; the handle knows the widths from the config)
    B, H, W, L = 2, 16, 16, 7
    hd.attach(image)             # planning reads no device memory: a host image is enough to lay the program out
    nbytes = hd.plan(B, H, W, L)
    plan = model._get_plan(B, H, W, L)
    n_py = len(plan.prog) + (1 if cfg.get("addition_embed_type") == "text_time" else 0)
    assert hd.num_launches() == n_py
    assert nbytes > 0 and nbytes % 256 == 0


@pytest.mark.parametrize("H,W", [(18, 18), (20, 10), (9, 14)])
def test_odd_latent_sizes_plan_the_python_planners_launches(H, W):
    """`forward_upsample_size` (unet_2d_condition.py:900-906, 1165-1169): latents that are not multiples of 2^(levels - 1) -- the
    C++ planner refused them through round 5; it now materialises the cropped nearest-x2 tensor with the same row copies as the
    Python planner (bit-identity on the device: tests/test_gpu_cexec.py)"""
    cfg = MINI_XL
    P = synth_unet_params(cfg, seed=5)
    hd = UNetHandle(cfg)
    hd.load(P)
    image = hd.pack()
    hd.attach(image)
    assert hd.plan(1, H, W, 77) > 0
    model = on_emulator(UNet2DConditionModel, cfg, P)
    assert hd.num_launches() == len(model._get_plan(1, H, W, 77).prog) + 1     # (+ the time_ids sinusoid the Python plan inserts at its first call)
    n = ctypes.c_size_t()
    assert _lib.load().mi355x_sd_unet_plan(hd.h, 1, 2, 2, 77, ctypes.byref(n)) != 0   # nothing left at the lowest of three levels


def test_weight_life_cycle_after_the_host_image_is_released():
    """attach / finalize release the packed host image; every later call has to work from the cached size, never repack
    from the (consumed) fp32 tensors. pack -> attach -> weight_bytes / attach again / pack: sizes stay, pack says why it cannot,
    a moved weight buffer drops the plan (its launches hold absolute weight addresses)."""
    lib = _lib.load()
    P = synth_unet_params(TINY, seed=5)
    hd = UNetHandle(TINY)
    hd.load(P)
    n = hd.weight_bytes()
    image = hd.pack()
    hd.attach(image)
    assert hd.weight_bytes() == n                       # (used to re-run the packer on emptied tensors: SIGSEGV)
    hd.plan(2, 16, 16, 7)
    assert hd.num_launches() > 50
    hd.attach(image)                                    # same buffer again: the plan stays
    assert hd.num_launches() > 50
    raw = torch.empty(n + 256, dtype=torch.uint8)
    image2 = raw[(-raw.data_ptr()) % 256:][:n]
    image2.copy_(image)
    hd.attach(image2)                                   # the buffer moved: plan (absolute addresses) and graph are dropped
    assert hd.num_launches() == 0
    ws = ctypes.c_size_t()
    assert lib.mi355x_sd_unet_bind_workspace(hd.h, image2.data_ptr(), n) != 0 and b"no plan" in lib.mi355x_sd_last_error()
    assert hd.plan(2, 16, 16, 7) > 0 and hd.num_launches() > 50
    buf = torch.empty(n, dtype=torch.uint8)
    assert lib.mi355x_sd_unet_pack_weights(hd.h, buf.data_ptr(), n) != 0 and b"already released" in lib.mi355x_sd_last_error()
    assert lib.mi355x_sd_unet_finalize_weights(hd.h, image2.data_ptr(), n, None) != 0 and b"already released" in lib.mi355x_sd_last_error()
    shp = (ctypes.c_int64 * 4)(*P["conv_in.weight"].shape)
    assert lib.mi355x_sd_unet_load_weight(hd.h, b"conv_in.weight", P["conv_in.weight"].data_ptr(), shp, 4, 0) != 0   # too late
    del ws


def test_config_values_are_validated_at_create():
    """what parse_config cannot build must fail at create with a message -- a zero head count used to be an integer
    division by zero (SIGFPE) while packing, `mid_block_type: null` silently built a mid block."""
    lib = _lib.load()
    for bad, frag in ((dict(TINY, attention_head_dim=0), b"attention_head_dim"), (dict(TINY, attention_head_dim=3), b"attention_head_dim"),
                      (dict(TINY, block_out_channels=(64, 100)), b"block_out_channels"), (dict(TINY, norm_num_groups=0), b"norm_num_groups"),
                      (dict(TINY, mid_block_type=None), b"mid_block_type"), (dict(TINY, upcast_attention=True), b"upcast_attention"),
                      (dict(TINY, data_format="NHWC"), b"data_format"), (dict(TINY, layers_per_block=0), b"layers_per_block"),
                      (dict(TINY, cross_attention_dim=12), b"cross_attention_dim")):
        with pytest.raises(_lib.MI355XError):
            UNetHandle(bad)
        assert frag in lib.mi355x_sd_last_error(), (bad, lib.mi355x_sd_last_error())
    h = ctypes.c_void_p()
    assert lib.mi355x_sd_unet_create(b'{"block_out_channels": [64, "128"], "down_block_types": ["DownBlock2D", "DownBlock2D"], '
                                     b'"up_block_types": ["UpBlock2D", "UpBlock2D"]}', ctypes.byref(h)) != 0


def test_refusals_are_loud():
    lib = _lib.load()
    for bad in (dict(TINY, class_embeddings_concat=True), dict(MINI_XL, class_embed_type="timestep", class_embeddings_concat=True),
                dict(TINY, time_cond_proj_dim=12), dict(TINY, attention_type="gated"),
                dict(TINY, class_embed_type="projection"), dict(TINY, class_embed_type="no-such-type"),
                dict(TINY, down_block_types=("DownBlock2D", "AttnDownBlock2D")), dict(TINY, encoder_hid_dim_type="text_proj", encoder_hid_dim=8),
                dict(TINY, encoder_hid_dim_type="ip_image_proj"), dict(TINY, encoder_hid_dim=8)):
        with pytest.raises(_lib.MI355XError):
            UNetHandle(bad)
    for bad in (b"{not json", b"", b"[]", b'{"block_out_channels": "x"}', b'{"a": {"b": [1,2,',
                b'{"block_out_channels": [32, 64], "layers_per_block": 99999999999999999999}'):
        h = ctypes.c_void_p()      # malformed config text: an error code and a message, never an exception across the C boundary
        assert lib.mi355x_sd_unet_create(bad, ctypes.byref(h)) != 0 and not h.value and lib.mi355x_sd_last_error()
    hd = UNetHandle(TINY)
    # inputs the model does not have cannot be bound; unknown names neither (mi355x_sd_unet_set_input)
    dummy = ctypes.c_void_p(64)
    assert lib.mi355x_sd_unet_set_input(hd.h, b"class_labels", dummy) != 0 and b"no class embedding" in lib.mi355x_sd_last_error()
    assert lib.mi355x_sd_unet_set_input(hd.h, b"timestep_cond", dummy) != 0 and b"time_cond_proj_dim" in lib.mi355x_sd_last_error()
    assert lib.mi355x_sd_unet_set_input(hd.h, b"image_embeds", dummy) != 0 and b"no IP-Adapter" in lib.mi355x_sd_last_error()
    assert lib.mi355x_sd_unet_set_input(hd.h, b"no_such_input", dummy) != 0 and b"unknown input" in lib.mi355x_sd_last_error()
    assert lib.mi355x_sd_unet_set_ip_adapter_scale(hd.h, 0.5) != 0 and b"no IP-Adapter" in lib.mi355x_sd_last_error()
    assert lib.mi355x_sd_unet_set_input(hd.h, b"class_labels", None) == 0
    hc = UNetHandle(CLASS_CONFIGS["class-table"])
    assert lib.mi355x_sd_unet_set_input(hc.h, b"class_labels", dummy) == 0 and lib.mi355x_sd_unet_set_input(hc.h, b"class_labels", None) == 0
    with pytest.raises(_lib.MI355XError, match="never loaded"):
        hd.weight_bytes()
    w = torch.zeros(3, 3)
    shp = (ctypes.c_int64 * 2)(3, 3)
    assert lib.mi355x_sd_unet_load_weight(hd.h, b"conv_in.weight", w.data_ptr(), shp, 2, 0) != 0
    assert b"shape" in lib.mi355x_sd_last_error()
    assert lib.mi355x_sd_unet_load_weight(hd.h, b"no.such.weight", w.data_ptr(), shp, 2, 0) != 0
    n = ctypes.c_size_t()
    assert lib.mi355x_sd_unet_plan(hd.h, 1, 8, 8, 7, ctypes.byref(n)) != 0      # weights not finalized


def test_plan_ex_flags_and_skip_shapes():
    """mi355x_sd_unet_plan_ex on the host (planning touches no device memory): unknown flags are refused, the optional inputs add
    their launches (mask -> bias; one in-place NCHW add per skip tensor + the mid output), the skip shapes are what the Python
    planner hands ControlNet, and a self-attention mask is refused where the reference's shapes cannot take it."""
    from paddlemix_amd.unet import synth_unet_params
    lib = _lib.load()
    hd = UNetHandle(TINY)
    hd.load(synth_unet_params(TINY, seed=1))
    image = hd.pack()
    hd.attach(image)
    n = ctypes.c_size_t()
    assert lib.mi355x_sd_unet_plan_ex(hd.h, 2, 16, 16, 7, 8, ctypes.byref(n)) != 0        # unknown flag bit
    base_ws = hd.plan(2, 16, 16, 7)
    base = hd.num_launches()
    shapes = hd.skip_shapes()
    from tests.test_host_logic import _controlnet_residuals
    down, mid = _controlnet_residuals(TINY, 2, 16, 16)
    assert shapes == [tuple(d.shape[1:]) for d in down] + [tuple(mid.shape[1:])]
    assert hd.plan(2, 16, 16, 7, _lib.UNET_ENC_MASK) > base_ws and hd.num_launches() == base + 1
    assert hd.plan(2, 16, 16, 7, _lib.UNET_CONTROLNET) > base_ws and hd.num_launches() == base + len(shapes)
    assert hd.plan(2, 16, 16, 7, _lib.UNET_ENC_MASK | _lib.UNET_CONTROLNET) > base_ws and hd.num_launches() == base + 1 + len(shapes)
    # TINY attends at two resolutions: a mask over the H*W tokens of the first cannot fit the second (the reference fails on shapes)
    with pytest.raises(_lib.MI355XError, match="key tokens"):
        hd.plan(2, 16, 16, 7, _lib.UNET_SELF_MASK)
    assert hd.num_launches() == 0 and lib.mi355x_sd_unet_num_skips(hd.h) == -1             # a failed plan leaves no plan behind
    assert hd.plan(2, 16, 16, 7) == base_ws and hd.num_launches() == base


def test_ip_adapter_scale_is_a_constant_of_the_plan():
    """mi355x_sd_unet_set_ip_adapter_scale: 0 drops the image-token attention launches (one per cross-attention, like the Python
    planner), a new value discards the plan, a non-finite one is refused"""
    cfg = CLASS_CONFIGS["ip-adapter"]
    P = synth_unet_params(cfg, seed=5)
    lib = _lib.load()
    hd = UNetHandle(cfg)
    hd.load(P)
    image = hd.pack()
    hd.attach(image)
    hd.plan(2, 16, 16, 7)
    n1 = hd.num_launches()
    model = on_emulator(UNet2DConditionModel, cfg, P)
    assert n1 == len(model._get_plan(2, 16, 16, 7).prog)
    assert lib.mi355x_sd_unet_set_ip_adapter_scale(hd.h, 1.0) == 0 and hd.num_launches() == n1      # unchanged value: plan kept
    assert lib.mi355x_sd_unet_set_ip_adapter_scale(hd.h, 0.0) == 0 and hd.num_launches() == 0       # plan discarded
    hd.plan(2, 16, 16, 7)
    model.set_ip_adapter_scale(0.0)
    n0 = hd.num_launches()
    assert n0 == len(model._get_plan(2, 16, 16, 7).prog) and n0 < n1
    assert lib.mi355x_sd_unet_set_ip_adapter_scale(hd.h, float("nan")) != 0 and b"finite" in lib.mi355x_sd_last_error()
