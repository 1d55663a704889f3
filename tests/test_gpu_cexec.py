"""-m gpu: seam B1 of the C ABI (mi355x_sd_unet_*, csrc/unet_exec.hip) on the device.
(1) the handle-based model equals the Python-planned model bit for bit (same launches, same packing);
(2) a plain-C client (tests/c/unet_exec_test.c: gcc, hipMalloc, no torch, no Python) produces the same bits as the Python path
    on the same LCG-generated weights and inputs."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from tests.configs import MINI_XL, TINY
from tests.test_gpu_unet import _cuda, _inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,cfg,B,H,W,L", [("tiny", TINY, 2, 16, 16, 7), ("mini-xl", MINI_XL, 2, 32, 32, 77)])
@pytest.mark.parametrize("rd", [None, "fp32"])
def test_handle_model_equals_python_planned_model(name, cfg, B, H, W, L, rd):
    from paddlemix_amd.cexec import CUNet2DConditionModel
    from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params
    P = synth_unet_params(cfg, seed=1234)
    sample, enc, added = _inputs(cfg, B, H, W, L)
    want = UNet2DConditionModel(cfg, P, residual_dtype=rd, use_graph=False)(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    for graph in (False, True):
        m = CUNet2DConditionModel(cfg, P, use_graph=graph, residual_dtype=rd)
        got = m(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
        again = m(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
        assert torch.equal(got, want) and torch.equal(again, want), (name, rd, graph, (got - want).abs().max())
    # scale_model_input folded into conv_in through the optional device scalar
    scaled = m(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added), in_scale=0.5).sample
    ref = UNet2DConditionModel(cfg, P, residual_dtype=rd)
    plan = ref._get_plan(B, H, W, L)
    with torch.cuda.stream(ref._stream):
        ref.stage_inputs(plan, _cuda(sample), 501, _cuda(enc), _cuda(added), in_scale=0.5)
        want_s = ref.run(plan).clone()
    torch.cuda.synchronize()
    assert torch.equal(scaled, want_s)
    if cfg.get("addition_embed_type") == "text_time":
        from paddlemix_amd import _lib
        # (the wrapper checks its inputs before the C call and raises what the reference raises, unet_2d_condition.py:993-1001; the
        # C entry point itself answers MI355X_SD_ERR_INVALID with the same text: tests/c/unet_exec_test.c)
        with pytest.raises((ValueError, _lib.MI355XError), match="text_time"):
            m(_cuda(sample), 501, _cuda(enc))


def _lcg_uniform(n, seed):
    """the generator of tests/c/unet_exec_test.c, vectorised: s_i = a^i s_0 + c (1 + a + ... + a^(i-1)) mod 2^64"""
    a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        A = np.cumprod(np.full(n, a, dtype=np.uint64))                    # a^1 .. a^n
        G = np.concatenate([np.ones(1, np.uint64), A[:-1]]).cumsum(dtype=np.uint64)   # 1 + a + ... + a^(i-1)
        s = A * np.uint64(seed) + c * G
    return ((s >> np.uint64(40)).astype(np.float64) / 8388608.0 - 1.0).astype(np.float32)


def _fnv(name):
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) % (1 << 64)
    return h


@pytest.mark.parametrize("name,cfg,B,H,W,L,rd", [("tiny", TINY, 2, 16, 16, 7, 0), ("mini-xl", MINI_XL, 1, 16, 16, 77, 0),
                                                 ("mini-xl-f32resid", MINI_XL, 1, 16, 16, 77, 1)])
def test_plain_c_client_matches_python_path(tmp_path, name, cfg, B, H, W, L, rd):
    from paddlemix_amd.unet import UNet2DConditionModel, unet_param_shapes
    exe = str(tmp_path / "unet_exec_test")
    cc = ["gcc", "-std=c11", "-O2", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
          os.path.join(ROOT, "tests", "c", "unet_exec_test.c"), "-L" + os.path.join(ROOT, "paddlemix_amd"), "-lmi355x_sd",
          "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cc, check=True)
    cj = tmp_path / "config.json"
    cj.write_text(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}))
    out = tmp_path / "out.bin"
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "paddlemix_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(cj), str(B), str(H), str(W), str(L), str(out), str(rd)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    print(r.stdout.strip())
    got = torch.from_numpy(np.fromfile(out, dtype=np.float32).reshape(B, 4, H, W))
    # the same weights / inputs on the Python side
    P = {}
    for pname, shape in unet_param_shapes(cfg).items():
        n = int(np.prod(shape))
        u = _lcg_uniform(n, _fnv(pname))
        if pname.endswith(".bias"):
            v = np.float32(0.0) + np.float32(0.03) * u
        elif len(shape) == 1:
            v = np.float32(1.0) + np.float32(0.03) * u
        elif len(shape) == 2:
            v = np.float32(0.0) + np.float32(np.float32(1.7) / np.sqrt(np.float32(shape[0]))) * u
        else:
            v = np.float32(0.0) + np.float32(np.float32(1.7) / np.sqrt(np.float32(shape[1] * shape[2] * shape[3]))) * u
        P[pname] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    cross = cfg["cross_attention_dim"]
    sample = torch.from_numpy((np.float32(1.7) * _lcg_uniform(B * 4 * H * W, 11)).reshape(B, 4, H, W))
    enc = torch.from_numpy((np.float32(1.7) * _lcg_uniform(B * L * cross, 12)).reshape(B, L, cross))
    added = None
    if cfg.get("addition_embed_type") == "text_time":
        td = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
        added = dict(text_embeds=torch.from_numpy((np.float32(1.7) * _lcg_uniform(B * td, 13)).reshape(B, td)),
                     time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(B, 1))
    model = UNet2DConditionModel(cfg, P, residual_dtype="fp32" if rd else None, use_graph=False)
    want = model(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample.cpu()
    assert torch.isfinite(got).all() and got.abs().max() > 0
    assert torch.equal(got, want), (got - want).abs().max()
