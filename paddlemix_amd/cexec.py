"""Thin Python wrapper over seam B1 of the C ABI (``mi355x_sd_unet_*``, csrc/unet_exec.hip): the whole UNet behind one handle.

The program builder, the weight packing and the execution live in the C library; this class only marshals: config -> JSON text,
parameters -> host pointers, two caller-owned device buffers (weights, workspace) as torch tensors, and device pointers per
call. It presents the reference's call contract (``unet(sample, t, encoder_hidden_states, added_cond_kwargs=...,
return_dict=False) -> (noise_pred,)``, unet_2d_condition.py:809-1207) like ``paddlemix_amd.unet.UNet2DConditionModel`` does --
the latter emits the same launches from Python (since round 6 the C++ planner builds every config field it builds: masks,
ControlNet residuals, class labels, timestep_cond, the IP-Adapter image embeddings, odd latent sizes); the handle form is what a
compiled host binds.
"""
from __future__ import annotations

import ctypes
import json
from types import SimpleNamespace
from typing import Dict, Mapping, Optional

import torch

from . import _lib
from .program import WORKSPACE_BYTES

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class UNetHandle:
    """owns the C handle; usable without a GPU up to `pack()` (host logic), which the CPU tests exercise"""

    def __init__(self, config: Mapping, residual_dtype: Optional[str] = None, fold_softmax_scale: bool = True):
        self.lib = _lib.load()
        self.config_dict = {k: (list(v) if isinstance(v, tuple) else v) for k, v in config.items() if not k.startswith("_")}
        self.h = ctypes.c_void_p()
        _lib.check(self.lib.mi355x_sd_unet_create(json.dumps(self.config_dict).encode(), ctypes.byref(self.h)))
        if residual_dtype == "fp32":
            _lib.check(self.lib.mi355x_sd_unet_set_option(self.h, b"residual_f32", 1))
        _lib.check(self.lib.mi355x_sd_unet_set_option(self.h, b"fold_softmax_scale", 1 if fold_softmax_scale else 0))

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.h.value:
                self.lib.mi355x_sd_unet_destroy(self.h)
                self.h = ctypes.c_void_p()
        except Exception:   # interpreter shutdown
            pass

    def param_shapes(self) -> Dict[str, tuple]:
        out = {}
        name, shape, nd = ctypes.c_char_p(), (ctypes.c_int64 * 4)(), ctypes.c_int()
        for i in range(self.lib.mi355x_sd_unet_num_params(self.h)):
            _lib.check(self.lib.mi355x_sd_unet_param_info(self.h, i, ctypes.byref(name), shape, ctypes.byref(nd)))
            out[name.value.decode()] = tuple(int(shape[j]) for j in range(nd.value))
        return out

    def load(self, params: Mapping[str, torch.Tensor]) -> None:
        for name in self.param_shapes():
            if name not in params:
                raise KeyError(f"missing parameter {name}")
            t = params[name].detach().cpu().contiguous()
            if t.dtype not in _DT:
                t = t.float()
            shp = (ctypes.c_int64 * t.dim())(*t.shape)
            _lib.check(self.lib.mi355x_sd_unet_load_weight(self.h, name.encode(), t.data_ptr(), shp, t.dim(), _DT[t.dtype]))

    def weight_bytes(self) -> int:
        n = ctypes.c_size_t()
        _lib.check(self.lib.mi355x_sd_unet_weight_bytes(self.h, ctypes.byref(n)))
        return n.value

    def pack(self) -> torch.Tensor:
        """the packed weight image in host memory (uint8)"""
        n = self.weight_bytes()
        raw = torch.empty(n + 256, dtype=torch.uint8)
        skip = (-raw.data_ptr()) % 256          # the image is addressed in 256-byte units
        buf = raw[skip:skip + n]
        _lib.check(self.lib.mi355x_sd_unet_pack_weights(self.h, buf.data_ptr(), buf.numel()))
        return buf

    def packed_tensor(self, image: torch.Tensor, key: str) -> torch.Tensor:
        off, nb, rows, cols = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.mi355x_sd_unet_packed_tensor(self.h, key.encode(), ctypes.byref(off), ctypes.byref(nb), ctypes.byref(rows),
                                                          ctypes.byref(cols)))
        raw = image[off.value:off.value + nb.value]
        if nb.value == rows.value * cols.value * 2:
            return raw.view(_lib.elem_dtype()).reshape(rows.value, cols.value)
        return raw.view(torch.float32).reshape(-1)

    def attach(self, device_or_host_buffer: torch.Tensor) -> None:
        _lib.check(self.lib.mi355x_sd_unet_attach_weights(self.h, device_or_host_buffer.data_ptr(), device_or_host_buffer.numel()))

    def plan(self, B: int, H: int, W: int, L: int, flags: int = 0) -> int:
        n = ctypes.c_size_t()
        _lib.check(self.lib.mi355x_sd_unet_plan_ex(self.h, B, H, W, L, flags, ctypes.byref(n)))
        return n.value

    def skip_shapes(self):
        """[(C, H, W)] of every skip tensor in production order, then of the mid block's output (ControlNet residual shapes)"""
        out = []
        for i in range(self.lib.mi355x_sd_unet_num_skips(self.h) + 1):
            c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            _lib.check(self.lib.mi355x_sd_unet_skip_shape(self.h, i, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)))
            out.append((c.value, h.value, w.value))
        return out

    def num_launches(self) -> int:
        return self.lib.mi355x_sd_unet_num_launches(self.h)


class CUNet2DConditionModel:
    """``UNet2DConditionModel`` on the handle API. Buffers (weights, workspace -- which contains the handle's split-K scratch since ABI 12) are torch
    tensors owned here -- i.e. by the caller of the C ABI."""

    def __init__(self, config: Mapping, params: Mapping[str, torch.Tensor], device="cuda", use_graph: bool = True,
                 residual_dtype: Optional[str] = None, fold_softmax_scale: bool = True):
        if not torch.cuda.is_available():
            raise _lib.MI355XError("CUNet2DConditionModel(mi355x) needs a GPU; there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.hd = UNetHandle(config, residual_dtype, fold_softmax_scale)
        _lib.check(self.hd.lib.mi355x_sd_init(self.device.index))
        self.config = SimpleNamespace(**self.hd.config_dict)
        if self.hd.config_dict.get("addition_embed_type") == "text_time":   # read by StableDiffusionXLPipeline._get_add_time_ids (:611)
            self.add_embedding = SimpleNamespace(linear_1=SimpleNamespace(in_features=self.hd.config_dict["projection_class_embeddings_input_dim"]))
        self.dtype = _lib.elem_dtype()
        self.use_graph = use_graph
        self._stream = torch.cuda.Stream(device=self.device)
        self.hd.load(params)
        self._weights = torch.empty(self.hd.weight_bytes(), device=self.device, dtype=torch.uint8)
        _lib.check(self.hd.lib.mi355x_sd_unet_finalize_weights(self.hd.h, self._weights.data_ptr(), self._weights.numel(),
                                                                self._stream.cuda_stream))
        self._geom = None
        self._workspace = None

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, return_dict: bool = True, in_scale=None,
                attention_mask=None, encoder_attention_mask=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, class_labels=None, timestep_cond=None):
        lib, h = self.hd.lib, self.hd.h
        cfgd = self.hd.config_dict
        if sample.dim() != 4 or sample.shape[1] != cfgd.get("in_channels", 4):
            raise ValueError(f"sample: expected [B, {cfgd.get('in_channels', 4)}, H, W], got {tuple(sample.shape)}")
        B, _, H, W = sample.shape
        cross = cfgd.get("cross_attention_dim", 1280)
        cross = cross[0] if isinstance(cross, (list, tuple)) else cross
        if encoder_hidden_states.dim() != 3 or encoder_hidden_states.shape[0] != B or encoder_hidden_states.shape[2] != cross:
            # the C side reads B * L * cross_attention_dim floats whatever it is handed
            raise ValueError(f"encoder_hidden_states: expected [{B}, L, {cross}], got {tuple(encoder_hidden_states.shape)}")
        L = encoder_hidden_states.shape[1]
        if torch.is_tensor(timestep) and timestep.numel() > 1:
            tv = timestep.reshape(-1)
            if tv.numel() != B or not bool((tv == tv[0]).all()):
                # the program reads ONE timestep (the pipelines' call form, pipeline_stable_diffusion.py:866-879); the reference
                # would broadcast a [B] tensor per sample (unet_2d_condition.py:946)
                raise ValueError("timestep: per-sample timesteps are not implemented (pass one value, or B equal values)")
        if cfgd.get("addition_embed_type") == "text_time":
            tids = None if added_cond_kwargs is None else added_cond_kwargs.get("time_ids")
            if tids is not None and tids.dim() == 2 and tids.shape[1] != 6:
                # (the Python planner takes any count -- the refiner's 5 micro-conditioning ids; the C++ planner lays out six)
                raise NotImplementedError(f"the C executor (mi355x_sd_unet_*) supports 6 time ids per sample, got {tids.shape[1]}; "
                                          "use paddlemix_amd.unet.UNet2DConditionModel")
            for key, width in (("text_embeds", cfgd["projection_class_embeddings_input_dim"] - 6 * cfgd["addition_time_embed_dim"]),
                               ("time_ids", 6)):
                if added_cond_kwargs is None or key not in added_cond_kwargs:
                    raise ValueError(f"{self.__class__} has the config param `addition_embed_type` set to 'text_time' which requires "
                                     f"the keyword argument `{key}` to be passed in `added_cond_kwargs`")
                if tuple(added_cond_kwargs[key].shape) != (B, width):
                    raise ValueError(f"{key}: expected [{B}, {width}], got {tuple(added_cond_kwargs[key].shape)}")
        # class_labels / timestep_cond: inputs that belong to the config (mi355x_sd_unet_set_input), same checks as the planned model
        ct, nce, ted = cfgd.get("class_embed_type"), cfgd.get("num_class_embeds"), 4 * cfgd.get("block_out_channels", (320,))[0]
        cl = None
        if ct is not None or nce is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            cl = class_labels if torch.is_tensor(class_labels) else torch.as_tensor(class_labels)
            if ct in (None, "timestep"):
                cl = cl.reshape(-1)
                cl = cl.expand(B) if cl.numel() == 1 else cl
                want = (B,)
            else:
                want = (B, ted if ct == "identity" else cfgd["projection_class_embeddings_input_dim"])
            if tuple(cl.shape) != want:
                raise ValueError(f"class_labels of shape {tuple(class_labels.shape)}, expected {want}")
            if ct is None and not cl.is_cuda and cl.numel() and not (0 <= int(cl.min()) and int(cl.max()) < nce):
                raise IndexError(f"class_labels must lie in [0, {nce}), got {cl.tolist()}")   # (host labels only: see unet.py)
        tcp = cfgd.get("time_cond_proj_dim")
        if timestep_cond is not None:
            if tcp is None:
                raise ValueError("timestep_cond was passed but the model has no `time_cond_proj_dim`")
            if tuple(timestep_cond.shape) != (B, tcp):
                raise ValueError(f"timestep_cond of shape {tuple(timestep_cond.shape)}, expected {(B, tcp)}")
        ie = None
        if cfgd.get("encoder_hid_dim_type") == "ip_image_proj":
            if added_cond_kwargs is None or "image_embeds" not in added_cond_kwargs:
                raise ValueError(f"{self.__class__} has the config param `encoder_hid_dim_type` set to 'ip_image_proj' which "
                                 "requires the keyword argument `image_embeds` to be passed in  `added_conditions`")
            ie = added_cond_kwargs["image_embeds"]
            if tuple(ie.shape) != (B, cfgd["encoder_hid_dim"]):
                raise ValueError(f"image_embeds of shape {tuple(ie.shape)}, expected {(B, cfgd['encoder_hid_dim'])}")
        controlnet = down_block_additional_residuals is not None
        if controlnet != (mid_block_additional_residual is not None):
            raise NotImplementedError("ControlNet residuals need both `down_block_additional_residuals` and "
                                      "`mid_block_additional_residual` (the T2I-adapter form is not implemented)")
        flags = ((_lib.UNET_ENC_MASK if encoder_attention_mask is not None else 0) | (_lib.UNET_SELF_MASK if attention_mask is not None else 0) |
                 (_lib.UNET_CONTROLNET if controlnet else 0))
        if encoder_attention_mask is not None and tuple(encoder_attention_mask.shape) != (B, L):
            raise ValueError(f"encoder_attention_mask: expected [{B}, {L}], got {tuple(encoder_attention_mask.shape)}")
        if attention_mask is not None and tuple(attention_mask.shape) != (B, H * W):
            # (a mask over another number of key tokens cannot match the self-attention of the first level either)
            raise ValueError(f"attention_mask: expected [{B}, {H * W}] (one entry per latent token), got {tuple(attention_mask.shape)}")
        if self._geom != (B, H, W, L, flags):
            nbytes = self.hd.plan(B, H, W, L, flags)
            self._workspace = torch.empty(nbytes, device=self.device, dtype=torch.uint8)
            _lib.check(lib.mi355x_sd_unet_bind_workspace(h, self._workspace.data_ptr(), nbytes))
            self._geom = (B, H, W, L, flags)
        if controlnet:
            shapes = self.hd.skip_shapes()
            res = list(down_block_additional_residuals) + [mid_block_additional_residual]
            if len(res) != len(shapes):
                raise ValueError(f"expected {len(shapes) - 1} down_block_additional_residuals, got {len(res) - 1}")
            for r, (c_, h_, w_) in zip(res, shapes):
                if tuple(r.shape) != (B, c_, h_, w_):
                    raise ValueError(f"ControlNet residual of shape {tuple(r.shape)}, expected {(B, c_, h_, w_)}")
        f32 = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()  # noqa: E731
        t = timestep if torch.is_tensor(timestep) else torch.tensor([float(timestep)])
        te = ti = None
        if added_cond_kwargs is not None:
            te = f32(added_cond_kwargs["text_embeds"]) if "text_embeds" in added_cond_kwargs else None
            ti = f32(added_cond_kwargs["time_ids"]) if "time_ids" in added_cond_kwargs else None
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            s, tt, e = f32(sample), f32(t.reshape(-1)[:1]), f32(encoder_hidden_states)
            sc = None if in_scale is None else f32(torch.tensor([float(in_scale)]))
            out = torch.empty((B, self.config.__dict__.get("out_channels", 4), H, W), device=self.device, dtype=torch.float32)
            p = lambda x: None if x is None else x.data_ptr()  # noqa: E731
            em = None if encoder_attention_mask is None else f32(encoder_attention_mask)
            sm = None if attention_mask is None else f32(attention_mask)
            rs = [f32(r) for r in down_block_additional_residuals] if controlnet else []
            rm = f32(mid_block_additional_residual) if controlnet else None
            arr = (ctypes.c_void_p * max(1, len(rs)))(*[r.data_ptr() for r in rs]) if controlnet else None
            clt = tc = None
            if cl is not None:   # table rows travel as int32, everything else as fp32 (include/mi355x_sd.h)
                clt = cl.to(device=self.device, dtype=torch.int32 if ct is None else torch.float32).contiguous()
                _lib.check(lib.mi355x_sd_unet_set_input(h, b"class_labels", clt.data_ptr()))
            if tcp is not None:
                tc = None if timestep_cond is None else f32(timestep_cond)
                _lib.check(lib.mi355x_sd_unet_set_input(h, b"timestep_cond", p(tc)))
            if ie is not None:
                ie = f32(ie)
                _lib.check(lib.mi355x_sd_unet_set_input(h, b"image_embeds", ie.data_ptr()))
            _lib.check(lib.mi355x_sd_unet_forward_ex(h, self._stream.cuda_stream, p(s), p(tt), p(e), p(te), p(ti), p(sc), p(em), p(sm), arr,
                                                     len(rs), p(rm), p(out), 1 if self.use_graph else 0))
            # the bindings pointed at this call's staging tensors: the handle must not keep them past the call
            for nm, bound in ((b"class_labels", clt), (b"timestep_cond", tc), (b"image_embeds", ie)):
                if bound is not None:
                    _lib.check(lib.mi355x_sd_unet_set_input(h, nm, None))
        cur.wait_stream(self._stream)
        for x in [s, tt, e, te, ti, sc, em, sm, rm, clt, tc, ie] + rs:     # keep the staging tensors alive until the stream has consumed them
            if x is not None:
                x.record_stream(self._stream)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)

    __call__ = forward

    def set_ip_adapter_scale(self, scale: float) -> None:
        """IPAdapterAttnProcessor.scale (the reference's ``set_ip_adapter_scale``); a constant of the plan: the next call plans again"""
        if self.hd.config_dict.get("encoder_hid_dim_type") != "ip_image_proj":
            raise ValueError("this UNet has no IP-Adapter (config encoder_hid_dim_type != 'ip_image_proj')")
        _lib.check(self.hd.lib.mi355x_sd_unet_set_ip_adapter_scale(self.hd.h, float(scale)))
        self._geom = None
