"""Checkpoint / weight formats of ppdiffusers models, for the MI355X model classes.

Mirrors what ``ModelMixin.from_pretrained`` resolves and ``load_state_dict`` reads
(PPD/models/modeling_utils.py:150-214, :661; file names PPD/utils/constants.py:42-56):

  ``diffusion_paddle_model.safetensors`` (+ ``.index.json`` shards)   Paddle layouts, metadata format "pd" / "np"
  ``diffusion_pytorch_model.safetensors`` (+ ``.index.json``)         torch layouts, metadata format "pt" (the default
                                                                      when the metadata is missing)
  ``model_state.pdparams``                                            ``paddle.save`` pickle of numpy arrays
  ``config.json``                                                     constructor kwargs (keys starting with "_" dropped)

torch -> Paddle conversion is the Linear transposition of ``convert_pytorch_state_dict_to_paddle``
(PPD/models/modeling_pytorch_paddle_utils.py:27-63): every 2-D ``.weight`` that belongs to an ``nn.Linear`` is stored
[out, in] by torch and [in, out] by Paddle; the set of Linear weights is taken from the model's own parameter table
(``*_param_shapes``), exactly like the reference takes it from ``named_sublayers``.

The models consume Paddle layouts (the reference's parameter names and shapes); fp8 quantisation of the SD3 block
matrices happens on load through ``weight_dtype="fp8"``.
"""
from __future__ import annotations

import io
import json
import os
import pickle
from typing import Callable, Dict, Mapping, Optional, Tuple

import numpy as np
import torch

CONFIG_NAME = "config.json"
PADDLE_SAFETENSORS_WEIGHTS_NAME = "diffusion_paddle_model.safetensors"
TORCH_SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
PADDLE_WEIGHTS_NAME = "model_state.pdparams"
# transformers-style text encoders (CLIP / T5 sub-folders of a pipeline) name their single file model.safetensors / model_state.pdparams
_CANDIDATES = (PADDLE_SAFETENSORS_WEIGHTS_NAME, TORCH_SAFETENSORS_WEIGHTS_NAME, PADDLE_WEIGHTS_NAME, "model.safetensors")


class Table(tuple):
    """Shape of a 2-D parameter that is NOT an nn.Linear weight (nn.Embedding tables, T5's relative-attention bias): the
    reference transposes only nn.Linear weights between the torch and Paddle layouts
    (convert_pytorch_state_dict_to_paddle, modeling_pytorch_paddle_utils.py:27-47), so these keep their [num, dim] shape in
    both formats. The *_param_shapes tables mark them with this type; it is a plain tuple everywhere else."""

Tensor = torch.Tensor


class _NumpyOnlyUnpickler(pickle.Unpickler):
    """``model_state.pdparams`` is a pickle; only numpy array reconstruction is allowed to run."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
                ("numpy._core.multiarray", "scalar"), ("collections", "OrderedDict"), ("builtins", "dict")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to unpickle {module}.{name} from a .pdparams file")


def load_state_dict(checkpoint_file: str) -> Tuple[Dict[str, Tensor], str]:
    """-> (name -> tensor, data_format in {"pd", "pt", "np"}); modeling_utils.py:150-214."""
    if checkpoint_file.endswith(".safetensors"):
        from safetensors import safe_open
        with safe_open(checkpoint_file, framework="pt") as f:
            meta = f.metadata() or {}
            fmt = meta.get("format", "pt")
            if fmt not in ("pt", "pd", "np"):
                raise OSError(f"The safetensors archive passed at {checkpoint_file} does not contain the valid metadata.")
            return {k: f.get_tensor(k) for k in f.keys()}, fmt
    if checkpoint_file.endswith(".pdparams"):
        with open(checkpoint_file, "rb") as fh:
            raw = _NumpyOnlyUnpickler(io.BytesIO(fh.read())).load()
        out = {}
        for k, v in raw.items():
            if k == "StructuredToParameterName@@":   # paddle.save bookkeeping entry
                continue
            if isinstance(v, np.ndarray):
                out[k] = torch.from_numpy(np.ascontiguousarray(v))
        return out, "pd"
    raise OSError(f"unsupported checkpoint format: {checkpoint_file} (expected .safetensors or .pdparams)")


def resolve_weight_files(model_dir: str):
    """[(file, keys or None)] in load order: a single file, or the shards an ``*.index.json`` weight_map names."""
    for name in _CANDIDATES:
        idx = os.path.join(model_dir, name + ".index.json")
        if os.path.isfile(idx):
            with open(idx) as fh:
                wm = json.load(fh)["weight_map"]
            return [os.path.join(model_dir, f) for f in sorted(set(wm.values()))]
        path = os.path.join(model_dir, name)
        if os.path.isfile(path):
            return [path]
    raise OSError(f"no weights found under {model_dir}: looked for {', '.join(_CANDIDATES)} (+ .index.json)")


def to_paddle_layout(state: Mapping[str, Tensor], shapes: Mapping[str, tuple], data_format: str,
                     optional: Optional[Mapping[str, tuple]] = None) -> Dict[str, Tensor]:
    """Select the model's parameters and bring them to the reference's Paddle layouts (Linear: [in, out]). `optional`: entries
    of the reference's state dict that checkpoints may or may not carry (taken when present, never reported missing)."""
    out: Dict[str, Tensor] = {}
    missing, bad = [], []
    opt = dict(optional or {})
    for name, shape in list(shapes.items()) + list(opt.items()):
        if name not in state:
            if name not in opt:
                missing.append(name)
            continue
        t = state[name]
        if data_format == "pt" and t.dim() == 2 and len(shape) == 2 and not isinstance(shape, Table):   # nn.Linear: torch keeps [out, in]
            t = t.t()
        if tuple(t.shape) != tuple(shape):
            bad.append(f"{name}: checkpoint {tuple(t.shape)} vs model {tuple(shape)}")
            continue
        out[name] = t.to(torch.float32).contiguous()
    if missing:
        raise KeyError(f"checkpoint is missing {len(missing)} parameters, e.g. {missing[:4]}")
    if bad:
        raise ValueError("shape mismatch: " + "; ".join(bad[:4]))
    return out


def from_paddle_layout(params: Mapping[str, Tensor], data_format: str,
                       shapes: Optional[Mapping[str, tuple]] = None) -> Dict[str, Tensor]:
    """Inverse of `to_paddle_layout` for writing a checkpoint in torch layouts (format "pt"). `shapes` (the model's parameter
    table) tells nn.Linear weights from embedding tables; without it every 2-D tensor is taken for a Linear weight, which
    is only right for models without tables (the UNet without class embeddings, the VAE)."""
    if data_format == "pt":
        lin = lambda k, v: v.dim() == 2 and not (shapes is not None and isinstance(shapes.get(k), Table))  # noqa: E731
        return {k: (v.t().contiguous() if lin(k, v) else v.contiguous()) for k, v in params.items()}
    return {k: v.contiguous() for k, v in params.items()}


def save_pretrained(model_dir: str, config: Mapping, params: Mapping[str, Tensor], data_format: str = "pd",
                    shapes: Optional[Mapping[str, tuple]] = None) -> str:
    """Write ``config.json`` + one safetensors file the way ppdiffusers lays a model directory out."""
    from safetensors.torch import save_file
    if data_format not in ("pd", "pt"):
        raise ValueError("data_format must be 'pd' or 'pt'")
    os.makedirs(model_dir, exist_ok=True)
    with open(os.path.join(model_dir, CONFIG_NAME), "w") as fh:
        json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in config.items()}, fh, indent=2)
    name = PADDLE_SAFETENSORS_WEIGHTS_NAME if data_format == "pd" else TORCH_SAFETENSORS_WEIGHTS_NAME
    path = os.path.join(model_dir, name)
    save_file(from_paddle_layout({k: v.detach().cpu() for k, v in params.items()}, data_format, shapes), path,
              metadata={"format": data_format})
    return path


def load_pretrained(model_dir: str, shapes_fn: Callable[[Mapping], Mapping[str, tuple]],
                    subfolder: Optional[str] = None, optional_fn: Optional[Callable[[Mapping], Mapping[str, tuple]]] = None
                    ) -> Tuple[dict, Dict[str, Tensor]]:
    """-> (config dict, parameters in Paddle layouts) for a model directory (optionally ``subfolder`` of a pipeline)."""
    if subfolder:
        model_dir = os.path.join(model_dir, subfolder)
    cfg_path = os.path.join(model_dir, CONFIG_NAME)
    if not os.path.isfile(cfg_path):
        raise OSError(f"{cfg_path} not found")
    with open(cfg_path) as fh:
        config = {k: v for k, v in json.load(fh).items() if not k.startswith("_")}
    state: Dict[str, Tensor] = {}
    fmt = None
    for f in resolve_weight_files(model_dir):
        part, f_fmt = load_state_dict(f)
        if fmt is not None and f_fmt != fmt:
            raise OSError("shards of one checkpoint disagree on the data format")
        fmt = f_fmt
        state.update(part)
    return config, to_paddle_layout(state, shapes_fn(config), "pd" if fmt == "np" else fmt, optional_fn(config) if optional_fn else None)


def fuse_lora(params: Mapping[str, Tensor], lora: Mapping[str, Tensor], lora_scale: float = 1.0,
              network_alphas: Optional[Mapping[str, float]] = None, prefix: str = "unet.", safe_fusing: bool = False
              ) -> Dict[str, Tensor]:
    """The LoRA branch of LoRACompatibleLinear / LoRACompatibleConv (PPD/models/lora.py:364-377, 453-459), taken the way the
    reference takes it for inference: merged into the weights (``_fuse_lora``, lora.py:312-344 conv, :404-425 linear) --
    ``W + lora_scale * (down @ up) * network_alpha / rank`` -- so the kernels never see a second GEMM.

    ``params``: the model's state dict in Paddle layouts (Linear ``[in, out]``, conv OIHW). ``lora``: LoRA tensors named
    ``[prefix]<layer>.lora.down.weight`` / ``.lora.up.weight`` in Paddle layouts (Linear down ``[in, rank]``, up ``[rank, out]``;
    conv down ``[rank, in, kh, kw]``, up ``[out, rank, 1, 1]``), optional ``<layer>.alpha`` scalars (kohya ``network_alpha``) or a
    ``network_alphas`` mapping. Returns a new dict; raises on LoRA entries that match no layer."""
    out = dict(params)
    layers = {}
    for k in lora:
        name = k[len(prefix):] if prefix and k.startswith(prefix) else k
        for tail in (".lora.down.weight", ".lora.up.weight", ".alpha"):
            if name.endswith(tail):
                layers.setdefault(name[: -len(tail)], {})[tail] = lora[k]
                break
        else:
            raise KeyError(f"{k}: not a LoRA tensor name (<layer>.lora.down.weight / .lora.up.weight / .alpha)")
    for layer, t in layers.items():
        wkey = layer + ".weight"
        if wkey not in params:
            raise KeyError(f"LoRA layer {layer!r} has no counterpart {wkey!r} in the model")
        if ".lora.down.weight" not in t or ".lora.up.weight" not in t:
            raise KeyError(f"LoRA layer {layer!r} needs both .lora.down.weight and .lora.up.weight")
        w = params[wkey].to(torch.float32)
        down, up = t[".lora.down.weight"].to(torch.float32).to(w.device), t[".lora.up.weight"].to(torch.float32).to(w.device)
        alpha = t.get(".alpha")
        if alpha is None and network_alphas is not None:
            alpha = network_alphas.get(layer, network_alphas.get(prefix + layer if prefix else layer))
        if w.dim() == 2:     # Linear [in, out]: down [in, rank], up [rank, out]
            rank = down.shape[1]
            if alpha is not None:
                up = up * (float(alpha) / rank)
            delta = down @ up
        else:                # conv OIHW: up [O, rank, 1, 1] x down [rank, I, kh, kw]
            rank = down.shape[0]
            if alpha is not None:
                up = up * (float(alpha) / rank)
            delta = (up.flatten(1) @ down.flatten(1)).reshape(w.shape)
        if tuple(delta.shape) != tuple(w.shape):
            raise ValueError(f"{layer}: LoRA delta of shape {tuple(delta.shape)} does not fit weight {tuple(w.shape)}")
        fused = w + lora_scale * delta
        if safe_fusing and torch.isnan(fused).any():
            raise ValueError(f"This LoRA weight seems to be broken. Encountered NaN values when trying to fuse LoRA weights "
                             f"for {layer}. LoRA weights will not be fused.")
        out[wkey] = fused.to(params[wkey].dtype)
    return out


class PretrainedMixin:
    """``Model.from_pretrained(dir, subfolder=..., **ctor_kwargs)`` for the MI355X model classes; the class names its
    parameter table in ``_param_shapes``."""
    _param_shapes: Callable[[Mapping], Mapping[str, tuple]] = None
    _optional_param_shapes: Optional[Callable[[Mapping], Mapping[str, tuple]]] = None   # taken from the checkpoint when present

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, **kwargs):
        if not os.path.isdir(pretrained_model_name_or_path):
            raise OSError(f"{pretrained_model_name_or_path} is not a local directory (there is no hub access here)")
        config, params = load_pretrained(pretrained_model_name_or_path, cls._param_shapes, subfolder, cls._optional_param_shapes)
        return cls(config, params, **kwargs)
