"""TEST INFRASTRUCTURE (oracle/): load the REFERENCE's model files, unmodified, from /root/reference and run them on CPU over
oracle/paddle_shim.py (torch fp32 standing in for Paddle's array library).

Only usable in the build container (where /root/reference exists); the GPU box never sees it -- scripts/make_reference_golden.py
uses it to write tests/golden/reference_modules/*.npz, and tests/test_reference_modules.py compares the oracle against the
live reference when it is present and against those committed vectors always.

What is real and what is a stand-in:
  real   ppdiffusers/ppdiffusers/models/*.py for the hot path (the files `MODEL_FILES` names): constructors, forward methods,
         attention processors, parameter names and shapes -- executed as they lie under /root/reference, never copied;
  shim   `paddle` (oracle/paddle_shim.py); and the reference's own plumbing that has nothing to do with the arithmetic:
         ppdiffusers.utils (logging / deprecate / flags), configuration_utils (register_to_config: keeps the constructor
         arguments in `self.config`), modeling_utils.ModelMixin (checkpoint I/O: dropped), loaders (LoRA / IP-Adapter file
         loading: dropped), ppdiffusers.transformers (CLIP classes used for isinstance checks in lora.py).
  `is_ppxformers_available()` answers False, so the reference takes its plain-math attention processors (AttnProcessor:
  baddbmm-softmax-bmm, attention_processor.py) rather than the fused flash-attention custom op.
"""
from __future__ import annotations

import functools
import importlib.util
import inspect
import logging as _pylogging
import os
import sys
import types

REF_ROOT = "/root/reference/ppdiffusers"
PKG = "ppdiffusers"


def available() -> bool:
    """False on machines without the reference checkout (ORACLE_NO_REFERENCE=1 simulates one: the committed vectors must suffice)"""
    return not os.environ.get("ORACLE_NO_REFERENCE") and os.path.isdir(os.path.join(REF_ROOT, PKG, "models"))


class FrozenConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def _register_to_config(init):
    """configuration_utils.py register_to_config: every constructor argument (defaults filled in) lands in self.config"""
    sig = inspect.signature(init)

    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        init_kwargs = {k: v for k, v in kwargs.items() if not k.startswith("_")}
        ba = sig.bind(self, *args, **init_kwargs)
        ba.apply_defaults()
        cfg = {k: v for k, v in ba.arguments.items() if k != "self" and sig.parameters[k].kind != inspect.Parameter.VAR_KEYWORD}
        object.__setattr__(self, "_internal_dict", FrozenConfig(cfg))   # before the body: constructors read self.config
        init(self, *args, **init_kwargs)
    return inner


def _stub_modules(shim):
    P = shim["paddle"]
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        mods[name] = m
        return m

    class _Logging:
        @staticmethod
        def get_logger(name=None):
            return _pylogging.getLogger(name or "ppdiffusers")

    class BaseOutput(dict):
        """utils/outputs.py BaseOutput: a dataclass that also indexes like a tuple / dict"""
        def __post_init__(self):
            for f in self.__dataclass_fields__:
                v = getattr(self, f)
                if v is not None:
                    dict.__setitem__(self, f, v)

        def __getitem__(self, k):
            if isinstance(k, str):
                return dict.__getitem__(self, k)
            return self.to_tuple()[k]

        def to_tuple(self):
            return tuple(dict.values(self))

    def deprecate(*a, **k):
        return None

    class _NoLoaders:
        def maybe_convert_prompt(self, prompt, tokenizer):   # TextualInversionLoaderMixin: no learned tokens loaded
            return prompt

    flags = dict(USE_PEFT_BACKEND=False, BaseOutput=BaseOutput, deprecate=deprecate, logging=_Logging,
                 scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None,
                 is_ppxformers_available=lambda: False, recompute_use_reentrant=lambda: False, use_old_recompute=lambda: False,
                 is_paddle_available=lambda: True, is_torch_available=lambda: False, NEG_INF=-1e4,
                 apply_forward_hook=lambda fn: fn, replace_example_docstring=lambda doc: (lambda fn: fn),
                 is_pp_invisible_watermark_available=lambda: False, PIL_INTERPOLATION={}, CONFIG_NAME="config.json")
    utils = mod(f"{PKG}.utils", **flags)
    utils.__path__ = []
    mod(f"{PKG}.utils.import_utils", is_ppxformers_available=lambda: False, is_torch_available=lambda: False)
    def randn_tensor(shape, generator=None, dtype=None, **_):
        """utils/paddle_utils.py randn_tensor: here the caller owns the draws -- `generator` is a callable shape -> torch tensor
        (the schedulers' deterministic paths draw and then multiply by zero: without one they get zeros)"""
        import torch
        t = generator(list(shape)) if callable(generator) else torch.zeros(list(shape))
        return P.Tensor(t)

    mod(f"{PKG}.utils.paddle_utils", maybe_allow_in_graph=lambda cls: cls, apply_freeu=None, randn_tensor=randn_tensor)

    class ConfigMixin:
        config_name = "config.json"

        @property
        def config(self):
            return self._internal_dict

        def register_to_config(self, **kw):
            d = FrozenConfig(getattr(self, "_internal_dict", {}))
            d.update(kw)
            object.__setattr__(self, "_internal_dict", d)

    mod(f"{PKG}.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=_register_to_config, FrozenDict=FrozenConfig)
    mod(f"{PKG}.loaders", **{n: type(n, (_NoLoaders,), {}) for n in (
        "UNet2DConditionLoadersMixin", "FromOriginalVAEMixin", "FromOriginalControlnetMixin", "PeftAdapterMixin", "FromOriginalModelMixin",
        "FromSingleFileMixin", "IPAdapterMixin", "LoraLoaderMixin", "TextualInversionLoaderMixin", "StableDiffusionXLLoraLoaderMixin",
        "SD3LoraLoaderMixin")})
    mod(f"{PKG}.loaders.single_file_model", FromOriginalModelMixin=_NoLoaders)
    # ppdiffusers.transformers: a bare package over the reference's directory (clip/modeling.py and t5/modeling.py load for real,
    # see _stub_text_encoders); lora.py only wants two class objects for isinstance checks
    tr = mod(f"{PKG}.transformers", **{n: type(n, (), {}) for n in (
        "CLIPTextModel", "CLIPTextModelWithProjection", "CLIPImageProcessor", "CLIPTokenizer", "CLIPVisionModelWithProjection",
        "T5EncoderModel", "T5Tokenizer")})
    tr.__path__ = [os.path.join(REF_ROOT, PKG, "transformers")]
    for sub in ("clip", "t5"):
        m = mod(f"{PKG}.transformers.{sub}")
        m.__path__ = [os.path.join(REF_ROOT, PKG, "transformers", sub)]
    _stub_text_encoders(P, mod)
    _stub_pipelines(P, mod, ConfigMixin)

    class ModelMixin(P.nn.Layer):
        _supports_gradient_checkpointing = False
        gradient_checkpointing = False

        @property
        def dtype(self):
            return P.float32

    mod(f"{PKG}.models.modeling_utils", ModelMixin=ModelMixin)
    mod(f"{PKG}.models.simplified_facebook_dit", SimplifiedFacebookDIT=type("SimplifiedFacebookDIT", (), {}))
    mod(f"{PKG}.models.simplified_sd3", SimplifiedSD3=type("SimplifiedSD3", (), {}))
    return mods


def _stub_pipelines(P, mod, ConfigMixin):
    """what pipelines/stable_diffusion*/pipeline_*.py import besides models and schedulers: the DiffusionPipeline base class (hub /
    device placement / progress bar: dropped to attribute registration), the safety checker (absent). image_processor.py
    (VaeImageProcessor: tensor pre-processing, mask binarisation, resize) is the reference's real file."""
    for sub in ("pipelines", "pipelines.stable_diffusion", "pipelines.stable_diffusion_xl", "pipelines.stable_diffusion_3", "pipelines.controlnet",
                "pipelines.dit", "pipelines.latent_consistency_models"):
        m = mod(f"{PKG}.{sub}")
        m.__path__ = [os.path.join(REF_ROOT, PKG, *sub.split("."))]

    class _Bar:
        def update(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class DiffusionPipeline(ConfigMixin):
        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        def progress_bar(self, iterable=None, total=None):
            return iterable if iterable is not None else _Bar()     # `for t in self.progress_bar(ts)` | `with self.progress_bar(total=n)`

        def maybe_free_model_hooks(self):
            pass

        def set_progress_bar_config(self, **kw):
            pass

        @property
        def _execution_device(self):
            return None

    class ImagePipelineOutput:
        def __init__(self, images):
            self.images = images

    mod(f"{PKG}.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline, ImagePipelineOutput=ImagePipelineOutput)

    mod(f"{PKG}.pipelines.stable_diffusion.safety_checker", StableDiffusionSafetyChecker=type("StableDiffusionSafetyChecker", (), {}))


def ref_pipeline(module: str, package: str = "pipelines.stable_diffusion"):
    """import ppdiffusers.pipelines.<...>.<module> from the reference tree; `from ...models import UNet2DConditionModel` and
    `from ...schedulers import KarrasDiffusionSchedulers` inside it resolve to the real classes"""
    install()
    models, scheds = sys.modules[f"{PKG}.models"], sys.modules[f"{PKG}.schedulers"]
    models.UNet2DConditionModel = ref_module("unet_2d_condition").UNet2DConditionModel
    models.AutoencoderKL = ref_module("autoencoder_kl").AutoencoderKL
    scheds.KarrasDiffusionSchedulers = ref_module("scheduling_utils", "schedulers").KarrasDiffusionSchedulers
    scheds.FlowMatchEulerDiscreteScheduler = ref_module("scheduling_flow_match_euler_discrete", "schedulers").FlowMatchEulerDiscreteScheduler
    models.ControlNetModel = ref_module("controlnet").ControlNetModel
    models.Transformer2DModel = ref_module("transformer_2d").Transformer2DModel
    models.AsymmetricAutoencoderKL = type("AsymmetricAutoencoderKL", (), {})      # isinstance checks of the inpaint pipeline only
    scheds.LCMScheduler = ref_module("scheduling_lcm", "schedulers").LCMScheduler
    sdpkg = sys.modules[f"{PKG}.pipelines.stable_diffusion"]
    sdpkg.StableDiffusionSafetyChecker = sys.modules[f"{PKG}.pipelines.stable_diffusion.safety_checker"].StableDiffusionSafetyChecker
    sdpkg.StableDiffusionPipelineOutput = importlib.import_module(f"{PKG}.pipelines.stable_diffusion.pipeline_output").StableDiffusionPipelineOutput
    return importlib.import_module(f"{PKG}.{package}.{module}")


def _stub_text_encoders(P, mod):
    """what transformers/clip/modeling.py and transformers/t5/modeling.py import besides paddle: PaddleNLP's activation table,
    output containers and PretrainedModel / PretrainedConfig plumbing (checkpoint I/O, name mappings: dropped)"""
    import dataclasses

    F = P.nn.functional

    def quick_gelu(x):
        return x * F.sigmoid(1.702 * x)

    def gelu_new(x):   # paddlenlp/transformers/activations.py NewGELUActivation: the tanh approximation
        return F.gelu(x, approximate=True)

    act = {"quick_gelu": quick_gelu, "gelu": F.gelu, "gelu_new": gelu_new, "relu": F.relu, "silu": F.silu, "swish": F.silu}
    mod("paddlenlp")
    mod("paddlenlp.transformers")
    mod("paddlenlp.utils")
    mod("paddlenlp.transformers.activations", ACT2FN=act)

    class ModelOutput(dict):
        """paddlenlp model_outputs.ModelOutput: dataclass fields, None entries dropped from the tuple / dict view"""
        def __post_init__(self):
            for f in dataclasses.fields(self):
                v = getattr(self, f.name)
                if v is not None:
                    dict.__setitem__(self, f.name, v)

        def __getitem__(self, k):
            return dict.__getitem__(self, k) if isinstance(k, str) else self.to_tuple()[k]

        def to_tuple(self):
            return tuple(dict.values(self))

    def output_class(name, fields):
        return dataclasses.make_dataclass(name, [(f, object, None) for f in fields], bases=(ModelOutput,), eq=False)

    outs = {n: output_class(n, f) for n, f in {
        "BaseModelOutput": ("last_hidden_state", "hidden_states", "attentions"),
        "BaseModelOutputWithPooling": ("last_hidden_state", "pooler_output", "hidden_states", "attentions"),
        "BaseModelOutputWithPastAndCrossAttentions": ("last_hidden_state", "past_key_values", "hidden_states", "attentions", "cross_attentions"),
    }.items()}
    for n in ("Seq2SeqLMOutput", "Seq2SeqModelOutput", "Seq2SeqQuestionAnsweringModelOutput", "Seq2SeqSequenceClassifierOutput"):
        outs[n] = output_class(n, ("loss", "logits"))
    mod("paddlenlp.transformers.model_outputs", ModelOutput=ModelOutput, **outs)
    mod("paddlenlp.utils.converter", StateDictNameMapping=object)
    mod("paddlenlp.transformers.conversion_utils", StateDictNameMapping=object, init_name_mappings=lambda *a, **k: None,
        split_or_merge_func=lambda *a, **k: None)
    mod("paddlenlp.transformers.model_utils", register_base_model=lambda cls: cls)

    class PretrainedConfig:
        """the attribute bag the model constructors read (configuration_utils.PretrainedConfig, minus I/O)"""
        def __init__(self, **kw):
            base = dict(output_attentions=False, output_hidden_states=False, use_return_dict=True, return_dict=True, use_cache=False,
                        tensor_parallel_degree=1, pad_token_id=None, bos_token_id=None, eos_token_id=None, is_encoder_decoder=False,
                        is_decoder=False, tie_word_embeddings=True, recompute=False)
            base.update(kw)
            for k, v in base.items():
                if not hasattr(type(self), k) or not isinstance(getattr(type(self), k), property):
                    setattr(self, k, v)

    class PretrainedModel(P.nn.Layer):
        config_class = PretrainedConfig
        base_model_prefix = ""

        def __init__(self, config=None, *a, **k):
            super().__init__()
            self.config = config

        def post_init(self):
            pass

        def init_weights(self, *a, **k):
            pass

        @property
        def dtype(self):
            return P.float32

        def get_extended_attention_mask(self, attention_mask, input_shape, dtype=None):
            """transformers/model_utils.py:130-174, encoder branch: [B, S] ones / zeros -> additive [B, 1, 1, S]"""
            assert attention_mask.ndim == 2 and not self.config.is_decoder
            m = attention_mask[:, None, None, :].cast(P.float32)
            return (1.0 - m) * P.finfo(P.float32).min

    mod(f"{PKG}.transformers.model_utils", PretrainedModel=PretrainedModel, PretrainedConfig=PretrainedConfig, ALL_LAYERNORM_LAYERS=[])


_installed = None


def install():
    """Put the shim `paddle` and the stubbed `ppdiffusers` plumbing into sys.modules; returns the shim's module dict."""
    global _installed
    if _installed is not None:
        return _installed
    if not available():
        raise RuntimeError("oracle.reference_runner: /root/reference is not present (build container only)")
    if "paddle" in sys.modules and not getattr(sys.modules["paddle"], "__version__", "").endswith("torch-shim"):
        raise RuntimeError("a real paddle is importable here: run the reference directly instead")
    from . import paddle_shim
    shim = paddle_shim.build_modules()
    sys.modules.update(shim)
    root = types.ModuleType(PKG)
    root.__path__ = [os.path.join(REF_ROOT, PKG)]
    sys.modules[PKG] = root
    for sub in ("models", "schedulers"):                       # bare packages: the real files import from here on demand,
        m = types.ModuleType(f"{PKG}.{sub}")                   # the reference's own __init__.py (hub / lazy-import plumbing) never runs
        m.__path__ = [os.path.join(REF_ROOT, PKG, sub)]
        sys.modules[f"{PKG}.{sub}"] = m
        setattr(root, sub, m)
    stubs = _stub_modules(shim)
    sys.modules.update(stubs)
    for name, m in stubs.items():
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, m)
    _installed = shim
    return shim


def ref_module(name: str, package: str = "models"):
    """import ppdiffusers.<package>.<name> from the reference tree (the real, unmodified file)"""
    install()
    return importlib.import_module(f"{PKG}.{package}.{name}")


def to_shim(x):
    from . import paddle_shim
    import torch
    if x is None or isinstance(x, paddle_shim.Tensor):
        return x
    if isinstance(x, torch.Tensor):
        return paddle_shim.Tensor(x)
    if isinstance(x, dict):
        return {k: to_shim(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(to_shim(v) for v in x)
    return x


def from_shim(x):
    from . import paddle_shim
    if isinstance(x, paddle_shim.Tensor):
        return x.t
    if isinstance(x, (list, tuple)):
        return type(x)(from_shim(v) for v in x)
    return x


def load_params(layer, params: dict, computed=()):
    """set_state_dict that insists the oracle's parameter dictionary and the reference layer's agree name for name, shape for shape.
    `computed`: state-dict entries of the reference that are not free parameters of the oracle (persistable buffers the reference
    derives from the config, zero-initialised biases no checkpoint carries) -- named explicitly by the caller, left as constructed."""
    own = layer.state_dict()
    missing = sorted(k for k in own if k not in params and not any(k.endswith(c) for c in computed))
    unexpected = sorted(k for k in params if k not in own)
    if missing or unexpected:
        raise KeyError(f"parameter names differ from the reference's: missing {missing[:8]} ({len(missing)}), unexpected {unexpected[:8]} ({len(unexpected)})")
    for k, v in params.items():
        own[k].set_value(v.detach().float())
    return layer


def build_unet(config: dict, params: dict):
    """the reference's UNet2DConditionModel (models/unet_2d_condition.py:66) with the oracle's parameters, in eval mode"""
    m = ref_module("unet_2d_condition")
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in config.items()}
    for k in ("down_block_types", "up_block_types", "block_out_channels"):
        if k in cfg:
            cfg[k] = tuple(cfg[k])
    net = m.UNet2DConditionModel(**cfg)
    net.eval()
    return load_params(net, params)
