"""ctypes binding of libmi355x_sd.so (C ABI in include/mi355x_sd.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))

# The library comes in two builds of the same sources (include/mi355x_sd.h, mi355x_sd_elem_dtype): bfloat16 elements
# (default; the dtype BASELINE.json's configurations name) and IEEE-half elements (MI355X_SD_DTYPE=fp16 or
# set_elem_dtype("fp16") before the first load: same speed, 3 more mantissa bits -> ~6x tighter parity). One element
# type per process.
_BUILDS = {"bf16": ("libmi355x_sd.so", 0), "fp16": ("libmi355x_sd_f16.so", 1)}
ELEM_NAME = os.environ.get("MI355X_SD_DTYPE", "bf16")
if ELEM_NAME not in _BUILDS:
    raise ValueError(f"MI355X_SD_DTYPE must be one of {sorted(_BUILDS)}, got {ELEM_NAME!r}")
LIB_PATH = os.path.join(_HERE, _BUILDS[ELEM_NAME][0])
# The production libraries never read the environment. The MI355X_SD_* A/B switches of the launchers (tile overrides, "the other
# way" forms the variant tests compare bit for bit: csrc/common.h sd_switch) exist in a third build only, bf16 elements with
# -DMI355X_SD_DEBUG_SWITCHES: MI355X_SD_LIB=dbg selects it (tests/test_gpu_switches.py, tests/test_gpu_gemm_variants.py, scripts/).
if os.environ.get("MI355X_SD_LIB") == "dbg":
    if ELEM_NAME != "bf16":
        raise ValueError("MI355X_SD_LIB=dbg: the debug-switch build exists for bf16 elements only")
    LIB_PATH = os.path.join(_HERE, "libmi355x_sd_dbg.so")

ABI_VERSION = 12
GEGLU, OUT_F32, SILU, GELU_TANH, PAD_BR, R_F32, CONV_KB64 = 1, 2, 4, 8, 16, 32, 64
UNET_ENC_MASK, UNET_SELF_MASK, UNET_CONTROLNET = 1, 2, 4   # mi355x_sd_unet_plan_ex flags
SDPA_LOG2 = 1
MOD_F32, MOD_ELEM = 0, 1

# name -> (restype, argtypes); must list every symbol include/mi355x_sd.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "mi355x_sd_abi_version": (c_int, []),
    "mi355x_sd_elem_dtype": (c_int, []),
    "mi355x_sd_last_error": (c_char_p, []),
    "mi355x_sd_init": (c_int, [c_int]),
    "mi355x_sd_program_load": (c_int, [c_char_p, POINTER(c_void_p)]),
    "mi355x_sd_program_destroy": (c_int, [c_void_p]),
    "mi355x_sd_program_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "mi355x_sd_program_num_launches": (c_int, [c_void_p]),
    "mi355x_sd_program_device_bytes": (c_int, [c_void_p, POINTER(ctypes.c_size_t)]),
    "mi355x_sd_program_bind": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "mi355x_sd_program_num_io": (c_int, [c_void_p]),
    "mi355x_sd_program_io_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int), POINTER(c_int64),
                                          POINTER(c_int), POINTER(ctypes.c_size_t), POINTER(c_void_p)]),
    "mi355x_sd_program_run": (c_int, [c_void_p, c_void_p]),
    # (GEMM-class calls, ABI 12: `ws, ws_bytes` = the call's split-K / widening scratch, before the stream)
    "mi355x_sd_linear": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_int, c_int, c_void_p, c_int, c_float, c_int, c_void_p, ctypes.c_size_t, c_void_p]),
    "mi355x_sd_linear_ex": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int,
                                    c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_float,
                                    c_int, c_void_p, ctypes.c_size_t, c_void_p]),
    "mi355x_sd_row_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "mi355x_sd_linear_ln": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_void_p, c_int, c_void_p, ctypes.c_size_t, c_void_p]),
    "mi355x_sd_linear_f8": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                    c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "mi355x_sd_adaln_f8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    "mi355x_sd_linear_f8_q": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p,
                                      c_int, c_int, c_int, c_void_p, c_float, c_int, c_void_p]),
    "mi355x_sd_quantize_rows": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "mi355x_sd_adaln": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_int,
                                c_void_p]),
    "mi355x_sd_adaln_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int,
                                   c_void_p]),
    "mi355x_sd_patchify": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mi355x_sd_unpatchify": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mi355x_sd_conv3x3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                  c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_int, c_void_p, ctypes.c_size_t,
                                  c_void_p]),
    "mi355x_sd_sdpa": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_int64, c_int, c_int64, c_int, c_int64, c_int, c_int64, c_int, c_int64, c_int64,
                               c_int64, c_float, c_void_p]),
    "mi355x_sd_sdpa_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_int64, c_int, c_int64, c_int, c_int64, c_int, c_int64, c_int, c_int64, c_int64,
                                  c_int64, c_float, c_int, c_void_p]),
    "mi355x_sd_sdpa_accum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_int64, c_int, c_int64, c_int, c_int64, c_int, c_int64, c_int, c_int64, c_int64,
                                     c_int64, c_float, c_float, c_void_p]),
    "mi355x_sd_groupnorm_workspace_floats": (c_int, [c_int, c_int, c_int]),
    "mi355x_sd_groupnorm_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    "mi355x_sd_scale_shift_act": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                          c_void_p]),
    "mi355x_sd_layernorm": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int,
                                    c_void_p]),
    # fp32-residual-stream forms (x_f32 = 1: the input rows are fp32)
    "mi355x_sd_groupnorm_act_fits": (c_int, [c_int, c_int, c_int]),
    "mi355x_sd_groupnorm_act": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                        c_void_p]),
    "mi355x_sd_groupnorm_stats_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_int, c_void_p]),
    "mi355x_sd_scale_shift_act_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                             c_int, c_void_p, c_int, c_void_p]),
    "mi355x_sd_layernorm_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int,
                                       c_int, c_void_p]),
    "mi355x_sd_cast_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "mi355x_sd_conv_in3x3_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_void_p]),
    "mi355x_sd_add_nchw_ex": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]),
    "mi355x_sd_fused_adaln_scale_residual": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                     c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                                     c_void_p]),
    "mi355x_sd_fused_adaln_scale_residual_ex": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                        c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                                        c_void_p]),
    "mi355x_sd_split_concat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mi355x_sd_timestep_embedding": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float,
                                             c_void_p, c_int, c_void_p]),
    "mi355x_sd_silu": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "mi355x_sd_conv_in3x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_void_p]),
    "mi355x_sd_conv_out3x3": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    "mi355x_sd_copy_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "mi355x_sd_add_nchw": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "mi355x_sd_latent_dist": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "mi355x_sd_embed_tokens": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "mi355x_sd_rmsnorm": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p]),
    "mi355x_sd_gated_activation": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "mi355x_sd_activation": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mi355x_sd_conv1x1_nchw": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64,
                                         c_void_p]),
    "mi355x_sd_softmax_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "mi355x_sd_axpby": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mi355x_sd_mask_to_bias": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mi355x_sd_cfg_axpby": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    # seam B1: the whole UNet behind a handle (csrc/unet_exec.hip)
    "mi355x_sd_unet_create": (c_int, [c_char_p, POINTER(c_void_p)]),
    "mi355x_sd_unet_destroy": (c_int, [c_void_p]),
    "mi355x_sd_unet_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "mi355x_sd_unet_num_params": (c_int, [c_void_p]),
    "mi355x_sd_unet_param_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int64), POINTER(c_int)]),
    "mi355x_sd_unet_load_weight": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int, c_int]),
    "mi355x_sd_unet_weight_bytes": (c_int, [c_void_p, POINTER(ctypes.c_size_t)]),
    "mi355x_sd_unet_finalize_weights": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "mi355x_sd_unet_pack_weights": (c_int, [c_void_p, c_void_p, ctypes.c_size_t]),
    "mi355x_sd_unet_attach_weights": (c_int, [c_void_p, c_void_p, ctypes.c_size_t]),
    "mi355x_sd_unet_packed_tensor": (c_int, [c_void_p, c_char_p, POINTER(ctypes.c_size_t), POINTER(ctypes.c_size_t), POINTER(c_int),
                                             POINTER(c_int)]),
    "mi355x_sd_unet_plan": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(ctypes.c_size_t)]),
    "mi355x_sd_unet_bind_workspace": (c_int, [c_void_p, c_void_p, ctypes.c_size_t]),
    "mi355x_sd_unet_num_launches": (c_int, [c_void_p]),
    "mi355x_sd_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int]),
    "mi355x_sd_unet_plan_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(ctypes.c_size_t)]),
    "mi355x_sd_unet_num_skips": (c_int, [c_void_p]),
    "mi355x_sd_unet_skip_shape": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "mi355x_sd_unet_set_input": (c_int, [c_void_p, c_char_p, c_void_p]),
    "mi355x_sd_unet_set_ip_adapter_scale": (c_int, [c_void_p, c_float]),
    "mi355x_sd_unet_forward_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, POINTER(c_void_p), c_int, c_void_p, c_void_p, c_int]),
    "mi355x_sd_graph_begin": (c_int, [c_void_p]),
    "mi355x_sd_graph_end": (c_int, [c_void_p, POINTER(c_void_p)]),
    "mi355x_sd_graph_launch": (c_int, [c_void_p, c_void_p]),
    "mi355x_sd_graph_destroy": (c_int, [c_void_p]),
    "mi355x_sd_probe_layouts": (c_int, [c_void_p, c_void_p]),
    # multi-GPU entry points (csrc/comm.hip: RCCL through dlopen)
    "mi355x_sd_comm_unique_id": (c_int, [c_void_p]),
    "mi355x_sd_comm_init": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "mi355x_sd_comm_broadcast": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p]),
    "mi355x_sd_comm_all_gather": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "mi355x_sd_comm_destroy": (c_int, [c_void_p]),
}

_lib = None


class MI355XError(RuntimeError):
    pass


def set_elem_dtype(name: str) -> None:
    """Select the library build ("bf16" | "fp16"). Must happen before the library is first loaded."""
    global ELEM_NAME, LIB_PATH
    if name not in _BUILDS:
        raise ValueError(f"element dtype must be one of {sorted(_BUILDS)}, got {name!r}")
    if _lib is not None and name != ELEM_NAME:
        raise MI355XError(f"library already loaded with {ELEM_NAME} elements; one element type per process")
    ELEM_NAME = name
    LIB_PATH = os.path.join(_HERE, _BUILDS[name][0])


def elem_dtype():
    """torch dtype of the 16-bit activations / weights of the selected build"""
    import torch
    return torch.float16 if ELEM_NAME == "fp16" else torch.bfloat16


def load() -> ctypes.CDLL:
    """Load the HIP library.  No fallback: a missing/unbuilt library is a hard error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MI355XError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C paddlemix_amd/csrc`). paddlemix_amd has no CPU / PyTorch fallback.")
    # torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so); the library links /opt/rocm's. Whichever is loaded
    # first becomes the process's runtime -- and if it is /opt/rocm's, torch's copy then finds no device ("no HIP device
    # visible" from mi355x_sd_init). The Python host shares device memory and streams with torch, so torch's goes first.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mi355x_sd_abi_version() != ABI_VERSION:
        raise MI355XError(f"ABI mismatch: library {lib.mi355x_sd_abi_version()} != binding {ABI_VERSION}")
    if lib.mi355x_sd_elem_dtype() != _BUILDS[ELEM_NAME][1]:
        raise MI355XError(f"{LIB_PATH} was not built for {ELEM_NAME} elements")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().mi355x_sd_last_error()
        raise MI355XError(f"libmi355x_sd error {rc}: {msg.decode() if msg else '?'}")
