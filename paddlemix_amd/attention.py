"""The reference's two inner attention seams, MI355X-backed.

B3  ``scaled_dot_product_attention_`` -- same signature, layout and error behaviour as
    ``F.scaled_dot_product_attention_`` (PPD/patches/paddle_patch.py:414-529): q [B, Sq, h, d], k / v [B, Skv, h, d],
    additive ``attn_mask`` broadcastable to [B, h, Sq, Skv], returns [B, Sq, h, d]; ``attention_op`` selects the backend
    and an unknown name raises ``ValueError`` (:523-526).  The only backend here is the fused flash kernel behind
    ``mi355x_sd_sdpa`` ("mi355x"; ``None`` / "auto" resolve to it).
B2  ``MI355XAttnProcessor`` -- an object with the ``AttnProcessor.__call__(attn, hidden_states,
    encoder_hidden_states, attention_mask, temb, scale)`` contract (attention_processor.py:673-735; installed with
    ``Attention.set_processor`` :352-385 / ``UNet2DConditionModel.set_attn_processor``, unet_2d_condition.py:657-691).
    It reads the projection weights from the ``attn`` module it is handed (``to_q / to_k / to_v / to_out[0]``,
    ``heads``, ``scale``, ``group_norm``, ``residual_connection``, ``rescale_output_factor``) and runs QKV projection,
    attention, output projection (+ residual) through the C ABI.  ``Attention`` below is the minimal weight holder with
    exactly those attributes, for use without Paddle.

B2' ``MI355XJointAttnProcessor`` -- the SD3 form of the same seam: ``JointAttnProcessor2_5.__call__(attn, hidden_states,
    encoder_hidden_states, attention_mask)`` (attention_processor.py:909-983) returns ``(hidden_states, encoder_hidden_states)``:
    image-token and text-token projections (``to_q/k/v`` and ``add_q/k/v_proj``) written side by side into one joint
    [B, S_img + S_txt, .] buffer (the GEMMs' row remap: no concat), ONE attention over the joint sequence, the two output
    projections (``to_out[0]``; ``to_add_out`` unless ``context_pre_only``) reading their halves of it in place (no split).
    ``JointAttention`` is the matching weight holder.

These seams exist for drop-in use and parity testing; the fast path is the whole-UNet program (seam B1, unet.py), which
fuses QKV, batches the cross-attention K/V projections and replays one hipGraph per step.  No CPU fallback.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Mapping, Optional

import torch

from . import ops

Tensor = torch.Tensor
_OPS = (None, "auto", "mi355x")


def scaled_dot_product_attention_(query: Tensor, key: Tensor, value: Tensor, attn_mask: Optional[Tensor] = None,
                                  dropout_p: float = 0.0, is_causal: bool = False, scale: Optional[float] = None,
                                  training: bool = True, attention_op: Optional[str] = None) -> Tensor:
    if attention_op not in _OPS:
        raise ValueError("ppxformers's attention_op shoulde be in "   # message as at paddle_patch.py:523-526
                         f"{list(_OPS)}, but we got {attention_op}!")
    if dropout_p != 0.0 and training:
        raise NotImplementedError("attention dropout is a training feature; the MI355X path is inference only")
    if query.dim() != 4:
        raise ValueError("query: expected [batch, seq_len, num_heads, head_dim]")
    B, Sq, H, D = query.shape
    Skv = key.shape[1]
    bias = None
    if is_causal:
        causal = torch.triu(torch.full((Sq, Skv), -1e30, device=query.device, dtype=torch.float32), diagonal=1)
        attn_mask = causal[None, None] if attn_mask is None else attn_mask.float() + causal[None, None]
    if attn_mask is not None:
        m = attn_mask
        if m.dtype == torch.bool:   # boolean keep-mask -> additive (paddle_patch.py:452-455 semantics)
            m = torch.zeros(m.shape, device=m.device, dtype=torch.float32).masked_fill(~m, -1e30)
        while m.dim() < 4:
            m = m[None]
        if m.shape[-1] != Skv or any(s not in (1, f) for s, f in zip(m.shape[:3], (B, H, Sq))):
            raise ValueError(f"attn_mask of shape {tuple(attn_mask.shape)} is not broadcastable to {[B, H, Sq, Skv]}")
        bias = m.to(torch.float32).contiguous()
    to = lambda t: t.to(ops._lib.elem_dtype()).contiguous()  # noqa: E731
    out = ops.sdpa(to(query), to(key), to(value), bias=bias, scale=scale)
    return out.to(query.dtype)


class _Linear(SimpleNamespace):
    """weight in the Paddle layout [in, out] (what ``attn.to_q.weight`` is in the reference), optional bias"""


class Attention:
    """Weight holder with the attributes ``AttnProcessor`` reads (attention_processor.py:31-207): built from a state
    dict slice ``{to_q.weight, to_k.weight, to_v.weight, to_out.0.weight, to_out.0.bias, [to_q.bias, ...,
    group_norm.weight, group_norm.bias]}`` in Paddle layouts."""

    def __init__(self, params: Mapping[str, Tensor], heads: int, norm_num_groups: Optional[int] = None,
                 eps: float = 1e-5, residual_connection: bool = False, rescale_output_factor: float = 1.0,
                 device="cuda"):
        def lin(name):
            w = params[name + ".weight"].to(device=device, dtype=torch.float32)
            b = params.get(name + ".bias")
            # the device GEMM wants [N][K] bf16: transpose once here
            return _Linear(weight=w, bias=None if b is None else b.to(device=device, dtype=torch.float32).contiguous(),
                           w_nk=w.t().to(ops._lib.elem_dtype()).contiguous())

        self.to_q, self.to_k, self.to_v = lin("to_q"), lin("to_k"), lin("to_v")
        self.to_out = [lin("to_out.0"), None]   # [Linear, Dropout]
        self.heads = heads
        inner = self.to_q.weight.shape[1]
        self.scale = (inner // heads) ** -0.5
        self.group_norm = None
        if norm_num_groups is not None:
            self.group_norm = SimpleNamespace(num_groups=norm_num_groups, eps=eps,
                                              weight=params["group_norm.weight"].to(device=device, dtype=torch.float32).contiguous(),
                                              bias=params["group_norm.bias"].to(device=device, dtype=torch.float32).contiguous())
        self.norm_cross = None
        self.spatial_norm = None
        self.residual_connection = residual_connection
        self.rescale_output_factor = rescale_output_factor
        self.processor = MI355XAttnProcessor()

    def set_processor(self, processor) -> None:
        self.processor = processor

    def __call__(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        """attention_processor.py:587-630 with out_dim = 4: [B | B*heads, Sq | 1, Skv] -> [B, heads, Sq | 1, Skv']
        (including the reference's quirk of padding by `target_length` when the last dim differs)."""
        if attention_mask is None:
            return None
        m = attention_mask
        if m.shape[-1] != target_length:
            m = torch.nn.functional.pad(m, (0, target_length), value=0.0)
        if m.shape[0] < batch_size * self.heads:
            m = m.repeat_interleave(self.heads, dim=0)
        return m.reshape(batch_size, self.heads, -1, m.shape[-1])


class MI355XAttnProcessor:
    """B2: drop-in ``AttnProcessor`` (attention_processor.py:668-735) running on the MI355X kernels."""

    def __call__(self, attn, hidden_states: Tensor, encoder_hidden_states: Optional[Tensor] = None,
                 attention_mask: Optional[Tensor] = None, temb: Optional[Tensor] = None, scale: float = 1.0, **kwargs):
        if getattr(attn, "spatial_norm", None) is not None or getattr(attn, "norm_cross", None):
            raise NotImplementedError("spatial_norm / norm_cross attention variants are not implemented")
        if scale != 1.0:
            raise NotImplementedError("LoRA scale != 1.0")
        residual = hidden_states
        in_dtype = hidden_states.dtype
        input_ndim = hidden_states.dim()
        if input_ndim == 4:
            B, C, Hh, Ww = hidden_states.shape
            hidden_states = hidden_states.reshape(B, C, Hh * Ww).transpose(1, 2)
        B, Sq, C = hidden_states.shape
        x = hidden_states.to(ops._lib.elem_dtype()).contiguous()
        Skv = Sq if encoder_hidden_states is None else encoder_hidden_states.shape[1]
        mask4 = attn.prepare_attention_mask(attention_mask, Skv, B)
        if attn.group_norm is not None:
            gn = attn.group_norm
            x = ops.group_norm(x, gn.weight, gn.bias, gn.num_groups, gn.eps, silu=False)
        ctx = x if encoder_hidden_states is None else encoder_hidden_states.to(ops._lib.elem_dtype()).contiguous()
        q = ops.linear(x.reshape(B * Sq, C), attn.to_q.w_nk, attn.to_q.bias)
        k = ops.linear(ctx.reshape(B * Skv, ctx.shape[-1]), attn.to_k.w_nk, attn.to_k.bias)
        v = ops.linear(ctx.reshape(B * Skv, ctx.shape[-1]), attn.to_v.w_nk, attn.to_v.bias)
        inner = q.shape[-1]
        d = inner // attn.heads
        o = scaled_dot_product_attention_(q.reshape(B, Sq, attn.heads, d), k.reshape(B, Skv, attn.heads, d),
                                          v.reshape(B, Skv, attn.heads, d), attn_mask=mask4, scale=attn.scale)
        out = ops.linear(o.reshape(B * Sq, inner), attn.to_out[0].w_nk, attn.to_out[0].bias).reshape(B, Sq, -1)
        out = out.to(in_dtype)
        if input_ndim == 4:
            out = out.transpose(1, 2).reshape(B, C, Hh, Ww)
        if attn.residual_connection:
            out = out + residual
        return out / attn.rescale_output_factor


class JointAttention:
    """Weight holder with the attributes ``JointAttnProcessor2_5`` reads (attention_processor.py:96-207, 909-983): the image stream's
    ``to_q / to_k / to_v / to_out[0]``, the text stream's ``add_q_proj / add_k_proj / add_v_proj`` and, unless ``context_pre_only``,
    ``to_add_out``; ``heads``; built from a state-dict slice in Paddle layouts (Linear.weight [in, out])."""

    def __init__(self, params: Mapping[str, Tensor], heads: int, context_pre_only: bool = False, device="cuda"):
        def lin(name):
            w = params[name + ".weight"].to(device=device, dtype=torch.float32)
            b = params.get(name + ".bias")
            return _Linear(weight=w, bias=None if b is None else b.to(device=device, dtype=torch.float32).contiguous(),
                           w_nk=w.t().to(ops._lib.elem_dtype()).contiguous())

        self.to_q, self.to_k, self.to_v = lin("to_q"), lin("to_k"), lin("to_v")
        self.add_q_proj, self.add_k_proj, self.add_v_proj = lin("add_q_proj"), lin("add_k_proj"), lin("add_v_proj")
        self.to_out = [lin("to_out.0"), None]
        self.context_pre_only = context_pre_only
        self.to_add_out = None if context_pre_only else lin("to_add_out")
        self.heads = heads
        self.scale = (self.to_q.weight.shape[1] // heads) ** -0.5
        self.processor = MI355XJointAttnProcessor()

    def set_processor(self, processor) -> None:
        self.processor = processor

    def __call__(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)


class MI355XJointAttnProcessor:
    """B2, SD3 form: drop-in ``JointAttnProcessor2_5`` (attention_processor.py:909-983) on the MI355X kernels. Same contract -- returns
    ``(hidden_states, encoder_hidden_states)``, 4-D inputs are flattened and restored the way the reference does -- and the
    reference's own limits: no attention mask is applied by that processor (``attention_mask`` is accepted and must be None here: a
    mask the caller expects to be honoured is refused instead of silently ignored)."""

    def __call__(self, attn, hidden_states: Tensor, encoder_hidden_states: Tensor = None, attention_mask: Optional[Tensor] = None,
                 *args, **kwargs):
        if encoder_hidden_states is None:
            raise ValueError("JointAttnProcessor needs encoder_hidden_states (the text stream of the MMDiT block)")
        if attention_mask is not None:
            raise NotImplementedError("the reference's JointAttnProcessor2_5 applies no mask; pass attention_mask=None")
        ed = ops._lib.elem_dtype()
        in_dtype, ctx_dtype = hidden_states.dtype, encoder_hidden_states.dtype
        input_ndim, context_input_ndim = hidden_states.dim(), encoder_hidden_states.dim()
        if input_ndim == 4:
            B, C, Hh, Ww = hidden_states.shape
            hidden_states = hidden_states.reshape(B, C, Hh * Ww).transpose(1, 2)
        if context_input_ndim == 4:
            Bc, Cc, Hc, Wc = encoder_hidden_states.shape
            encoder_hidden_states = encoder_hidden_states.reshape(Bc, Cc, Hc * Wc).transpose(1, 2)
        B, S1, C = hidden_states.shape
        S2 = encoder_hidden_states.shape[1]
        x = hidden_states.to(ed).contiguous().reshape(B * S1, C)
        c = encoder_hidden_states.to(ed).contiguous().reshape(B * S2, encoder_hidden_states.shape[-1])
        inner = attn.to_q.w_nk.shape[0]
        d = inner // attn.heads
        S = S1 + S2
        # q / k / v of both streams side by side in ONE joint buffer: image rows of batch b at b * S, text rows behind them (the
        # C row remap of mi355x_sd_linear_ex = the reference's three concats, attention_processor.py:948-951)
        qkv = [torch.empty(B * S * inner, device=x.device, dtype=ed) for _ in range(3)]
        for buf, l_img, l_txt in zip(qkv, (attn.to_q, attn.to_k, attn.to_v), (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj)):
            ops.linear_ex(x, l_img.w_nk, l_img.bias, out=buf, c_rows_per_batch=S1, c_batch_stride=S * inner, M=B * S1)
            ops.linear_ex(c, l_txt.w_nk, l_txt.bias, out=buf[S1 * inner:], c_rows_per_batch=S2, c_batch_stride=S * inner, M=B * S2)
        q, k, v = (t.view(B, S, attn.heads, d) for t in qkv)
        o = scaled_dot_product_attention_(q, k, v, dropout_p=0.0, is_causal=False).reshape(B * S * inner)
        # the two output projections read their halves of the joint attention output in place (A row remap = the reference's split, :965-969)
        out = ops.linear_ex(o, attn.to_out[0].w_nk, attn.to_out[0].bias, a_rows_per_batch=S1, a_batch_stride=S * inner, M=B * S1)
        out = out.reshape(B, S1, -1).to(in_dtype)
        if not attn.context_pre_only:
            enc = ops.linear_ex(o[S1 * inner:], attn.to_add_out.w_nk, attn.to_add_out.bias, a_rows_per_batch=S2, a_batch_stride=S * inner,
                                M=B * S2).reshape(B, S2, -1).to(ctx_dtype)
        else:   # the reference returns the raw attention rows of the text stream in this case (attention_processor.py:975-976)
            enc = o.view(B, S, inner)[:, S1:].to(ctx_dtype)
        if input_ndim == 4:
            out = out.transpose(1, 2).reshape(B, C, Hh, Ww)
        if context_input_ndim == 4:
            enc = enc.transpose(1, 2).reshape(Bc, Cc, Hc, Wc)
        return out, enc
