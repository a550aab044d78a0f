"""torch.library registration of the five operators: `torch.ops.flash_attn_mi355.{fwd, bwd,
varlen_fwd, varlen_bwd, fwd_kvcache}`.

Counterpart of the reference's TorchBind block (kernel/fused_mha_api.cpp:308-358: `fwd`, `bwd`,
`varlen_fwd`, `varlen_bwd`, `fwd_kvcache` under `flash_attn_v100_cuda`).  The argument ORDER follows
the reference schemas (include/mha.h:27-41, :67-87, :116-139, :170-195, :224-245); the tensor
LAYOUT is the public Python one - (B, S, H, D) / (T, H, D) - because this build takes strides
and never permutes ("replicate the Python API, not the alias").
The optional in-place outputs of the reference (`out` of fwd / varlen_fwd, `dq`, `dk`, `dv` of bwd:
include/mha.h:31,73-75) live in the separate ops `fwd_out`, `varlen_fwd_out` and `bwd_out`, which MUTATE
caller-allocated tensors (a torch.library op may not return an alias of an input); the plain ops return fresh
tensors.  `varlen_fwd` carries the reference's op-level extras `seqused_k`, `leftpad_k`, `zero_tensors`,
`num_splits` (include/mha.h:116-139).  Every op has a fake (meta) implementation so the path is traceable by
torch.compile / FakeTensorMode, and `fwd` / `varlen_fwd` carry autograd formulas that call the
`bwd` ops.

Import this module to register the ops (flash_attn_mi355/__init__.py does NOT import it, so
that plain users of the functional API pay nothing).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import flash_attn_interface as _fi

_NS = "flash_attn_mi355"


def _rng_tensor(rng, device):
    seed = int(rng[0]) & 0xFFFFFFFFFFFFFFFF
    if seed >= 1 << 63:                         # int64 carrier: two's complement, lossless
        seed -= 1 << 64
    return torch.tensor([seed, int(rng[1])], dtype=torch.int64, device=device)


def _rng_tuple(rng_state: Optional[Tensor]):
    if rng_state is None:
        return None
    s = rng_state.tolist()                      # host sync, as the reference's rng_state.cpu()
    return (int(s[0]) & 0xFFFFFFFFFFFFFFFF, int(s[1]))


# ------------------------------------------------------------------------------------------
# dense
# ------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::fwd", mutates_args=(), device_types="cuda")
def fwd(q: Tensor, k: Tensor, v: Tensor, alibi_slopes: Optional[Tensor], p_dropout: float,
        softmax_scale: float, is_causal: bool, window_size_left: int, window_size_right: int,
        softcap: float, return_softmax: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    out, lse, dmask, _, rng, _ = _fi._dense_forward(
        q, k, v, p_dropout, softmax_scale, is_causal, (window_size_left, window_size_right), softcap,
        alibi_slopes, return_softmax)
    return out, lse, dmask, _rng_tensor(rng, q.device)


@fwd.register_fake
def _(q, k, v, alibi_slopes, p_dropout, softmax_scale, is_causal, window_size_left,
      window_size_right, softcap, return_softmax):
    B, M, H, D = q.shape
    N = k.shape[1]
    out = q.new_empty((B, M, H, D))
    lse = q.new_empty((B, H, M), dtype=torch.float32)
    dmask = q.new_empty((B, H, M, N) if (return_softmax and p_dropout > 0.0) else (0,))
    return out, lse, dmask, q.new_empty((2,), dtype=torch.int64)


@torch.library.custom_op(f"{_NS}::bwd", mutates_args=(), device_types="cuda")
def bwd(dout: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, softmax_lse: Tensor,
        alibi_slopes: Optional[Tensor], p_dropout: float, softmax_scale: float, is_causal: bool,
        window_size_left: int, window_size_right: int, softcap: float, deterministic: bool,
        rng_state: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    d = q.shape[-1]
    dpad = (d + 7) // 8 * 8
    q_, k_, v_, out_ = (_fi._prep(t, dpad) for t in (q, k, v, out))
    dq_, dk_, dv_ = torch.empty_like(q_), torch.empty_like(k_), torch.empty_like(v_)
    dq_, dk_, dv_ = (_fi._prep(t, dpad) for t in (dq_, dk_, dv_))
    softmax_d = _fi._dense_backward(dout, q_, k_, v_, out_, softmax_lse, alibi_slopes, p_dropout,
                                    softmax_scale, is_causal, (window_size_left, window_size_right),
                                    softcap, _rng_tuple(rng_state) if p_dropout > 0.0 else (0, 0),
                                    dq_, dk_, dv_, deterministic=deterministic)
    if dpad != d:
        dq_, dk_, dv_ = (t[..., :d].contiguous() for t in (dq_, dk_, dv_))
    return dq_, dk_, dv_, softmax_d


@bwd.register_fake
def _(dout, q, k, v, out, softmax_lse, alibi_slopes, p_dropout, softmax_scale, is_causal,
      window_size_left, window_size_right, softcap, deterministic, rng_state):
    B, M, H, _ = q.shape
    return (torch.empty_like(q), torch.empty_like(k), torch.empty_like(v),
            q.new_empty((B, H, M), dtype=torch.float32))


def _fwd_setup(ctx, inputs, output):
    (q, k, v, alibi_slopes, p_dropout, softmax_scale, is_causal, wl, wr, softcap, _) = inputs
    out, lse, _, rng_state = output
    ctx.save_for_backward(q, k, v, out, lse, rng_state, alibi_slopes)
    ctx.args = (p_dropout, softmax_scale, is_causal, wl, wr, softcap)


def _fwd_backward(ctx, dout, dlse, ddmask, drng):
    q, k, v, out, lse, rng_state, alibi_slopes = ctx.saved_tensors
    p_dropout, softmax_scale, is_causal, wl, wr, softcap = ctx.args
    dq, dk, dv, _ = bwd(dout, q, k, v, out, lse, alibi_slopes, p_dropout, softmax_scale, is_causal,
                        wl, wr, softcap, False, rng_state)
    return (dq, dk, dv) + (None,) * 8


fwd.register_autograd(_fwd_backward, setup_context=_fwd_setup)


# ------------------------------------------------------------------------------------------
# varlen
# ------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::varlen_fwd", mutates_args=(), device_types="cuda")
def varlen_fwd(q: Tensor, k: Tensor, v: Tensor, cu_seqlens_q: Tensor, cu_seqlens_k: Tensor,
               block_table: Optional[Tensor], alibi_slopes: Optional[Tensor], max_seqlen_q: int,
               max_seqlen_k: int, p_dropout: float, softmax_scale: float, is_causal: bool,
               window_size_left: int, window_size_right: int, softcap: float,
               return_softmax: bool, seqused_k: Optional[Tensor] = None, leftpad_k: Optional[Tensor] = None,
               zero_tensors: bool = False, num_splits: int = 0) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    if num_splits > 1:
        raise RuntimeError("num_splits > 1 not supported")          # fused_mha_forward_varlen.cu:422
    out, lse, dmask, _, rng, _ = _fi._varlen_forward(
        q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, p_dropout, softmax_scale,
        is_causal, (window_size_left, window_size_right), softcap, alibi_slopes, return_softmax,
        block_table, seqused_k=seqused_k, leftpad_k=leftpad_k, zero_tensors=zero_tensors)
    return out, lse, dmask, _rng_tensor(rng, q.device)


@varlen_fwd.register_fake
def _(q, k, v, cu_seqlens_q, cu_seqlens_k, block_table, alibi_slopes, max_seqlen_q, max_seqlen_k,
      p_dropout, softmax_scale, is_causal, window_size_left, window_size_right, softcap,
      return_softmax, seqused_k=None, leftpad_k=None, zero_tensors=False, num_splits=0):
    T, H, D = q.shape
    dmask = q.new_empty((T, H, max_seqlen_k) if (return_softmax and p_dropout > 0.0) else (0,))
    return (q.new_empty((T, H, D)), q.new_empty((H, T), dtype=torch.float32), dmask,
            q.new_empty((2,), dtype=torch.int64))


@torch.library.custom_op(f"{_NS}::varlen_bwd", mutates_args=(), device_types="cuda")
def varlen_bwd(dout: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, softmax_lse: Tensor,
               cu_seqlens_q: Tensor, cu_seqlens_k: Tensor, alibi_slopes: Optional[Tensor],
               max_seqlen_q: int, max_seqlen_k: int, p_dropout: float, softmax_scale: float,
               is_causal: bool, window_size_left: int, window_size_right: int, softcap: float,
               deterministic: bool, rng_state: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    d = q.shape[-1]
    dpad = (d + 7) // 8 * 8
    q_, k_, v_, out_ = (_fi._prep(t, dpad) for t in (q, k, v, out))
    dq_, dk_, dv_ = torch.empty_like(q_), torch.empty_like(k_), torch.empty_like(v_)
    dq_, dk_, dv_ = (_fi._prep(t, dpad) for t in (dq_, dk_, dv_))
    cu_q = cu_seqlens_q.to(torch.int32).contiguous()
    cu_k = cu_seqlens_k.to(torch.int32).contiguous()
    softmax_d = _fi._varlen_backward(dout, q_, k_, v_, out_, softmax_lse, cu_q, cu_k, alibi_slopes,
                                     max_seqlen_q, max_seqlen_k, p_dropout, softmax_scale, is_causal,
                                     (window_size_left, window_size_right), softcap,
                                     _rng_tuple(rng_state) if p_dropout > 0.0 else (0, 0),
                                     dq_, dk_, dv_, deterministic=deterministic)
    if dpad != d:
        dq_, dk_, dv_ = (t[..., :d].contiguous() for t in (dq_, dk_, dv_))
    return dq_, dk_, dv_, softmax_d


@varlen_bwd.register_fake
def _(dout, q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, alibi_slopes, max_seqlen_q,
      max_seqlen_k, p_dropout, softmax_scale, is_causal, window_size_left, window_size_right,
      softcap, deterministic, rng_state):
    T, H, _ = q.shape
    return (torch.empty_like(q), torch.empty_like(k), torch.empty_like(v),
            q.new_empty((H, T), dtype=torch.float32))


def _varlen_setup(ctx, inputs, output):
    (q, k, v, cu_q, cu_k, block_table, alibi_slopes, max_q, max_k, p_dropout, softmax_scale,
     is_causal, wl, wr, softcap, _, seqused_k, _leftpad, _zero, _splits) = inputs
    if block_table is not None:
        raise RuntimeError("backward through paged K/V (block_table) is not supported")
    if seqused_k is not None:
        raise RuntimeError("seqused_k is a forward-only argument (the reference's varlen_bwd has none)")
    out, lse, _, rng_state = output
    ctx.save_for_backward(q, k, v, out, lse, cu_q, cu_k, rng_state, alibi_slopes)
    ctx.args = (max_q, max_k, p_dropout, softmax_scale, is_causal, wl, wr, softcap)


def _varlen_backward_formula(ctx, dout, dlse, ddmask, drng):
    q, k, v, out, lse, cu_q, cu_k, rng_state, alibi_slopes = ctx.saved_tensors
    max_q, max_k, p_dropout, softmax_scale, is_causal, wl, wr, softcap = ctx.args
    dq, dk, dv, _ = varlen_bwd(dout, q, k, v, out, lse, cu_q, cu_k, alibi_slopes, max_q, max_k,
                               p_dropout, softmax_scale, is_causal, wl, wr, softcap, False, rng_state)
    return (dq, dk, dv) + (None,) * 17


varlen_fwd.register_autograd(_varlen_backward_formula, setup_context=_varlen_setup)


# ------------------------------------------------------------------------------------------
# in-place forms: the reference's optional `out` / `dq` / `dk` / `dv` arguments (include/mha.h:31,73-75)
# ------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::fwd_out", mutates_args=("out",), device_types="cuda")
def fwd_out(q: Tensor, k: Tensor, v: Tensor, out: Tensor, alibi_slopes: Optional[Tensor], p_dropout: float,
            softmax_scale: float, is_causal: bool, window_size_left: int, window_size_right: int,
            softcap: float, return_softmax: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """writes `out` (shape / dtype of q, contiguous last dim); returns (softmax_lse, dmask, rng_state)"""
    _, lse, dmask, _, rng, _ = _fi._dense_forward(
        q, k, v, p_dropout, softmax_scale, is_causal, (window_size_left, window_size_right), softcap,
        alibi_slopes, return_softmax, out=out)
    return lse, dmask, _rng_tensor(rng, q.device)


@fwd_out.register_fake
def _(q, k, v, out, alibi_slopes, p_dropout, softmax_scale, is_causal, window_size_left, window_size_right,
      softcap, return_softmax):
    B, M, H, _ = q.shape
    dmask = q.new_empty((B, H, M, k.shape[1]) if (return_softmax and p_dropout > 0.0) else (0,))
    return q.new_empty((B, H, M), dtype=torch.float32), dmask, q.new_empty((2,), dtype=torch.int64)


@torch.library.custom_op(f"{_NS}::varlen_fwd_out", mutates_args=("out",), device_types="cuda")
def varlen_fwd_out(q: Tensor, k: Tensor, v: Tensor, out: Tensor, cu_seqlens_q: Tensor, cu_seqlens_k: Tensor,
                   seqused_k: Optional[Tensor], leftpad_k: Optional[Tensor], block_table: Optional[Tensor],
                   alibi_slopes: Optional[Tensor], max_seqlen_q: int, max_seqlen_k: int, p_dropout: float,
                   softmax_scale: float, zero_tensors: bool, is_causal: bool, window_size_left: int,
                   window_size_right: int, softcap: float, return_softmax: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """the reference's argument order (include/mha.h:116-139) with a caller-allocated `out`"""
    _, lse, dmask, _, rng, _ = _fi._varlen_forward(
        q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, p_dropout, softmax_scale,
        is_causal, (window_size_left, window_size_right), softcap, alibi_slopes, return_softmax,
        block_table, seqused_k=seqused_k, leftpad_k=leftpad_k, zero_tensors=zero_tensors, out=out)
    return lse, dmask, _rng_tensor(rng, q.device)


@varlen_fwd_out.register_fake
def _(q, k, v, out, cu_seqlens_q, cu_seqlens_k, seqused_k, leftpad_k, block_table, alibi_slopes, max_seqlen_q,
      max_seqlen_k, p_dropout, softmax_scale, zero_tensors, is_causal, window_size_left, window_size_right,
      softcap, return_softmax):
    T, H, _ = q.shape
    dmask = q.new_empty((T, H, max_seqlen_k) if (return_softmax and p_dropout > 0.0) else (0,))
    return q.new_empty((H, T), dtype=torch.float32), dmask, q.new_empty((2,), dtype=torch.int64)


@torch.library.custom_op(f"{_NS}::bwd_out", mutates_args=("dq", "dk", "dv"), device_types="cuda")
def bwd_out(dout: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, softmax_lse: Tensor,
            dq: Tensor, dk: Tensor, dv: Tensor, alibi_slopes: Optional[Tensor], p_dropout: float,
            softmax_scale: float, is_causal: bool, window_size_left: int, window_size_right: int,
            softcap: float, deterministic: bool, rng_state: Optional[Tensor]) -> Tensor:
    """writes caller-allocated dq / dk / dv (shapes of q / k / v); returns softmax_d"""
    d = q.shape[-1]
    dpad = (d + 7) // 8 * 8
    for name, g, ref in (("dq", dq, q), ("dk", dk, k), ("dv", dv, v)):
        if g.dtype != ref.dtype or tuple(g.shape) != tuple(ref.shape) or g.stride(-1) != 1:
            raise RuntimeError(f"{name} must have the dtype and shape of its input and a contiguous last dimension")
    q_, k_, v_, out_ = (_fi._prep(t, dpad) for t in (q, k, v, out))
    direct = dpad == d and all(t.data_ptr() % 16 == 0 and all(st % 8 == 0 for st in t.stride()[:-1]) for t in (dq, dk, dv))
    if direct:
        dq_, dk_, dv_ = dq, dk, dv
    else:
        dq_, dk_, dv_ = (_fi._prep(torch.empty_like(t), dpad) for t in (q_, k_, v_))
    softmax_d = _fi._dense_backward(dout, q_, k_, v_, out_, softmax_lse, alibi_slopes, p_dropout,
                                    softmax_scale, is_causal, (window_size_left, window_size_right),
                                    softcap, _rng_tuple(rng_state) if p_dropout > 0.0 else (0, 0),
                                    dq_, dk_, dv_, deterministic=deterministic)
    if not direct:
        dq.copy_(dq_[..., :d]); dk.copy_(dk_[..., :d]); dv.copy_(dv_[..., :d])
    return softmax_d


@bwd_out.register_fake
def _(dout, q, k, v, out, softmax_lse, dq, dk, dv, alibi_slopes, p_dropout, softmax_scale, is_causal,
      window_size_left, window_size_right, softcap, deterministic, rng_state):
    B, M, H, _ = q.shape
    return q.new_empty((B, H, M), dtype=torch.float32)


# ------------------------------------------------------------------------------------------
# kv-cache (mutates the caches in place, like the reference op)
# ------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::fwd_kvcache", mutates_args=("kcache", "vcache"), device_types="cuda")
def fwd_kvcache(q: Tensor, kcache: Tensor, vcache: Tensor, k: Optional[Tensor], v: Optional[Tensor],
                seqlens_k: Optional[Tensor], rotary_cos: Optional[Tensor], rotary_sin: Optional[Tensor],
                cache_batch_idx: Optional[Tensor], leftpad_k: Optional[Tensor],
                block_table: Optional[Tensor], alibi_slopes: Optional[Tensor], softmax_scale: float,
                is_causal: bool, window_size_left: int, window_size_right: int, softcap: float,
                is_rotary_interleaved: bool, num_splits: int) -> Tuple[Tensor, Tensor]:
    out, lse = _fi.flash_attn_with_kvcache(
        q, kcache, vcache, k=k, v=v, rotary_cos=rotary_cos, rotary_sin=rotary_sin,
        cache_seqlens=seqlens_k, cache_batch_idx=cache_batch_idx, cache_leftpad=leftpad_k,
        block_table=block_table, softmax_scale=softmax_scale, causal=is_causal,
        window_size=(window_size_left, window_size_right), softcap=softcap,
        rotary_interleaved=is_rotary_interleaved, alibi_slopes=alibi_slopes, num_splits=num_splits,
        return_softmax_lse=True)
    return out, lse


@fwd_kvcache.register_fake
def _(q, kcache, vcache, k, v, seqlens_k, rotary_cos, rotary_sin, cache_batch_idx, leftpad_k,
      block_table, alibi_slopes, softmax_scale, is_causal, window_size_left, window_size_right,
      softcap, is_rotary_interleaved, num_splits):
    B, T, H, _ = q.shape
    return torch.empty_like(q), q.new_empty((B, H, T), dtype=torch.float32)


__all__ = ["fwd", "bwd", "varlen_fwd", "varlen_bwd", "fwd_kvcache", "fwd_out", "varlen_fwd_out", "bwd_out"]
