"""Sharding of the attention path across the GPUs of one node.

The path has no exchange step: every (batch, kv-head) unit is independent in
forward and backward, so N ranks split the kv-heads (keeping every q-head of a kv-head on the
same rank, so dK/dV need no cross-rank sum) and, when there are fewer kv-heads than ranks, the
batch.  No collective is involved; a caller that wants the gathered output all-gathers `out`.
"""
from typing import Tuple


def shard_units(batch: int, nheads_q: int, nheads_k: int, world: int, rank: int) -> Tuple[slice, slice, slice]:
    """Return (batch_slice, q_head_slice, kv_head_slice) owned by `rank`.

    kv-heads are split first (contiguous ranges); if `world` does not divide nheads_k but does
    divide nheads_k * k for some batch split, the remaining factor splits the batch."""
    assert nheads_q % nheads_k == 0 and 0 <= rank < world
    group = nheads_q // nheads_k
    import math
    gh = math.gcd(world, nheads_k)           # ranks along the head axis
    gb = world // gh                         # ranks along the batch axis
    if batch % gb != 0:
        raise ValueError(f"cannot shard batch={batch}, kv-heads={nheads_k} over {world} ranks")
    rh, rb = rank % gh, rank // gh
    hk0, hk1 = rh * (nheads_k // gh), (rh + 1) * (nheads_k // gh)
    b0, b1 = rb * (batch // gb), (rb + 1) * (batch // gb)
    return slice(b0, b1), slice(hk0 * group, hk1 * group), slice(hk0, hk1)


def shard_alibi(alibi_slopes, q_heads: slice, batch: slice):
    """ALiBi slopes travel with their heads ([H] or [B, H])."""
    if alibi_slopes is None:
        return None
    if alibi_slopes.dim() == 1:
        return alibi_slopes[q_heads].contiguous()
    return alibi_slopes[batch, q_heads].contiguous()


def merge_attention_shards(outs, lses):
    """Combine attention computed over DISJOINT key shards into attention over their union.

    outs[s]: (B, Sq, H, D) output of flash_attn_func(q, k_s, v_s, ..., return_attn_probs=True)
    lses[s]: (B, H, Sq) fp32 log-sum-exp of the same call (natural log; -inf for rows that saw no key).
    Returns (out, lse):  LSE = logsumexp_s(lse_s),  out = sum_s exp(lse_s - LSE) out_s  (fp32 math, out in
    outs[0].dtype).  This is the per-step reduction of context / ring parallelism (SURVEY.md section 8(f)
    row 4): every rank attends its local K/V shard and the partial results are merged with the returned LSE;
    the exchange itself (all-gather or ring send/recv of K/V or of (out, lse)) belongs to the caller - the
    attention path of this package has no collective.
    """
    import torch
    lse = torch.stack([l.float() for l in lses], 0)                       # [n, B, H, Sq]
    m = lse.max(0).values
    m_safe = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
    w = torch.exp(lse - m_safe)                                           # -inf -> 0
    den = w.sum(0)
    tot = torch.where(den > 0, m_safe + torch.log(torch.where(den > 0, den, torch.ones_like(den))),
                      torch.full_like(den, float("-inf")))
    w = w / torch.where(den > 0, den, torch.ones_like(den))
    out = sum(o.float() * w[i].transpose(1, 2).unsqueeze(-1) for i, o in enumerate(outs))
    return out.to(outs[0].dtype), tot


def _local_fwd(q, k, v, window, scale):
    """Local attention of one rank: all queries over its key shard, the shard's causal offset as a right window.  That window
    is >= the shard's length on every rank but the last, where the public API - like the reference, fused_mha_forward.cu:351-352
    - would drop it; the private keep_window route (FA_FLAG_KEEP_WINDOW) keeps it."""
    from . import flash_attn_interface as fi
    o, lse, *_ = fi._dense_forward(q, k, v, 0.0, scale, False, window, 0.0, None, False, keep_window=True)
    return o, lse


def _local_bwd(dout, q, k, v, out, lse, window, scale):
    """fa_bwd of the LOCAL problem (q over this rank's key shard) with the GLOBAL out / lse: P = exp(s - LSE_global) and
    D = rowsum(dO o O_global) are then the true probabilities / row-dots of the full problem, so dk, dv of the shard are
    complete and dq is this shard's share of the sum over keys."""
    import torch
    from . import flash_attn_interface as fi
    d = q.shape[-1]
    dpad = (d + 7) // 8 * 8
    q_, k_, v_, o_ = (fi._prep(t, dpad) for t in (q, k, v, out))
    dq_, dk_, dv_ = (fi._prep(torch.empty_like(t), dpad) for t in (q_, k_, v_))
    fi._dense_backward(dout, q_, k_, v_, o_, lse.contiguous(), None, 0.0, d ** -0.5 if scale is None else scale, False,
                       window, 0.0, None, dq_, dk_, dv_, keep_window=True)
    return dq_[..., :d], dk_[..., :d], dv_[..., :d]


def _cp_function():
    import torch
    import torch.distributed as dist

    class ContextParallelAttnFunc(torch.autograd.Function):
        """forward: local attention, all-gather of (out, lse), LSE merge.  backward: the local backward kernels with the
        merged out / lse (dk, dv of the shard complete; no exchange), then ONE all-reduce (sum, fp32) of the partial dq."""

        @staticmethod
        def forward(ctx, q, k_shard, v_shard, group, window, scale, attn_fn, bwd_fn):
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            out, lse = attn_fn(q, k_shard, v_shard, window, scale)
            if world > 1:
                outs = [torch.empty_like(out) for _ in range(world)]
                lses = [torch.empty_like(lse) for _ in range(world)]
                dist.all_gather(outs, out.contiguous(), group=group)
                dist.all_gather(lses, lse.contiguous(), group=group)
                out, lse = merge_attention_shards(outs, lses)
            ctx.save_for_backward(q, k_shard, v_shard, out, lse)
            ctx.group, ctx.window, ctx.scale, ctx.bwd_fn, ctx.world = group, window, scale, bwd_fn, world
            ctx.mark_non_differentiable(lse)
            return out, lse

        @staticmethod
        def backward(ctx, dout, dlse):
            q, k, v, out, lse = ctx.saved_tensors
            dq, dk, dv = ctx.bwd_fn(dout.contiguous(), q, k, v, out, lse, ctx.window, ctx.scale)
            if ctx.world > 1:
                dq32 = dq.float().contiguous()                          # partial sums over key shards: add in fp32
                dist.all_reduce(dq32, op=dist.ReduceOp.SUM, group=ctx.group)
                dq = dq32.to(q.dtype)
            return dq, dk, dv, None, None, None, None, None

    return ContextParallelAttnFunc


def context_parallel_attention(q, k_shard, v_shard, group=None, causal=False, softmax_scale=None, attn_fn=None,
                               bwd_fn=None):
    """Attention over keys / values that are SHARDED along the sequence across the ranks of `group`
    (context parallelism; the consumers SURVEY.md section 8(f) row 4 cites: flash_attn_interface.py:17-112 - an
    autograd Function, i.e. it trains -, utils/benchmarks/benchmark_unsloth.py:19-39).

    Every rank holds all queries `q` (B, Sq, H, D) and one contiguous, equally sized shard of the keys / values
    `k_shard`, `v_shard` (B, Sk / N, Hk, D), rank r owning keys [r Sk/N, (r+1) Sk/N).  Forward: each rank runs ONE local
    attention call, the partial (out, lse) pairs are all-gathered (RCCL over xGMI under the "nccl" backend) and merged
    with merge_attention_shards.  causal=True is the bottom-right aligned causal mask of the GLOBAL problem: in rank r's
    local coordinates that is a right window of (N - 1 - r) Sk/N keys, which the kernels take as window_size=(-1, wr) -
    no mask tensor is ever built.

    Backward (autograd): each rank runs the local backward kernels over its shard with the MERGED out / lse - dk_shard
    and dv_shard are then complete without any exchange - and the partial dq (sum over this rank's keys) is all-reduced
    once in fp32.  `dout` is the gradient of the replicated output (every rank passes the same values, as every rank
    holds the same q and out).

    attn_fn(q, k, v, window_size, softmax_scale) -> (out, lse) and bwd_fn(dout, q, k, v, out, lse, window_size,
    softmax_scale) -> (dq, dk, dv) replace the local kernels (tests on CPU); the defaults are the HIP path.
    Returns (out, lse) of the full problem on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    skl = k_shard.shape[1]
    window = (-1, (world - 1 - rank) * skl) if causal else (-1, -1)
    return _cp_function().apply(q, k_shard, v_shard, group, window, softmax_scale, attn_fn or _local_fwd,
                                bwd_fn or _local_bwd)
