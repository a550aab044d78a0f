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
