"""Packed variable-length attention restatement (TEST INFRASTRUCTURE ONLY).

Follows include/template.h:55-69,199-242 (ragged offsets, LSE layout [Hq, Tq]),
kernel/fused_mha_forward_varlen.cu:184-199 (paged K/V via block_table),
kernel/fused_mha_forward_varlen.cu:425,481-482 (flag normalisation with max_seqlen_*),
kernel/fused_mha_backward_varlen.cu:286-288 (softmax_d [Hq, Tq] written).

Documented divergences: dropout replay uses N_glob = max_seqlen_k in BOTH directions
(the reference's bwd uses seqlen_k, kernel/fused_mha_backward_varlen.cu:260 - an
inconsistency with its own forward, fused_mha_forward_varlen.cu:235); empty key ranges
give O = 0 / LSE = -inf (reference: -1e30).
"""
import numpy as np

from .attention import normalize_flags, score_matrix, _slope
from .philox import dropout_keep_mask


def _gather_kv(kv, b, s0, length, block_table, h):
    """Return [length, D] rows of kv for batch b, kv-head h."""
    if block_table is None:
        return kv[s0:s0 + length, h]
    page = kv.shape[1]
    j = np.arange(length)
    blk = np.asarray(block_table)[b, j // page]
    return kv[blk, j % page, h]


def _seq_bounds(cu_q, cu_k, seqused_k, b, paged):
    q0, q1 = int(cu_q[b]), int(cu_q[b + 1])
    k0, k1 = int(cu_k[b]), int(cu_k[b + 1])
    sk = k1 - k0
    if seqused_k is not None:
        su = int(seqused_k[b])
        sk = min(sk, su) if su > 0 else 0
    return q0, q1 - q0, (0 if paged else k0), sk


def varlen_fwd(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, scale,
               causal=False, window=(-1, -1), softcap=0.0, alibi_slopes=None,
               seqused_k=None, block_table=None, dropout_p=0.0, seed=0, offset=0):
    """q [Tq,Hq,D]; k,v [Tk,Hk,D] or paged [nblk,page,Hk,D].
    Returns out [Tq,Hq,D] fp64, lse [Hq,Tq] fp32."""
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    Tq, Hq, D = q.shape
    Hk = k.shape[-2]
    group = Hq // Hk
    B = len(cu_seqlens_q) - 1
    wl, wr = window
    causal, wl, wr = normalize_flags(max_seqlen_q, max_seqlen_k, causal, wl, wr,
                                     alibi_slopes is not None)
    out = np.zeros((Tq, Hq, v.shape[-1]), dtype=np.float64)
    lse = np.full((Hq, Tq), -np.inf, dtype=np.float64)
    for b in range(B):
        q0, sq, k0, sk = _seq_bounds(cu_seqlens_q, cu_seqlens_k, seqused_k, b,
                                     block_table is not None)
        if sq == 0:
            continue
        keep = None
        if dropout_p > 0.0 and sk > 0:
            # include/softmax.h:97-104 with GLOBAL_ROW_OFFSET = cu_q[b] + row,
            # GLOBAL_N = max_seqlen_k (fused_mha_forward_varlen.cu:235)
            keep = dropout_keep_mask(seed, offset, dropout_p, sq, sk, row0=q0,
                                     n_glob=max_seqlen_k)
        for h in range(Hq):
            g = h // group
            kk = _gather_kv(k, b, k0, sk, block_table, g)
            vv = _gather_kv(v, b, k0, sk, block_table, g)
            if sk == 0:
                continue
            s, vis = score_matrix(q[q0:q0 + sq, h], kk, scale, causal, wl, wr, softcap,
                                  _slope(alibi_slopes, b, h))
            m = np.max(s, axis=1, keepdims=True)
            m_safe = np.where(np.isfinite(m), m, 0.0)
            e = np.where(vis, np.exp(s - m_safe), 0.0)
            l = e.sum(axis=1, keepdims=True)
            p = np.where(l > 0, e / np.where(l > 0, l, 1.0), 0.0)
            if keep is not None:
                p = np.where(keep, p / (1.0 - dropout_p), 0.0)
            out[q0:q0 + sq, h] = p @ vv
            has = l[:, 0] > 0
            lse[h, q0:q0 + sq] = np.where(
                has, m_safe[:, 0] + np.log(np.where(has, l[:, 0], 1.0)), -np.inf)
    return out, lse.astype(np.float32)


def varlen_bwd(dout, q, k, v, out, lse, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
               max_seqlen_k, scale, causal=False, window=(-1, -1), softcap=0.0,
               alibi_slopes=None, dropout_p=0.0, seed=0, offset=0):
    """Non-paged varlen backward.  Returns dq [Tq,Hq,D], dk, dv [Tk,Hk,D], softmax_d [Hq,Tq]."""
    dout = np.asarray(dout, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    out = np.asarray(out, dtype=np.float64)
    lse = np.asarray(lse, dtype=np.float64)
    Tq, Hq, D = q.shape
    Hk = k.shape[1]
    group = Hq // Hk
    B = len(cu_seqlens_q) - 1
    wl, wr = window
    causal, wl, wr = normalize_flags(max_seqlen_q, max_seqlen_k, causal, wl, wr,
                                     alibi_slopes is not None)
    dq = np.zeros_like(q)
    dk = np.zeros_like(k)
    dv = np.zeros_like(v)
    softmax_d = np.zeros((Hq, Tq), dtype=np.float64)
    for b in range(B):
        q0, sq, k0, sk = _seq_bounds(cu_seqlens_q, cu_seqlens_k, None, b, False)
        if sq == 0 or sk == 0:
            continue
        keep = None
        if dropout_p > 0.0:
            keep = dropout_keep_mask(seed, offset, dropout_p, sq, sk, row0=q0,
                                     n_glob=max_seqlen_k)
        for h in range(Hq):
            g = h // group
            qq, kk, vv = q[q0:q0 + sq, h], k[k0:k0 + sk, g], v[k0:k0 + sk, g]
            do, oo = dout[q0:q0 + sq, h], out[q0:q0 + sq, h]
            s, vis = score_matrix(qq, kk, scale, causal, wl, wr, softcap,
                                  _slope(alibi_slopes, b, h))
            l_ = lse[h, q0:q0 + sq][:, None]
            l_safe = np.where(np.isfinite(l_), l_, 0.0)
            p = np.where(vis & np.isfinite(l_), np.exp(np.where(vis, s, 0.0) - l_safe), 0.0)
            pd = p if keep is None else np.where(keep, p / (1.0 - dropout_p), 0.0)
            d_row = (oo * do).sum(axis=1, keepdims=True)
            softmax_d[h, q0:q0 + sq] = d_row[:, 0]
            dp = do @ vv.T
            ds = (pd * dp - p * d_row) * scale
            if softcap and softcap > 0.0:
                ds = ds * (1.0 - (np.where(vis, s, 0.0) / softcap) ** 2)
            ds = np.where(vis, ds, 0.0)
            dv[k0:k0 + sk, g] += pd.T @ do
            dk[k0:k0 + sk, g] += ds.T @ qq
            dq[q0:q0 + sq, h] = ds @ kk
    return dq, dk, dv, softmax_d.astype(np.float32)
