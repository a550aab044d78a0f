"""Dense attention forward/backward restatement (TEST INFRASTRUCTURE ONLY).

CPU (numpy, fp64) restatement of the reference's dense path in the extension-level
layout q [B,Hq,Sq,D], k/v [B,Hk,Sk,D]:
  * host-side flag normalisation      kernel/fused_mha_forward.cu:343-352
  * scores, masks, ALiBi, softcap     include/mat_mul.h:82-157 (bottom-right aligned:
                                      j' = j - (Sk - Sq), include/template.h:87)
  * online softmax, LSE, dropout      include/softmax.h:41-203,
                                      kernel/fused_mha_forward.cu:215-223
  * backward                          include/product.h:72-94, include/softmax.h:282-314,
                                      kernel/fused_mha_backward.cu:168-242,351-474
The oracle computes the mathematically exact result (fp64) of the 16-bit inputs; the
finite NEG_INF=-1e30 of include/kernel.h:20 is treated as -inf.

Documented divergences from the reference (SURVEY.md 8a "quirks"):
  * rows with no visible key give O = 0 and LSE = -inf (the reference gives uniform
    attention inside a processed tile, or O=0/LSE=-1e30 on skipped tiles);
  * any Sk is accepted (the reference's dense path needs Sk % 4 == 0);
  * softmax_d (rowsum(O*dO)) is returned for the dense path too (the reference leaves
    it uninitialised, kernel/fused_mha_backward.cu:692).
"""
import numpy as np

from .philox import dropout_keep_mask

try:  # torch is only used for bf16/fp16 rounding helpers
    import torch
except Exception:  # pragma: no cover
    torch = None


def round_to(x, dtype):
    """Round an fp64/fp32 array to `dtype` ('fp16' | 'bf16' | 'fp32' | 'fp64') and return fp64."""
    x = np.asarray(x, dtype=np.float64)
    if dtype in ("fp64", None):
        return x
    if dtype == "fp32":
        return x.astype(np.float32).astype(np.float64)
    if dtype == "fp16":
        return x.astype(np.float32).astype(np.float16).astype(np.float64)
    if dtype == "bf16":
        t = torch.from_numpy(np.ascontiguousarray(x.astype(np.float32)))
        return t.to(torch.bfloat16).to(torch.float64).numpy()
    raise ValueError(dtype)


def normalize_flags(seqlen_q, seqlen_k, causal, window_left, window_right, has_alibi,
                    kvcache=False, keep_window=False):
    """kernel/fused_mha_forward.cu:343-352 (dense), fused_mha_forward_varlen.cu:425,481-482
    (varlen: pass max_seqlen_*), fused_mha_forward_kvcache.cu:465-466,597-598 (kvcache).
    keep_window: not the reference - the product's FA_FLAG_KEEP_WINDOW (include/fa_mi355.h), used by its context-parallel
    wrapper only: a right window of >= seqlen_k keys is dropped only where it hides nothing (seqlen_q > seqlen_k: key
    j' > i + wr is hidden and j' reaches seqlen_q - 1)."""
    if seqlen_q == 1 and not has_alibi:
        causal = False
    if kvcache and causal:
        window_right = 0
    if window_left >= seqlen_k:
        window_left = -1
    if window_right >= seqlen_k and (not keep_window or window_right >= seqlen_q - 1):
        window_right = -1
    return causal, window_left, window_right


def visible_mask(seqlen_q, seqlen_k, causal, window_left, window_right):
    """Boolean [Sq, Sk]: True where key j is visible to query i.  include/mat_mul.h:92-106."""
    i = np.arange(seqlen_q)[:, None]
    jp = np.arange(seqlen_k)[None, :] - (seqlen_k - seqlen_q)
    vis = np.ones((seqlen_q, seqlen_k), dtype=bool)
    if causal:
        vis &= ~(jp > i)
    if window_left >= 0:
        vis &= ~(jp < i - window_left)
    if window_right >= 0:
        vis &= ~(jp > i + window_right)
    return vis


def score_matrix(q, k, scale, causal, window_left, window_right, softcap, slope):
    """One head: q [Sq,D], k [Sk,D] (fp64).  Returns (s, vis): s = capped/biased scaled
    scores with -inf where masked.  ALiBi THEN softcap - the reference's order,
    include/mat_mul.h:113-116."""
    sq, sk = q.shape[0], k.shape[0]
    s = (q @ k.T) * scale
    vis = visible_mask(sq, sk, causal, window_left, window_right)
    if slope is not None:
        i = np.arange(sq)[:, None]
        jp = np.arange(sk)[None, :] - (sk - sq)
        s = s - float(slope) * np.abs(i - jp)
    if softcap and softcap > 0.0:
        s = softcap * np.tanh(s / softcap)
    s = np.where(vis, s, -np.inf)
    return s, vis


def _slope(alibi_slopes, b, h):
    if alibi_slopes is None:
        return None
    a = np.asarray(alibi_slopes, dtype=np.float64)
    return a[h] if a.ndim == 1 else a[b, h]


def attn_fwd(q, k, v, scale, causal=False, window=(-1, -1), softcap=0.0, alibi_slopes=None,
             dropout_p=0.0, seed=0, offset=0, out_dtype=None, normalize=True,
             return_p=False, keep_window=False):
    """q [B,Hq,Sq,D], k/v [B,Hk,Sk,D] -> out [B,Hq,Sq,D] fp64 (rounded to out_dtype if
    given), lse [B,Hq,Sq] fp32, keep-mask [Sq,Sk] or None.

    LSE_i = m_i + ln(sum_j exp(s_ij - m_i)) with the PRE-dropout sum
    (include/softmax.h:187, kernel/fused_mha_forward.cu:220-223)."""
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    B, Hq, Sq, D = q.shape
    Hk, Sk = k.shape[1], k.shape[2]
    wl, wr = window
    if normalize:
        causal, wl, wr = normalize_flags(Sq, Sk, causal, wl, wr, alibi_slopes is not None, keep_window=keep_window)
    group = Hq // Hk
    out = np.zeros((B, Hq, Sq, v.shape[3]), dtype=np.float64)
    lse = np.full((B, Hq, Sq), -np.inf, dtype=np.float64)
    keep = None
    if dropout_p > 0.0:
        keep = dropout_keep_mask(seed, offset, dropout_p, Sq, Sk)
    ps = [] if return_p else None
    for b in range(B):
        for h in range(Hq):
            g = h // group
            s, vis = score_matrix(q[b, h], k[b, g], scale, causal, wl, wr, softcap,
                                  _slope(alibi_slopes, b, h))
            if Sk == 0:
                continue
            m = np.max(s, axis=1, keepdims=True)
            m_safe = np.where(np.isfinite(m), m, 0.0)
            e = np.where(vis, np.exp(s - m_safe), 0.0)
            l = e.sum(axis=1, keepdims=True)
            has = l[:, 0] > 0
            p = np.where(l > 0, e / np.where(l > 0, l, 1.0), 0.0)
            pd = p
            if keep is not None:
                pd = np.where(keep, p / (1.0 - dropout_p), 0.0)
            out[b, h] = pd @ v[b, g]
            with np.errstate(divide="ignore"):
                lse[b, h] = np.where(has, m_safe[:, 0] + np.log(np.where(has, l[:, 0], 1.0)),
                                     -np.inf)
            if return_p:
                ps.append(p)
    if out_dtype is not None:
        out = round_to(out, out_dtype)
    res = (out, lse.astype(np.float32), keep)
    if return_p:
        res = res + (ps,)
    return res


def attn_bwd(dout, q, k, v, out, lse, scale, causal=False, window=(-1, -1), softcap=0.0,
             alibi_slopes=None, dropout_p=0.0, seed=0, offset=0, normalize=True, keep_window=False):
    """Returns dq [B,Hq,Sq,D], dk, dv [B,Hk,Sk,D] (fp64) and softmax_d [B,Hq,Sq].

      D_i   = sum_d O_id * dO_id   (from the saved 16-bit O)    include/product.h:72-94
      P_ij  = exp(s_ij - LSE_i), 0 where masked                  include/softmax.h:282-286
      dS_ij = (Pdrop_ij * dP_ij - P_ij * D_i) * scale            include/softmax.h:308-309
      softcap: dS *= 1 - (s/c)^2 with s the capped score         include/softmax.h:311-314
      dV_j = sum_h sum_i Pdrop_ij dO_i ; dK_j = sum_h sum_i dS_ij q_i ; dQ_i = sum_j dS_ij k_j
    """
    dout = np.asarray(dout, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    out = np.asarray(out, dtype=np.float64)
    lse = np.asarray(lse, dtype=np.float64)
    B, Hq, Sq, D = q.shape
    Hk, Sk = k.shape[1], k.shape[2]
    wl, wr = window
    if normalize:
        causal, wl, wr = normalize_flags(Sq, Sk, causal, wl, wr, alibi_slopes is not None, keep_window=keep_window)
    group = Hq // Hk
    dq = np.zeros_like(q)
    dk = np.zeros_like(k)
    dv = np.zeros_like(v)
    softmax_d = np.zeros((B, Hq, Sq), dtype=np.float64)
    keep = None
    if dropout_p > 0.0:
        keep = dropout_keep_mask(seed, offset, dropout_p, Sq, Sk)
    for b in range(B):
        for h in range(Hq):
            g = h // group
            s, vis = score_matrix(q[b, h], k[b, g], scale, causal, wl, wr, softcap,
                                  _slope(alibi_slopes, b, h))
            lse_bh = lse[b, h][:, None]
            lse_safe = np.where(np.isfinite(lse_bh), lse_bh, 0.0)
            p = np.where(vis & np.isfinite(lse_bh), np.exp(np.where(vis, s, 0.0) - lse_safe), 0.0)
            pd = p
            if keep is not None:
                pd = np.where(keep, p / (1.0 - dropout_p), 0.0)
            d_row = (out[b, h] * dout[b, h]).sum(axis=1, keepdims=True)
            softmax_d[b, h] = d_row[:, 0]
            dp = dout[b, h] @ v[b, g].T
            ds = (pd * dp - p * d_row) * scale
            if softcap and softcap > 0.0:
                ds = ds * (1.0 - (np.where(vis, s, 0.0) / softcap) ** 2)
            ds = np.where(vis, ds, 0.0)
            dv[b, g] += pd.T @ dout[b, h]
            dk[b, g] += ds.T @ q[b, h]
            dq[b, h] = ds @ k[b, g]
    return dq, dk, dv, softmax_d.astype(np.float32)
