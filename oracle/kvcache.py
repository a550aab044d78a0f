"""KV-cache attention (append + RoPE + paged KV) restatement (TEST INFRASTRUCTURE ONLY).

Follows kernel/fused_mha_forward_kvcache.cu:79-86,134-141,180-217,465-472,597-598 and
include/rotary.h:58-147 (cache append with RoPE on K), :176-262 (Q tile with RoPE).

  cb = cache_batch_idx[b] or b ; lp = cache_leftpad[b] or 0 ; L = cache_seqlens[b] or 0
  new rows r: pos = L + lp + r ; k_cache[cb,pos,g] = rope(k[b,r,g], pos) ; v_cache[...] = v
  S_k = L + T_new ; keys at k_cache[cb, lp + j, g] ; paged: block_table[b, (lp+j)//page]
  q rotated at L + lp + (i if causal-or-window else 0)      (rotary.h:177,201-202)

Documented divergence: L is indexed by b (upstream flash-attn semantics); the reference
indexes cache_seqlens by cb (fused_mha_forward_kvcache.cu:85) - identical whenever
cache_batch_idx is absent.  Optional fp8-e4m3 cache (k_descale/v_descale) is an
extension defined by this build: stored = round_e4m3(x / descale), read = stored*descale.
"""
import numpy as np

from .attention import normalize_flags, score_matrix, round_to, _slope


def apply_rope(x, cos, sin, pos, interleaved, io_dtype=None):
    """x [..., D] fp64; cos/sin [seqlen_ro, rd/2]; pos scalar.  include/rotary.h:91-141.
    Math in fp32-like precision (we use fp64), result rounded to io_dtype."""
    x = np.array(x, dtype=np.float64, copy=True)
    rd = 2 * cos.shape[1]
    c = np.asarray(cos[pos], dtype=np.float64)
    s = np.asarray(sin[pos], dtype=np.float64)
    if interleaved:
        x0 = x[..., 0:rd:2].copy()
        x1 = x[..., 1:rd:2].copy()
        x[..., 0:rd:2] = x0 * c - x1 * s
        x[..., 1:rd:2] = x0 * s + x1 * c
    else:
        x0 = x[..., : rd // 2].copy()
        x1 = x[..., rd // 2: rd].copy()
        x[..., : rd // 2] = x0 * c - x1 * s
        x[..., rd // 2: rd] = x0 * s + x1 * c
    return round_to(x, io_dtype) if io_dtype else x


def round_e4m3(x):
    """Round fp64 array to OCP fp8 e4m3fn (saturating at +-448), return fp64."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(np.clip(x, -448.0, 448.0).astype(np.float32)))
    return t.to(torch.float8_e4m3fn).to(torch.float64).numpy()


def kvcache_fwd(q, k_cache, v_cache, k=None, v=None, rotary_cos=None, rotary_sin=None,
                cache_seqlens=None, cache_batch_idx=None, cache_leftpad=None,
                block_table=None, scale=None, causal=False, window=(-1, -1), softcap=0.0,
                rotary_interleaved=True, alibi_slopes=None, io_dtype="fp16",
                k_descale=None, v_descale=None):
    """q [B,Tq,Hq,D]; caches [Bc,Smax,Hk,D] or paged [nblk,page,Hk,D] (fp64 arrays holding
    the stored values: 16-bit values, or fp8 codes' values when *_descale is given).
    Mutates k_cache / v_cache in place.  Returns out [B,Tq,Hq,D] fp64, lse [B,Hq,Tq] fp32."""
    q = np.asarray(q, dtype=np.float64)
    B, Tq, Hq, D = q.shape
    Hk = k_cache.shape[2]
    group = Hq // Hk
    paged = block_table is not None
    page = k_cache.shape[1]
    max_seqlen_k = (np.asarray(block_table).shape[1] * page) if paged else k_cache.shape[1]
    if scale is None:
        scale = D ** -0.5
    wl, wr = window
    causal, wl, wr = normalize_flags(Tq, max_seqlen_k, causal, wl, wr,
                                     alibi_slopes is not None, kvcache=True)
    local = causal or wl >= 0 or wr >= 0
    t_new = 0 if k is None else k.shape[1]
    out = np.zeros((B, Tq, Hq, D), dtype=np.float64)
    lse = np.full((B, Hq, Tq), -np.inf, dtype=np.float64)
    kd = 1.0 if k_descale is None else float(k_descale)
    vd = 1.0 if v_descale is None else float(v_descale)

    def slot(b, pos):
        if paged:
            return int(np.asarray(block_table)[b, pos // page]), pos % page
        cb = int(cache_batch_idx[b]) if cache_batch_idx is not None else b
        return cb, pos

    for b in range(B):
        lp = int(cache_leftpad[b]) if cache_leftpad is not None else 0
        L = int(cache_seqlens[b]) if cache_seqlens is not None else 0
        for r in range(t_new):
            pos = L + lp + r
            i0, i1 = slot(b, pos)
            kr = np.asarray(k[b, r], dtype=np.float64)
            if rotary_cos is not None:
                kr = apply_rope(kr, rotary_cos, rotary_sin, pos, rotary_interleaved, io_dtype)
            vr = np.asarray(v[b, r], dtype=np.float64)
            if k_descale is not None:
                kr = round_e4m3(kr / kd)
            if v_descale is not None:
                vr = round_e4m3(vr / vd)
            k_cache[i0, i1] = kr
            v_cache[i0, i1] = vr
        sk = L + t_new
        if sk == 0 or Tq == 0:
            continue
        idx = [slot(b, lp + j) for j in range(sk)]
        i0 = np.array([a for a, _ in idx])
        i1 = np.array([c for _, c in idx])
        kk_all = np.asarray(k_cache[i0, i1], dtype=np.float64) * kd   # [sk, Hk, D]
        vv_all = np.asarray(v_cache[i0, i1], dtype=np.float64) * vd
        for h in range(Hq):
            g = h // group
            qq = q[b, :, h].copy()
            if rotary_cos is not None:
                for i in range(Tq):
                    pos = L + lp + (i if local else 0)
                    qq[i] = apply_rope(qq[i], rotary_cos, rotary_sin, pos, rotary_interleaved,
                                       io_dtype)
            s, vis = score_matrix(qq, kk_all[:, g], scale, causal, wl, wr, softcap,
                                  _slope(alibi_slopes, b, h))
            m = np.max(s, axis=1, keepdims=True)
            m_safe = np.where(np.isfinite(m), m, 0.0)
            e = np.where(vis, np.exp(s - m_safe), 0.0)
            l = e.sum(axis=1, keepdims=True)
            p = np.where(l > 0, e / np.where(l > 0, l, 1.0), 0.0)
            out[b, :, h] = p @ vv_all[:, g]
            has = l[:, 0] > 0
            lse[b, h] = np.where(has, m_safe[:, 0] + np.log(np.where(has, l[:, 0], 1.0)), -np.inf)
    return out, lse.astype(np.float32)
