"""GPU parity: dense forward through the Python API -> C ABI -> HIP kernels vs the oracle
and the reference-generated golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from util import (DT, assert_close, assert_lse_close, errs, f64, lowp_attention_bhsd, rand16)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _fa():
    import flash_attn
    return flash_attn


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "dense_*.npz"))))
def test_golden_dense_fwd(path):
    """Reference protocol (test.py:273-277): max|o - o_ref| <= 2 max|o_fp16torch - o_ref| + 1e-5."""
    g = np.load(path)
    q, k, v = (torch.from_numpy(g[n]).cuda() for n in ("q", "k", "v"))      # [B,H,S,D] fp16
    causal, scale = bool(g["causal"]), float(g["scale"])
    out, lse, _ = _fa().flash_attn_func(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                        softmax_scale=scale, causal=causal, return_attn_probs=True)
    out = out.transpose(1, 2)
    o_ref = torch.from_numpy(g["o"]).cuda()
    o_pt = lowp_attention_bhsd(q, k, v, scale, causal)
    err = (out.float() - o_ref).abs().max().item()
    err_pt = (o_pt.float() - o_ref).abs().max().item()
    assert torch.isfinite(out).all()
    assert err <= 2 * err_pt + 1e-5, (err, err_pt)
    _, lse_ref, _ = oracle.attn_fwd(g["q"].astype(np.float64), g["k"].astype(np.float64),
                                    g["v"].astype(np.float64), scale, causal=causal, normalize=False)
    assert_lse_close(f64(lse), lse_ref, "lse")


CASES = [
    # B, Hq, Hk, Sq, Sk, D, dtype, causal, window, softcap, alibi
    (2, 4, 4, 128, 128, 128, "bf16", False, (-1, -1), 0.0, False),
    (2, 4, 4, 128, 128, 128, "bf16", True, (-1, -1), 0.0, False),
    (1, 4, 2, 200, 200, 128, "fp16", True, (-1, -1), 0.0, False),
    (1, 4, 1, 257, 300, 64, "bf16", True, (-1, -1), 0.0, False),      # Sq < Sk bottom-right
    (1, 2, 2, 300, 130, 64, "fp16", True, (-1, -1), 0.0, False),      # Sq > Sk: empty rows
    (1, 2, 2, 1, 77, 128, "fp16", False, (-1, -1), 0.0, False),       # single query
    (2, 2, 2, 33, 65, 128, "bf16", False, (-1, -1), 0.0, False),
    (1, 4, 4, 384, 384, 128, "fp16", False, (100, 0), 0.0, False),    # sliding window (causal band)
    (1, 4, 4, 384, 384, 64, "bf16", False, (64, 32), 0.0, False),     # two-sided window
    (1, 4, 4, 256, 256, 128, "fp16", True, (-1, -1), 0.0, True),      # ALiBi
    (2, 4, 2, 256, 256, 128, "bf16", False, (-1, -1), 30.0, False),   # softcap
    (1, 4, 4, 192, 192, 64, "fp16", True, (-1, -1), 15.0, True),      # ALiBi then softcap
    (1, 2, 2, 512, 512, 128, "bf16", True, (-1, -1), 0.0, False),     # multi q-block
    # ALiBi: the causal / window_right == 0 cases take the rank-2 MFMA fast path, the others the general one
    (2, 8, 2, 700, 700, 128, "bf16", True, (-1, -1), 0.0, True),      # fast path, GQA, several tiles
    (1, 4, 4, 300, 520, 64, "fp16", True, (-1, -1), 0.0, True),       # fast path, Sq < Sk (off > 0)
    (1, 4, 4, 520, 300, 128, "bf16", True, (-1, -1), 0.0, True),      # fast path, Sq > Sk (empty rows)
    (1, 4, 4, 640, 640, 128, "bf16", False, (200, 0), 0.0, True),     # fast path via window_right == 0
    (1, 4, 4, 320, 320, 128, "bf16", False, (-1, -1), 0.0, True),     # general path (keys right of the diagonal)
    (1, 4, 4, 320, 320, 64, "fp16", False, (64, 32), 0.0, True),      # general path, two-sided window
    # softcap only (constants-folded variant; Gemma-2 style)
    (2, 8, 2, 333, 333, 128, "bf16", True, (-1, -1), 50.0, False),
    (1, 4, 4, 200, 450, 64, "fp16", True, (128, 0), 20.0, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(map(str, c)))
def test_dense_fwd_vs_oracle(case):
    B, Hq, Hk, Sq, Sk, D, dt, causal, window, softcap, alibi = case
    q = rand16((B, Sq, Hq, D), dt, 421)
    k = rand16((B, Sk, Hk, D), dt, 422)
    v = rand16((B, Sk, Hk, D), dt, 423)
    slopes = None
    if alibi:
        slopes = (2.0 ** (-8.0 * (torch.arange(Hq) + 1) / Hq)).float().cuda()
    out, lse, _ = _fa().flash_attn_func(q, k, v, causal=causal, window_size=window, softcap=softcap,
                                        alibi_slopes=slopes, return_attn_probs=True)
    assert out.shape == q.shape and out.dtype == q.dtype
    o_ref, lse_ref, _ = oracle.attn_fwd(
        f64(q).transpose(0, 2, 1, 3), f64(k).transpose(0, 2, 1, 3), f64(v).transpose(0, 2, 1, 3),
        D ** -0.5, causal=causal, window=window, softcap=softcap,
        alibi_slopes=None if slopes is None else f64(slopes))
    assert_close(f64(out).transpose(0, 2, 1, 3), o_ref, dt, "out")
    assert_lse_close(f64(lse), lse_ref, "lse")


def test_strided_inputs_no_copy_path():
    """q/k/v as slices of a packed qkv tensor (strided heads), output matches the oracle."""
    B, S, H, D = 2, 160, 4, 128
    qkv = rand16((B, S, 3, H, D), "bf16", 7)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    out = _fa().flash_attn_func(q, k, v, causal=True)
    o_ref, _, _ = oracle.attn_fwd(*(f64(t).transpose(0, 2, 1, 3) for t in (q, k, v)), D ** -0.5,
                                  causal=True)
    assert_close(f64(out).transpose(0, 2, 1, 3), o_ref, "bf16", "out")


def test_head_dim_padding_and_errors():
    q = rand16((1, 40, 2, 40), "fp16", 3)                       # D=40 -> padded to 64
    out = _fa().flash_attn_func(q, q, q, causal=True)
    o_ref, _, _ = oracle.attn_fwd(*(f64(t).transpose(0, 2, 1, 3) for t in (q, q, q)), 40 ** -0.5,
                                  causal=True)
    assert_close(f64(out).transpose(0, 2, 1, 3), o_ref, "fp16", "out")
    with pytest.raises(RuntimeError):                            # H_Q % H_K != 0
        _fa().flash_attn_func(rand16((1, 8, 3, 64), "fp16", 1), rand16((1, 8, 2, 64), "fp16", 2),
                              rand16((1, 8, 2, 64), "fp16", 2))
    with pytest.raises(RuntimeError):                            # CPU tensors: no fallback
        _fa().flash_attn_func(q.cpu(), q.cpu(), q.cpu())


def test_empty_keys():
    q = rand16((1, 5, 2, 64), "fp16", 3)
    k = rand16((1, 0, 2, 64), "fp16", 4)
    out, lse, _ = _fa().flash_attn_func(q, k, k, return_attn_probs=True)
    assert (out == 0).all() and torch.isneginf(lse).all()


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_full_size_property_config2(dt):
    """BASELINE config 2 shape (B8 H16 S4096 D128 causal) via a size-independent property:
    the kernel on the full tensor equals the kernel run per-(batch, head-pair) and a few
    sampled rows equal the oracle."""
    B, S, H, D = 8, 4096, 16, 128
    q = rand16((B, S, H, D), dt, 421); k = rand16((B, S, H, D), dt, 422); v = rand16((B, S, H, D), dt, 423)
    out, lse, _ = _fa().flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
    assert torch.isfinite(out).all() and torch.isfinite(lse).all()
    # consistency: slicing batch/heads commutes with attention
    sub = _fa().flash_attn_func(q[3:4, :, 5:7], k[3:4, :, 5:7], v[3:4, :, 5:7], causal=True)
    assert torch.equal(sub, out[3:4, :, 5:7])
    # sampled rows vs fp64 oracle: row i of causal attention only needs keys <= i
    for (b, h, i) in [(0, 0, 0), (7, 15, 4095), (3, 5, 1234), (2, 9, 63), (5, 1, 64), (1, 2, 2047)]:
        qq = f64(q[b, i:i + 1, h])[None, None]
        kk = f64(k[b, :i + 1, h])[None, None]
        vv = f64(v[b, :i + 1, h])[None, None]
        o_ref, lse_ref, _ = oracle.attn_fwd(qq, kk, vv, D ** -0.5)
        assert_close(f64(out[b, i, h]), o_ref[0, 0, 0], dt, f"row {b},{h},{i}", mult=2.0)
        assert abs(float(lse[b, h, i]) - float(lse_ref[0, 0, 0])) < 2e-3


def test_full_size_property_config5_shard():
    """BASELINE config 5, one GPU's shard at full size (B64, heads 4..7 of 32, S8192, D128, bf16,
    causal + ALiBi with the sharded slopes): slicing commutes with attention and sampled rows
    equal the oracle (row i of causal attention only needs keys <= i; bottom-right alignment makes
    the one-row problem carry the same ALiBi distances)."""
    from flash_attn_mi355.sharding import shard_alibi
    B, S, Htot, D, dt = 64, 8192, 32, 128, "bf16"
    slopes_all = (2.0 ** (-8.0 * (torch.arange(Htot) + 1) / Htot)).float()
    h0, h1 = 4, 8                                                  # rank 1 of 8
    slopes = shard_alibi(slopes_all, slice(h0, h1), slice(0, B)).cuda()
    H = h1 - h0
    q = rand16((B, S, H, D), dt, 421); k = rand16((B, S, H, D), dt, 422); v = rand16((B, S, H, D), dt, 423)
    out, lse, _ = _fa().flash_attn_func(q, k, v, causal=True, alibi_slopes=slopes, return_attn_probs=True)
    assert torch.isfinite(out).all() and torch.isfinite(lse).all()
    sub = _fa().flash_attn_func(q[9:10, :, 1:3], k[9:10, :, 1:3], v[9:10, :, 1:3], causal=True,
                                alibi_slopes=slopes[1:3].contiguous())
    assert torch.equal(sub, out[9:10, :, 1:3])
    for (b, h, i) in [(0, 0, 0), (63, 3, 8191), (17, 2, 4097), (5, 1, 63), (40, 0, 64), (33, 3, 6000)]:
        qq = f64(q[b, i:i + 1, h])[None, None]
        kk = f64(k[b, :i + 1, h])[None, None]
        vv = f64(v[b, :i + 1, h])[None, None]
        o_ref, lse_ref, _ = oracle.attn_fwd(qq, kk, vv, D ** -0.5, causal=True,
                                            alibi_slopes=f64(slopes[h:h + 1]))
        assert_close(f64(out[b, i, h]), o_ref[0, 0, 0], dt, f"row {b},{h},{i}", mult=2.0)
        assert abs(float(lse[b, h, i]) - float(lse_ref[0, 0, 0])) < 3e-3


def test_degenerate_shapes():
    """Empty query block (no-op, zero dK/dV), stride-0 expanded K/V heads == MQA, 1 x 1, and a 64K-token row."""
    fa = _fa()
    q = torch.randn(2, 0, 4, 64, device="cuda", dtype=torch.float16, requires_grad=True)
    k = torch.randn(2, 50, 4, 64, device="cuda", dtype=torch.float16, requires_grad=True)
    o = fa.flash_attn_func(q, k, k, causal=True)
    assert o.shape == (2, 0, 4, 64)
    dq, dk = torch.autograd.grad(o, (q, k), torch.zeros_like(o))
    assert dq.shape == q.shape and (dk == 0).all()
    q = torch.randn(1, 100, 8, 128, device="cuda", dtype=torch.bfloat16)
    k1 = torch.randn(1, 100, 1, 128, device="cuda", dtype=torch.bfloat16)
    a = fa.flash_attn_func(q, k1.expand(-1, -1, 8, -1), k1.expand(-1, -1, 8, -1), causal=True)
    assert torch.equal(a, fa.flash_attn_func(q, k1, k1, causal=True))
    o = fa.flash_attn_func(q[:, :1], k1[:, :1], k1[:, :1])
    assert torch.allclose(o.float(), k1[:, :1].expand(-1, -1, 8, -1).float(), atol=1e-2)
    q = torch.randn(1, 65536, 1, 64, device="cuda", dtype=torch.float16)
    o, lse, _ = fa.flash_attn_func(q, q, q, causal=True, return_attn_probs=True)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    # row 0 of causal attention is v[0]
    assert torch.allclose(o[0, 0].float(), q[0, 0].float(), atol=1e-3)


# Forward key split of one-wave causal launches (opt-in: include/fa_mi355.h FA_FLAG_FWD_KEY_SPLIT, fa_fwd_asm.hip): the heavy 256-row
# blocks of a launch whose blocks all get a CU at once leave fp32 partials per key range, a merge kernel combines them.
FS_CASES = [
    # B, Hq, Hk, Sq, Sk, dtype
    (1, 32, 32, 2048, 2048, "bf16"),
    (1, 32, 8, 2048, 2048, "fp16"),          # GQA
    (1, 32, 32, 1900, 1900, "bf16"),         # ragged last block
    (1, 16, 16, 2048, 3000, "bf16"),         # Sq < Sk: every block sees the 952 extra keys
    (2, 8, 8, 4096, 4096, "fp16"),
]


@pytest.mark.parametrize("case", FS_CASES, ids=lambda c: "-".join(map(str, c)))
def test_forward_key_split_vs_oracle(case, monkeypatch):
    from flash_attn_mi355 import flash_attn_interface as fi
    B, Hq, Hk, Sq, Sk, dt = case
    D = 128
    q, k, v = rand16((B, Sq, Hq, D), dt, 31), rand16((B, Sk, Hk, D), dt, 32), rand16((B, Sk, Hk, D), dt, 33)
    real = fi._workspace
    res = {}
    for on in (False, True):
        monkeypatch.setattr(fi, "FWD_SPLIT", on)
        seen = []
        monkeypatch.setattr(fi, "_workspace", lambda n, dev: (seen.append(n), real(n, dev))[1])
        out, lse, _ = _fa().flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
        again, lse2, _ = _fa().flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
        assert torch.equal(out, again) and torch.equal(lse, lse2)           # deterministic either way
        res[on] = (out, lse, max(seen) if seen else 0)
    assert res[True][2] > 0 and res[False][2] == 0                           # the split path ran (it asked for its partial buffers)
    assert_close(f64(res[True][0]), f64(res[False][0]), dt, "split vs unsplit")      # (one 16-bit ulp where the fp32 sums round differently)
    assert_lse_close(f64(res[True][1]), f64(res[False][1]), "lse split vs unsplit")
    hs = slice(0, Hq, max(1, Hq // 4))                                       # oracle on every fourth head
    t = lambda x: f64(x).transpose(0, 2, 1, 3)
    hk = slice(0, Hk, max(1, Hk // 4)) if Hq == Hk else None
    if hk is not None:
        o_ref, lse_ref, _ = oracle.attn_fwd(t(q)[:, hs], t(k)[:, hk], t(v)[:, hk], D ** -0.5, causal=True)
        assert_close(t(res[True][0])[:, hs], o_ref, dt, "out (split)")
        assert_lse_close(f64(res[True][1])[:, hs], lse_ref, "lse (split)")
    else:
        o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=True)
        assert_close(t(res[True][0]), o_ref, dt, "out (split)")
        assert_lse_close(f64(res[True][1]), lse_ref, "lse (split)")
