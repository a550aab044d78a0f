"""GPU parity: dense backward (autograd through the Python API -> fa_bwd) vs the oracle and
the reference-generated golden gradients."""
import glob
import warnings
import os

import numpy as np
import pytest
import torch

import oracle
from util import assert_close, f64, rand16

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _fa():
    import flash_attn
    return flash_attn


def _lowp_grads(q, k, v, do, scale, causal):
    q, k, v = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bhmd,bhnd->bhmn", q, k) * scale
    if causal:
        m = torch.triu(torch.ones(s.shape[-2], s.shape[-1], device=s.device, dtype=torch.bool), 1)
        s = s.masked_fill(m, float("-inf"))
    o = torch.einsum("bhmn,bhnd->bhmd", torch.softmax(s, -1), v)
    return torch.autograd.grad(o, (q, k, v), do)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "dense_*.npz"))))
def test_golden_dense_bwd(path):
    """Reference protocol (test.py:322-334): err <= 3 err_fp16torch + 1e-4 per gradient."""
    g = np.load(path)
    q, k, v, do = (torch.from_numpy(g[n]).cuda() for n in ("q", "k", "v", "do"))   # [B,H,S,D]
    causal, scale = bool(g["causal"]), float(g["scale"])
    qq, kk, vv = (t.transpose(1, 2).contiguous().requires_grad_(True) for t in (q, k, v))
    out = _fa().flash_attn_func(qq, kk, vv, softmax_scale=scale, causal=causal)
    dq, dk, dv = torch.autograd.grad(out, (qq, kk, vv), do.transpose(1, 2))
    pt = _lowp_grads(q, k, v, do, scale, causal)
    for name, got, lp in (("dq", dq, pt[0]), ("dk", dk, pt[1]), ("dv", dv, pt[2])):
        ref = torch.from_numpy(g[name]).cuda()
        got = got.transpose(1, 2).float()
        assert torch.isfinite(got).all()
        err = (got - ref).abs().max().item()
        err_pt = (lp.float() - ref).abs().max().item()
        assert err <= 3 * err_pt + 1e-4, (name, err, err_pt)


CASES = [
    # B, Hq, Hk, Sq, Sk, D, dtype, causal, window, softcap, alibi
    (2, 4, 4, 128, 128, 128, "bf16", False, (-1, -1), 0.0, False),
    (2, 4, 4, 128, 128, 128, "fp16", True, (-1, -1), 0.0, False),
    (1, 4, 2, 200, 200, 128, "bf16", True, (-1, -1), 0.0, False),     # GQA, ragged
    (1, 4, 1, 257, 300, 64, "fp16", True, (-1, -1), 0.0, False),      # MQA, Sq < Sk
    (1, 2, 2, 300, 130, 64, "bf16", True, (-1, -1), 0.0, False),      # Sq > Sk: empty rows
    (2, 2, 2, 33, 65, 128, "fp16", False, (-1, -1), 0.0, False),
    (1, 4, 4, 384, 384, 128, "bf16", False, (100, 0), 0.0, False),    # sliding window
    (1, 4, 4, 384, 384, 64, "fp16", False, (64, 32), 0.0, False),
    (1, 4, 4, 256, 256, 128, "fp16", True, (-1, -1), 0.0, True),      # ALiBi
    (2, 4, 2, 256, 256, 128, "bf16", False, (-1, -1), 30.0, False),   # softcap
    (1, 4, 4, 192, 192, 64, "fp16", True, (-1, -1), 15.0, True),
    (1, 2, 2, 512, 512, 128, "bf16", True, (-1, -1), 0.0, False),
    # ALiBi: causal / window_right == 0 take the matrix-pipe path in both backward kernels, the rest the general one
    (2, 8, 2, 700, 700, 128, "bf16", True, (-1, -1), 0.0, True),      # GQA (slope changes inside a dK/dV workgroup)
    (1, 4, 4, 300, 520, 64, "fp16", True, (-1, -1), 0.0, True),       # Sq < Sk (off > 0)
    (1, 4, 4, 520, 300, 128, "bf16", True, (-1, -1), 0.0, True),      # Sq > Sk (empty rows)
    (1, 4, 4, 640, 640, 128, "bf16", False, (200, 0), 0.0, True),     # window_right == 0
    (1, 4, 4, 320, 320, 128, "bf16", False, (-1, -1), 0.0, True),     # general path (keys right of the diagonal)
    (1, 2, 2, 1500, 1500, 128, "fp16", True, (-1, -1), 0.0, True),    # long distances: tile term split three ways
    # ... and in the hand-scheduled dQ kernel the bias is part of the exponent's arithmetic (a lane owns one query)
    (1, 4, 4, 300, 520, 128, "fp16", True, (-1, -1), 0.0, True),      # Sq < Sk (off > 0), rows past the last 256-row block
    (1, 4, 2, 1024, 1024, 128, "bf16", False, (300, 0), 0.0, True),   # GQA (per q-head slope) + left window
    (2, 2, 2, 2304, 2304, 128, "bf16", True, (-1, -1), 0.0, True),    # nine 256-row blocks: mirrored pairs + the middle one
    # hand-scheduled dK/dV kernel (D = 128, no bias): edges of its stage pipeline, masks and key-block pairing
    (1, 2, 2, 520, 300, 128, "bf16", True, (-1, -1), 0.0, False),     # Sq > Sk: rows without keys, key blocks without rows
    (1, 2, 2, 300, 520, 128, "fp16", True, (-1, -1), 0.0, False),     # Sq < Sk: shifted diagonal
    (2, 2, 1, 1, 700, 128, "bf16", False, (-1, -1), 0.0, False),      # one query row: a single (mostly empty) stage
    (1, 2, 2, 700, 2, 128, "fp16", False, (-1, -1), 0.0, False),      # two keys (with one key dQ = 0 exactly: P (dP - D) cancels)
    (1, 2, 2, 31, 1000, 128, "bf16", True, (-1, -1), 0.0, False),     # less than one stage of rows, eight key blocks
    (1, 4, 1, 1000, 1000, 128, "fp16", False, (64, 32), 0.0, False),  # two-sided window, four q-heads per kv-head
    (1, 2, 2, 1300, 1300, 128, "bf16", True, (-1, -1), 0.0, False),   # 11 key blocks: mirrored pairs + the middle one
    # head dim 256 without bias / dropout: two waves per key block (one computes P and dV, the other dP, dS and dK)
    (1, 2, 2, 300, 520, 256, "bf16", True, (-1, -1), 0.0, False),     # Sq < Sk, ragged tails
    (1, 2, 2, 520, 300, 256, "fp16", True, (-1, -1), 0.0, False),     # Sq > Sk: rows without keys, key blocks without rows
    (2, 4, 1, 700, 700, 256, "bf16", False, (-1, -1), 0.0, False),    # MQA group of four, no mask
    (1, 2, 2, 1000, 1000, 256, "fp16", False, (64, 32), 0.0, False),  # two-sided window
    (1, 2, 2, 1300, 1300, 256, "bf16", True, (-1, -1), 0.0, False),   # 11 key blocks: mirrored pairs + the middle one
    (1, 2, 2, 31, 1000, 256, "bf16", True, (-1, -1), 0.0, False),     # less than one stage of rows
    # ... head dims 129 .. 192 on the same kernel without the k-steps / accumulator blocks of the zero columns
    (1, 2, 2, 300, 520, 192, "bf16", True, (-1, -1), 0.0, False),
    (2, 4, 2, 513, 513, 160, "fp16", True, (-1, -1), 0.0, False),
    (1, 2, 1, 700, 900, 136, "bf16", False, (200, 50), 0.0, False),
    (1, 2, 2, 600, 600, 200, "fp16", True, (-1, -1), 0.0, False),     # 193 .. 256: the full width
    # ... and with softcap only (Gemma-2's head dim 256 form): the P wave hands P (1 - tanh^2) over
    (2, 4, 2, 700, 700, 256, "bf16", True, (-1, -1), 50.0, False),
    (1, 2, 2, 520, 300, 256, "fp16", False, (300, 0), 30.0, False),   # sliding window, Sq > Sk
    (1, 2, 1, 300, 520, 160, "bf16", True, (-1, -1), 20.0, False),
    # softcap only (constants-folded variant in all kernels; Gemma-2 style)
    (2, 8, 2, 333, 333, 128, "bf16", True, (-1, -1), 50.0, False),
    (1, 4, 4, 200, 450, 64, "fp16", True, (128, 0), 20.0, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(map(str, c)))
def test_dense_bwd_vs_oracle(case):
    B, Hq, Hk, Sq, Sk, D, dt, causal, window, softcap, alibi = case
    q = rand16((B, Sq, Hq, D), dt, 421).requires_grad_(True)
    k = rand16((B, Sk, Hk, D), dt, 422).requires_grad_(True)
    v = rand16((B, Sk, Hk, D), dt, 423).requires_grad_(True)
    do = rand16((B, Sq, Hq, D), dt, 424)
    slopes = None
    if alibi:
        slopes = (2.0 ** (-8.0 * (torch.arange(Hq) + 1) / Hq)).float().cuda()
    out = _fa().flash_attn_func(q, k, v, causal=causal, window_size=window, softcap=softcap,
                                alibi_slopes=slopes)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
    t = lambda x: f64(x).transpose(0, 2, 1, 3)
    kw = dict(causal=causal, window=window, softcap=softcap,
              alibi_slopes=None if slopes is None else f64(slopes))
    o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, **kw)
    dq_r, dk_r, dv_r, _ = oracle.attn_bwd(t(do), t(q), t(k), t(v), o_ref, lse_ref.astype(np.float64),
                                          D ** -0.5, **kw)
    assert_close(t(dq), dq_r, dt, "dq", mult=2.0)
    assert_close(t(dk), dk_r, dt, "dk", mult=2.0)
    assert_close(t(dv), dv_r, dt, "dv", mult=2.0)


def test_bwd_deterministic_and_linear():
    """Size-independent properties at a large shape: bitwise repeatability (atomic-free) and
    linearity of the gradients in dO."""
    B, S, H, D = 2, 2048, 8, 128
    q = rand16((B, S, H, D), "bf16", 1).requires_grad_(True)
    k = rand16((B, S, H, D), "bf16", 2).requires_grad_(True)
    v = rand16((B, S, H, D), "bf16", 3).requires_grad_(True)
    do = rand16((B, S, H, D), "bf16", 4)
    out = _fa().flash_attn_func(q, k, v, causal=True)
    g1 = torch.autograd.grad(out, (q, k, v), do, retain_graph=True)
    g2 = torch.autograd.grad(out, (q, k, v), do, retain_graph=True)
    for a_, b_ in zip(g1, g2):
        assert torch.equal(a_, b_)
    g3 = torch.autograd.grad(out, (q, k, v), do * 2)
    for a_, b_ in zip(g1, g3):
        assert torch.isfinite(a_).all()
        rel = ((b_.float() - 2 * a_.float()).abs().max() / (2 * a_.float()).abs().max()).item()
        assert rel < 2e-2, rel


@pytest.mark.parametrize("D,dt", [(256, "bf16"), (192, "fp16"), (32, "fp16"), (16, "bf16")])
def test_head_dims_of_the_reference(D, dt):
    """Reference head dims {16, 32, 64, 128, 256} (fused_mha_forward.cu:421-428) + a padded one."""
    B, S, Hq, Hk = 1, 160, 4, 2
    q = rand16((B, S, Hq, D), dt, 1).requires_grad_(True)
    k = rand16((B, S, Hk, D), dt, 2).requires_grad_(True)
    v = rand16((B, S, Hk, D), dt, 3).requires_grad_(True)
    do = rand16((B, S, Hq, D), dt, 4)
    out = _fa().flash_attn_func(q, k, v, causal=True)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    t = lambda x: f64(x).transpose(0, 2, 1, 3)
    o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=True)
    g = oracle.attn_bwd(t(do), t(q), t(k), t(v), o_ref, lse_ref.astype(np.float64), D ** -0.5, causal=True)
    assert_close(t(out), o_ref, dt, "out")
    assert_close(t(dq), g[0], dt, "dq", mult=2.0)
    assert_close(t(dk), g[1], dt, "dk", mult=2.0)
    assert_close(t(dv), g[2], dt, "dv", mult=2.0)


@pytest.mark.parametrize("D,dt", [(96, "bf16"), (80, "fp16"), (40, "bf16"), (192, "fp16"), (36, "fp16")])
def test_odd_head_dims_without_padded_copies(D, dt):
    """Head dims between the kernel widths run through head_dim_v (columns past D read as zero, never
    written): results equal the oracle at the true D, and the tensors handed to the C ABI are the
    caller's own (no padded copies) when D is a multiple of 8."""
    B, S, H, Hk = 2, 200, 4, 2
    q = rand16((B, S, H, D), dt, 421).requires_grad_(True)
    k = rand16((B, S, Hk, D), dt, 422).requires_grad_(True)
    v = rand16((B, S, Hk, D), dt, 423).requires_grad_(True)
    do = rand16((B, S, H, D), dt, 424)
    out, lse, _ = _fa().flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
    assert out.shape == q.shape
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    tr = lambda t: f64(t).transpose(0, 2, 1, 3)
    o_ref, lse_ref, _ = oracle.attn_fwd(tr(q), tr(k), tr(v), D ** -0.5, causal=True)
    g_ref = oracle.attn_bwd(tr(do), tr(q), tr(k), tr(v), tr(out), f64(lse), D ** -0.5, causal=True)
    assert_close(tr(out), o_ref, dt, "out")
    for name, got, ref in (("dq", dq, g_ref[0]), ("dk", dk, g_ref[1]), ("dv", dv, g_ref[2])):
        assert got.shape[-1] == D
        assert_close(tr(got), ref, dt, name, mult=1.5)


@pytest.mark.parametrize("S,D,H,Hk,causal,dt", [(1024, 128, 4, 2, True, "bf16"), (777, 128, 2, 2, False, "fp16"),
                                                (500, 64, 4, 4, True, "fp16"), (384, 256, 2, 1, True, "bf16")])
def test_unneeded_gradients_are_skipped_not_changed(S, D, H, Hk, causal, dt):
    """autograd's needs_input_grad reaches the C ABI as dq == NULL / dk == dv == NULL: the kernels that do run give the
    same bits as in the full backward (dQ alone; dK/dV alone then takes D = rowsum(dO o O) from the preprocess kernel
    instead of the dQ kernel's prologue - the same fp32 sum order is not promised there, so that pair is compared to
    tolerance), and a K/V-only or Q-only caller gets no gradient for the rest."""
    import flash_attn
    q = rand16((2, S, H, D), dt, 1).requires_grad_(True)
    k = rand16((2, S, Hk, D), dt, 2).requires_grad_(True)
    v = rand16((2, S, Hk, D), dt, 3).requires_grad_(True)
    do = rand16((2, S, H, D), dt, 4)
    o = flash_attn.flash_attn_func(q, k, v, causal=causal)
    dq, dk, dv = torch.autograd.grad(o, (q, k, v), do)
    # frozen K / V (only q requires grad): needs_input_grad = (True, False, False) -> dk = dv = NULL
    q1 = q.detach().clone().requires_grad_(True)
    flash_attn.flash_attn_func(q1, k.detach(), v.detach(), causal=causal).backward(do)
    assert torch.equal(q1.grad, dq)
    # frozen Q: dq = NULL, the preprocess kernel provides D
    k1, v1 = k.detach().clone().requires_grad_(True), v.detach().clone().requires_grad_(True)
    flash_attn.flash_attn_func(q.detach(), k1, v1, causal=causal).backward(do)
    assert_close(f64(k1.grad), f64(dk), dt, "dk (dk, dv only)", mult=0.5)
    assert_close(f64(v1.grad), f64(dv), dt, "dv (dk, dv only)", mult=0.5)
    # only v requires grad: dk is computed with it (one kernel) and dropped
    v2 = v.detach().clone().requires_grad_(True)
    flash_attn.flash_attn_func(q.detach(), k.detach(), v2, causal=causal).backward(do)
    assert torch.equal(v2.grad, v1.grad)


def test_large_lds_kernels_on_every_visible_device():
    """The > 64 KiB dynamic-LDS attribute is per device (fa_common.h: FA_SET_LDS_ONCE keeps a per-device flag): a D = 128
    forward + backward must launch on every GPU of one process.  Needs >= 2 visible devices."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device")
    import flash_attn
    ref = None
    for dev in range(torch.cuda.device_count()):
        q, k, v, do = (rand16((1, 512, 2, 128), "bf16", 10 + i, device=f"cuda:{dev}") for i in range(4))
        q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
        o = flash_attn.flash_attn_func(q, k, v, causal=True)
        g = torch.autograd.grad(o, (q, k, v), do)
        torch.cuda.synchronize(dev)
        got = [t.cpu() for t in (o.detach(),) + g]
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), f"device {dev} differs from device 0"


def test_backward_without_workspace_matches(monkeypatch):
    """workspace = NULL is legal: the generated dQ kernel then has no statistics planes to leave behind
    (empty descriptor) and the compiler-scheduled dK/dV kernel runs - same gradients as with the workspace."""
    import flash_attn
    from flash_attn_mi355 import flash_attn_interface as fi
    q = rand16((2, 1024, 4, 128), "bf16", 1).requires_grad_(True)
    k = rand16((2, 1024, 2, 128), "bf16", 2).requires_grad_(True)
    v = rand16((2, 1024, 2, 128), "bf16", 3).requires_grad_(True)
    do = rand16((2, 1024, 4, 128), "bf16", 4)
    o = flash_attn.flash_attn_func(q, k, v, causal=True)
    ref = torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
    monkeypatch.setattr(fi, "_workspace", lambda nbytes, device: None)
    got = torch.autograd.grad(o, (q, k, v), do)
    assert torch.equal(got[0], ref[0])                                   # same dQ kernel, same bits
    assert_close(f64(got[1]), f64(ref[1]), "bf16", "dk (no workspace)", mult=0.5)
    assert_close(f64(got[2]), f64(ref[2]), "bf16", "dv (no workspace)", mult=0.5)


@pytest.mark.parametrize("dt,scale", [("fp16", 2048.0), ("fp16", 1.0 / 64.0), ("bf16", 1048576.0)])
def test_backward_with_loss_scaled_gradients(dt, scale):
    """dO far from O(1) (mixed-precision loss scaling, or small gradients; powers of two, and small enough that dS stays a
    normal fp16 number): D = rowsum(dO o O) leaves the fp16 range at scale 2048 (|D| up to ~1e5) - the generated dQ kernel carries it as three 16-bit terms through the matrix pipe, the first one
    scaled by 2^-12 for fp16 - and the gradients must simply scale with dO."""
    import flash_attn
    q = rand16((1, 1024, 2, 128), dt, 1).requires_grad_(True)
    k = rand16((1, 1024, 2, 128), dt, 2).requires_grad_(True)
    v = rand16((1, 1024, 2, 128), dt, 3, scale=2.0).requires_grad_(True)
    do1 = rand16((1, 1024, 2, 128), dt, 4)
    o = flash_attn.flash_attn_func(q, k, v, causal=True)
    ref = torch.autograd.grad(o, (q, k, v), do1, retain_graph=True)
    got = torch.autograd.grad(o, (q, k, v), (do1.float() * scale).to(do1.dtype))          # (a power of two: exact)
    for name, g, r in zip(("dq", "dk", "dv"), got, ref):
        assert torch.isfinite(g.float()).all(), name
        assert_close(f64(g) / scale, f64(r), dt, f"{name} at dO x {scale}", mult=1.0)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_peaky_softmax_forward_and_backward(dt):
    """q, k three times the usual scale: logits of +-40, one or two keys carry a row - late rescales of the running maximum
    in the forward, P = exp2(S c - lse2) recomputed near 1 and near 0 in the backward kernels; against the oracle."""
    import flash_attn
    B, S, H, D = 1, 1024, 2, 128
    q = rand16((B, S, H, D), dt, 1, scale=3.0).requires_grad_(True)
    k = rand16((B, S, H, D), dt, 2, scale=3.0).requires_grad_(True)
    v = rand16((B, S, H, D), dt, 3).requires_grad_(True)
    do = rand16((B, S, H, D), dt, 4)
    o, lse, _ = flash_attn.flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
    g = torch.autograd.grad(o, (q, k, v), do)
    t = lambda x: f64(x).transpose(0, 2, 1, 3)
    o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=True)
    g_ref = oracle.attn_bwd(t(do), t(q), t(k), t(v), oracle.round_to(o_ref, dt), lse_ref.astype(np.float64), D ** -0.5, causal=True)
    assert_close(t(o), o_ref, dt, "out", mult=1.5)
    assert np.abs(f64(lse) - lse_ref).max() <= 1e-4
    for name, a, b in zip(("dq", "dk", "dv"), g, g_ref):
        assert_close(t(a), b, dt, name, mult=2.0)


# dK/dV launches smaller than the chip split their query rows over several workgroups and add fp32 partials
# (fa_bwd.hip: dkv_split_factor / dkv_reduce_kernel; include/fa_mi355.h: FA_FLAG_NO_DKV_SPLIT).  Every small case above runs
# that form; here: the shapes it was built for, both forms against the oracle and against each other.
SPLIT_CASES = [
    # B, Hq, Hk, Sq, Sk, D, dtype, causal, window
    (1, 8, 2, 1024, 1024, 128, "bf16", True, (-1, -1)),     # GQA at micro-batch 1: hand-scheduled kernel, group of 4, mirrored pairs
    (1, 4, 4, 1100, 1100, 128, "fp16", True, (-1, -1)),     # one q-head per kv-head: the fast copies start inside a split
    (1, 16, 1, 600, 600, 128, "bf16", False, (-1, -1)),     # MQA group of 16, no mask
    (2, 4, 4, 2048, 77, 64, "fp16", False, (-1, -1)),       # short-key cross-attention: one ragged key block
    (2, 4, 4, 1500, 77, 40, "fp16", False, (-1, -1)),       # ... head dim 40 (valid columns only)
    (1, 4, 2, 1024, 300, 96, "bf16", False, (-1, -1)),      # 128-wide compiler kernel, 96 valid columns
    (1, 14, 2, 900, 900, 64, "bf16", True, (-1, -1)),       # group of 7 at head dim 64
    (1, 4, 2, 1300, 1300, 128, "bf16", False, (200, 0)),    # left window: passes of different lengths
    (1, 2, 2, 1100, 1500, 64, "fp16", True, (-1, -1)),      # Sq < Sk: key blocks without rows
    (1, 4, 2, 1024, 1024, 256, "bf16", True, (-1, -1)),     # head dim 256: two waves per key block, each role stores its partial
    (1, 2, 2, 900, 300, 192, "fp16", False, (-1, -1)),      # ... 192 valid columns
]


def _split_bytes(B, Hq, Hk, Sq, Sk, D, dt, causal, window, flags):
    from flash_attn_mi355 import _lib
    import ctypes
    p = _lib.FaParams()
    p.batch, p.nheads_q, p.nheads_k, p.seqlen_q, p.seqlen_k = B, Hq, Hk, Sq, Sk
    p.head_dim = 64 if D <= 64 else (128 if D <= 128 else 256)
    p.head_dim_v = D if D != p.head_dim else 0
    p.dtype = p.kv_dtype = _lib.FA_BF16 if dt == "bf16" else _lib.FA_FP16
    p.is_causal, p.window_left, p.window_right = int(causal), window[0], window[1]
    p.softmax_scale = D ** -0.5
    p.q_row_stride = p.do_row_stride = Hq * D; p.q_head_stride = p.do_head_stride = D
    p.k_row_stride = p.v_row_stride = p.dk_row_stride = p.dv_row_stride = Hk * D
    p.flags = flags
    return int(_lib.lib.fa_bwd_workspace_bytes(ctypes.byref(p)))


@pytest.mark.parametrize("case", SPLIT_CASES, ids=lambda c: "-".join(map(str, c)))
def test_small_dkdv_launches_split_their_query_rows(case, monkeypatch):
    from flash_attn_mi355 import _lib, flash_attn_interface as fi
    B, Hq, Hk, Sq, Sk, D, dt, causal, window = case
    monkeypatch.setattr(fi, "DS_HANDOFF", False)             # (an FA_BWD_DS=1 environment would hand the UNSPLIT call's dS over: another dQ kernel)
    # the split is on for this shape: it asks for partial slabs on top of the unsplit call's workspace
    assert _split_bytes(*case, 0) > _split_bytes(*case, _lib.FA_FLAG_NO_DKV_SPLIT)
    q = rand16((B, Sq, Hq, D), dt, 521).requires_grad_(True)
    k = rand16((B, Sk, Hk, D), dt, 522).requires_grad_(True)
    v = rand16((B, Sk, Hk, D), dt, 523).requires_grad_(True)
    do = rand16((B, Sq, Hq, D), dt, 524)
    grads = {}
    for on in (True, False):
        run = lambda: _fa().flash_attn_func(q, k, v, causal=causal, window_size=window, deterministic=not on)   # the public switch
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            grads[on] = torch.autograd.grad(run(), (q, k, v), do)
            again = torch.autograd.grad(run(), (q, k, v), do)
        for a_, b_ in zip(grads[on], again):
            assert torch.equal(a_, b_)                       # both forms are deterministic
    assert torch.equal(grads[True][0], grads[False][0])      # dQ does not know about it
    t = lambda x: f64(x).transpose(0, 2, 1, 3)
    o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=causal, window=window)
    g = oracle.attn_bwd(t(do), t(q), t(k), t(v), o_ref, lse_ref.astype(np.float64), D ** -0.5, causal=causal, window=window)
    for on in (True, False):
        assert_close(t(grads[on][1]), g[1], dt, f"dk split={on}", mult=2.0)
        assert_close(t(grads[on][2]), g[2], dt, f"dv split={on}", mult=2.0)


def test_deterministic_means_batch_invariant_gradients():
    """`deterministic=True` sets FA_FLAG_NO_DKV_SPLIT: a sample's dK / dV have the same BITS alone and inside a larger batch
    (the split launches sum fp32 partials whose number follows batch x kv-heads and the CU count)."""
    B, S, Hq, Hk, D = 4, 1024, 8, 2, 128
    q = rand16((B, S, Hq, D), "bf16", 901).requires_grad_(True)
    k = rand16((B, S, Hk, D), "bf16", 902).requires_grad_(True)
    v = rand16((B, S, Hk, D), "bf16", 903).requires_grad_(True)
    do = rand16((B, S, Hq, D), "bf16", 904)

    def grads(n):
        qq, kk, vv = (t[:n].detach().clone().requires_grad_(True) for t in (q, k, v))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            out = _fa().flash_attn_func(qq, kk, vv, causal=True, deterministic=True)
        return torch.autograd.grad(out, (qq, kk, vv), do[:n])
    one, four = grads(1), grads(4)
    for a_, b_ in zip(one, four):
        assert torch.equal(a_[0], b_[0])


def test_a_full_round_of_dkdv_workgroups_is_not_split():
    from flash_attn_mi355 import _lib
    big = (8, 16, 16, 4096, 4096, 128, "bf16", True, (-1, -1))          # BASELINE config 2: 2048 workgroups
    assert _split_bytes(*big, 0) == _split_bytes(*big, _lib.FA_FLAG_NO_DKV_SPLIT)
    drop = (1, 4, 2, 1024, 1024, 128, "bf16", True, (-1, -1))
    assert _split_bytes(*drop, 0) > _split_bytes(*drop, _lib.FA_FLAG_NO_DKV_SPLIT)


@pytest.mark.parametrize("B,Hq,Hk", [(1, 4, 1), (1, 6, 3), (1, 5, 5), (3, 6, 3), (1, 11, 11), (2, 12, 6), (1, 26, 13)])
@pytest.mark.parametrize("D,dt,causal", [(128, "bf16", True), (64, "fp16", False)])
def test_units_that_do_not_fill_a_round_of_xcds(B, Hq, Hk, D, dt, causal):
    """batch x kv-heads = 1, 3, 5, 9, 11, 12, 13: the units past the last full round of eight are laid end to end and cut into
    eight runs (fa_common.h: decode_unit_item) in the forward, dQ and dK/dV grids - every (unit, item) exactly once."""
    S = 600
    q = rand16((B, S, Hq, D), dt, 31).requires_grad_(True)
    k = rand16((B, S, Hk, D), dt, 32).requires_grad_(True)
    v = rand16((B, S, Hk, D), dt, 33).requires_grad_(True)
    do = rand16((B, S, Hq, D), dt, 34)
    out = _fa().flash_attn_func(q, k, v, causal=causal)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    t = lambda x: f64(x).transpose(0, 2, 1, 3)
    o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=causal)
    g = oracle.attn_bwd(t(do), t(q), t(k), t(v), o_ref, lse_ref.astype(np.float64), D ** -0.5, causal=causal)
    assert_close(t(out), o_ref, dt, "out")
    assert_close(t(dq), g[0], dt, "dq", mult=2.0)
    assert_close(t(dk), g[1], dt, "dk", mult=2.0)
    assert_close(t(dv), g[2], dt, "dv", mult=2.0)


# dS hand-off (opt-in: include/fa_mi355.h FA_FLAG_DS_HANDOFF, fa_bwd_dq_ds.hip): the generated dK/dV kernel stores its packed dS
# tiles, a one-GEMM dQ kernel reads them back.  Launches that fill the chip only (smaller ones split their dK/dV pass instead).
DS_CASES = [
    # B, Hq, Hk, Sq, Sk, dtype, causal, window
    (4, 16, 16, 1024, 1024, "bf16", True, (-1, -1)),
    (2, 16, 16, 2048, 2048, "fp16", False, (-1, -1)),
    (8, 8, 8, 1000, 1000, "bf16", True, (-1, -1)),           # ragged: partial last row tile and key block
    (8, 32, 8, 1536, 1536, "bf16", True, (-1, -1)),          # GQA: the tile offsets wrap from head to head (64 units x 6 pairs fill the chip)
    (8, 8, 8, 700, 1300, "fp16", True, (-1, -1)),            # Sq < Sk (bottom-right aligned)
    (8, 8, 8, 1300, 700, "bf16", True, (-1, -1)),            # Sq > Sk: rows without keys
    (4, 16, 16, 1024, 1024, "bf16", False, (200, 0)),        # sliding window: key blocks start late
    (4, 16, 16, 1024, 1024, "fp16", False, (100, 50)),       # two-sided window
]


@pytest.mark.parametrize("case", DS_CASES, ids=lambda c: "-".join(map(str, c)))
def test_ds_handoff_backward_vs_oracle(case, monkeypatch):
    if os.environ.get("FA_BWD_ASM") == "0" or os.environ.get("FA_BWD_DQ_ASM") == "0":
        pytest.skip("the hand-off lives in the generated dK/dV kernel: an A/B run on the compiler-scheduled kernels has nothing to test here")
    from flash_attn_mi355 import flash_attn_interface as fi
    B, Hq, Hk, Sq, Sk, dt, causal, window = case
    D = 128
    q = rand16((B, Sq, Hq, D), dt, 811).requires_grad_(True)
    k = rand16((B, Sk, Hk, D), dt, 812).requires_grad_(True)
    v = rand16((B, Sk, Hk, D), dt, 813).requires_grad_(True)
    do = rand16((B, Sq, Hq, D), dt, 814)
    real = fi._workspace
    grads, ws = {}, {}
    for on in (False, True):
        monkeypatch.setattr(fi, "DS_HANDOFF", on)
        seen = []
        monkeypatch.setattr(fi, "_workspace", lambda n, dev: (seen.append(n), real(n, dev))[1])
        out = _fa().flash_attn_func(q, k, v, causal=causal, window_size=window)
        grads[on] = torch.autograd.grad(out, (q, k, v), do)
        ws[on] = max(seen)
        again = torch.autograd.grad(_fa().flash_attn_func(q, k, v, causal=causal, window_size=window), (q, k, v), do)
        for a_, b_ in zip(grads[on], again):
            assert torch.equal(a_, b_)                       # atomic-free either way
    assert ws[True] >= ws[False] + B * Hq * ((Sq + 31) // 32) * 4 * ((Sk + 127) // 128) * 2048      # the hand-off path ran
    sel = slice(0, 2)                                        # oracle on two batch entries (fp64 on the CPU)
    t = lambda x: f64(x[sel]).transpose(0, 2, 1, 3)
    o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=causal, window=window)
    g = oracle.attn_bwd(t(do), t(q), t(k), t(v), o_ref, lse_ref.astype(np.float64), D ** -0.5, causal=causal, window=window)
    for on in (False, True):
        for i, name in enumerate(("dq", "dk", "dv")):
            assert_close(t(grads[on][i]), g[i], dt, f"{name} handoff={on}", mult=2.0)
    # the two paths against each other: same products, another summation order for dQ (and D from another kernel)
    for i, name in enumerate(("dq", "dk", "dv")):
        assert_close(f64(grads[True][i]), f64(grads[False][i]), dt, f"{name} handoff vs recompute", mult=0.5)
