"""GPU parity: packed variable-length forward/backward (and paged K/V) vs the oracle."""
import warnings
import numpy as np
import pytest
import torch

import oracle
from util import LSE_ATOL_FP8, assert_close, assert_lse_close, f64, rand16

pytestmark = pytest.mark.gpu


def _fa():
    import flash_attn
    return flash_attn


def _cu(lens):
    return torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")


VCASES = [
    # lens_q, lens_k, Hq, Hk, D, dtype, causal, window, softcap, alibi
    ([5, 128, 77, 1], None, 4, 4, 64, "fp16", True, (-1, -1), 0.0, False),
    ([200, 3, 129, 64], None, 4, 2, 128, "bf16", False, (-1, -1), 0.0, False),
    ([130, 257], [300, 257], 2, 1, 64, "fp16", True, (-1, -1), 0.0, False),      # Sq != Sk
    ([100, 260, 31], None, 4, 4, 64, "fp16", False, (64, 0), 0.0, False),         # sliding window
    ([100, 260, 31], None, 2, 2, 128, "bf16", True, (-1, -1), 20.0, True),        # alibi + softcap
    ([0, 40, 0, 9], None, 2, 2, 64, "fp16", True, (-1, -1), 0.0, False),          # empty sequences
]


@pytest.mark.parametrize("case", VCASES, ids=lambda c: "-".join(map(str, c)))
def test_varlen_fwd_bwd_vs_oracle(case):
    lens_q, lens_k, Hq, Hk, D, dt, causal, window, softcap, alibi = case
    lens_k = lens_k or lens_q
    B = len(lens_q)
    Tq, Tk = sum(lens_q), sum(lens_k)
    q = rand16((Tq, Hq, D), dt, 1).requires_grad_(True)
    k = rand16((Tk, Hk, D), dt, 2).requires_grad_(True)
    v = rand16((Tk, Hk, D), dt, 3).requires_grad_(True)
    do = rand16((Tq, Hq, D), dt, 4)
    cu_q, cu_k = _cu(lens_q), _cu(lens_k)
    slopes = torch.tensor([0.25 * (i + 1) / Hq for i in range(Hq)], dtype=torch.float32, device="cuda") \
        if alibi else None
    mq, mk = max(lens_q), max(lens_k)
    out, lse, _ = _fa().flash_attn_varlen_func(q, k, v, cu_q, cu_k, mq, mk, causal=causal,
                                               window_size=window, softcap=softcap, alibi_slopes=slopes,
                                               return_attn_probs=True)
    assert out.shape == q.shape and lse.shape == (Hq, Tq)
    kw = dict(causal=causal, window=window, softcap=softcap,
              alibi_slopes=None if slopes is None else f64(slopes))
    cq, ck = cu_q.cpu().numpy(), cu_k.cpu().numpy()
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), f64(k), f64(v), cq, ck, mq, mk, D ** -0.5, **kw)
    assert_close(f64(out), o_ref, dt, "out")
    assert_lse_close(f64(lse), lse_ref, "lse")
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    dq_r, dk_r, dv_r, _ = oracle.varlen_bwd(f64(do), f64(q), f64(k), f64(v), o_ref,
                                            lse_ref.astype(np.float64), cq, ck, mq, mk, D ** -0.5, **kw)
    assert_close(f64(dq), dq_r, dt, "dq", mult=2.0)
    assert_close(f64(dk), dk_r, dt, "dk", mult=2.0)
    assert_close(f64(dv), dv_r, dt, "dv", mult=2.0)


@pytest.mark.parametrize("page,D", [(64, 128), (256, 128), (128, 96), (64, 64), (16, 128), (32, 64), (48, 128), (192, 128),
                                     (320, 64)])
def test_varlen_paged_kv(page, D):
    lens_q, lens_k = [70, 1, 300], [200, 513, 300]
    Hq, Hk, dt = 4, 2, "fp16"
    B = len(lens_q)
    nblk_per_seq = [(l + page - 1) // page for l in lens_k]
    max_blocks = max(nblk_per_seq)
    total_blocks = sum(nblk_per_seq) + 3
    perm = torch.randperm(total_blocks, generator=torch.Generator().manual_seed(5)).tolist()
    block_table = torch.zeros((B, max_blocks), dtype=torch.int32)
    it = iter(perm)
    for b in range(B):
        for j in range(nblk_per_seq[b]):
            block_table[b, j] = next(it)
    kp = rand16((total_blocks, page, Hk, D), dt, 11)
    vp = rand16((total_blocks, page, Hk, D), dt, 12)
    q = rand16((sum(lens_q), Hq, D), dt, 13)
    cu_q, cu_k = _cu(lens_q), _cu(lens_k)
    out, lse, _ = _fa().flash_attn_varlen_func(q, kp, vp, cu_q, cu_k, max(lens_q), max(lens_k), causal=True,
                                               return_attn_probs=True, block_table=block_table.cuda())
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), f64(kp), f64(vp), cu_q.cpu().numpy(), cu_k.cpu().numpy(),
                                       max(lens_q), max(lens_k), D ** -0.5, causal=True,
                                       block_table=block_table.numpy())
    assert_close(f64(out), o_ref, dt, "out")
    assert_lse_close(f64(lse), lse_ref, "lse")


@pytest.mark.parametrize("page,causal,use_seqused", [(64, True, False), (256, True, True), (128, False, False)])
def test_varlen_paged_prefill_on_the_hand_scheduled_forward(page, causal, use_seqused):
    """Packed prefill over a paged 16-bit cache at D = 128 with an average of >= 256 rows per sequence: the flat work list of
    256-row blocks on the hand-scheduled forward's PAGED bodies (FA_ASM_FORCE=1), page tails poisoned with NaN; oracle."""
    import os
    lens_q, lens_k = [300, 700, 257], [300, 1100, 900]
    Hq, Hk, D, dt = 4, 2, 128, "bf16"
    B = len(lens_q)
    used = [l - 13 if use_seqused and i == 1 else l for i, l in enumerate(lens_k)]
    nblk_per_seq = [(l + page - 1) // page for l in lens_k]
    total_blocks = sum(nblk_per_seq) + 3
    perm = torch.randperm(total_blocks, generator=torch.Generator().manual_seed(5)).tolist()
    block_table = torch.zeros((B, max(nblk_per_seq)), dtype=torch.int32)
    it = iter(perm)
    kp = rand16((total_blocks, page, Hk, D), dt, 11); vp = rand16((total_blocks, page, Hk, D), dt, 12)
    for b in range(B):
        for j in range(nblk_per_seq[b]):
            block_table[b, j] = next(it)
            lo = max(0, used[b] - j * page)
            if lo < page:
                kp[int(block_table[b, j]), lo:] = float("nan"); vp[int(block_table[b, j]), lo:] = float("nan")
    q = rand16((sum(lens_q), Hq, D), dt, 13)
    cu_q, cu_k = _cu(lens_q), _cu(lens_k)
    su = torch.tensor(used, dtype=torch.int32).cuda() if use_seqused else None
    os.environ["FA_ASM_FORCE"] = "1"
    try:
        out, lse, _ = _fa().flash_attn_varlen_func(q, kp, vp, cu_q, cu_k, max(lens_q), max(lens_k), causal=causal,
                                                   return_attn_probs=True, block_table=block_table.cuda(), seqused_k=su)
    finally:
        del os.environ["FA_ASM_FORCE"]
    assert not torch.isnan(out).any()
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), np.nan_to_num(f64(kp)), np.nan_to_num(f64(vp)), cu_q.cpu().numpy(), cu_k.cpu().numpy(),
                                       max(lens_q), max(lens_k), D ** -0.5, causal=causal, block_table=block_table.numpy(),
                                       seqused_k=None if su is None else su.cpu().numpy())
    assert_close(f64(out), o_ref, dt, "out", mult=1.5)
    assert_lse_close(f64(lse), lse_ref, "lse")


@pytest.mark.parametrize("Tq,Hq,Hk,D,dt,page,use_seqused,causal,window,softcap,alibi", [
    (1, 32, 8, 128, "bf16", 256, True, True, (-1, -1), 0.0, False),      # the vLLM-style decode step: token-major kernel
    (1, 8, 8, 128, "fp16", 64, False, True, (-1, -1), 0.0, False),       # lengths from cu_seqlens_k only
    (4, 16, 2, 128, "fp16", 128, True, True, (-1, -1), 0.0, False),      # 4 speculative tokens, G = 8: one row block
    (3, 8, 2, 64, "bf16", 64, True, True, (200, 0), 0.0, True),          # window + ALiBi, D = 64, seqused_k clamps cu_seqlens_k
    (2, 6, 2, 96, "fp16", 128, False, False, (-1, -1), 30.0, False),     # narrow head dim + softcap (varlen allows it without a window)
    (1, 4, 4, 256, "bf16", 256, True, True, (-1, -1), 0.0, False),       # D = 256
])
def test_varlen_decode_runs_the_decode_kernels(Tq, Hq, Hk, D, dt, page, use_seqused, causal, window, softcap, alibi):
    """Decode issued through the varlen op: every sequence brings the same T_q query tokens (total_q == B * max_seqlen_q), K / V
    are paged (block_table), lengths come from seqused_k and / or cu_seqlens_k.  With the workspace fa_fwd_workspace_bytes()
    asks for, fa_varlen_fwd hands the call to the decode kernels (GQA packing, split-KV: 240 -> 32 us at batch 1,
    tools/varlen_decode_probe.py); semantics are the varlen op's (include/template.h:65-68 for seqused_k, bottom-right causal
    alignment, LSE [H, T]) - checked against the varlen oracle, and against the general kernel (no workspace: the C ABI's
    fallback) to the io tolerance."""
    import ctypes
    from flash_attn_mi355 import _lib, flash_attn_interface as fi
    lens_k = [700, 33, 1024, 1, 257]
    B = len(lens_k)
    used = [600, 33, 1000, 1, 300] if use_seqused else None      # (last entry: seqused_k > the cu_seqlens_k difference)
    pps = [(l + page - 1) // page for l in lens_k]
    nblk = sum(pps) + 2
    g = torch.Generator().manual_seed(41)
    perm = iter(torch.randperm(nblk, generator=g).tolist())
    bt = torch.zeros((B, max(pps)), dtype=torch.int32)
    for b in range(B):
        for j in range(pps[b]):
            bt[b, j] = next(perm)
    kp = rand16((nblk, page, Hk, D), dt, 11); vp = rand16((nblk, page, Hk, D), dt, 12)
    q = rand16((B * Tq, Hq, D), dt, 13)
    cu_q, cu_k = _cu([Tq] * B), _cu(lens_k)
    su = None if used is None else torch.tensor(used, dtype=torch.int32).cuda()
    slopes = torch.tensor([0.03 * (i + 1) for i in range(Hq)], dtype=torch.float32, device="cuda") if alibi else None
    kw = dict(causal=causal, window_size=window, softcap=softcap, alibi_slopes=slopes, block_table=bt.cuda(), seqused_k=su)
    calls = []
    orig = _lib.call
    def spy(name, p, stream):
        calls.append((name, int(p.workspace_bytes)))
        return orig(name, p, stream)
    _lib.call = spy
    try:
        out, lse, _ = _fa().flash_attn_varlen_func(q, kp, vp, cu_q, cu_k, Tq, max(lens_k), return_attn_probs=True, **kw)
    finally:
        _lib.call = orig
    assert calls and calls[0][0] == "fa_varlen_fwd"
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), f64(kp), f64(vp), cu_q.cpu().numpy(), cu_k.cpu().numpy(), Tq, max(lens_k), D ** -0.5,
                                       causal=causal, window=window, softcap=softcap,
                                       alibi_slopes=None if slopes is None else f64(slopes),
                                       seqused_k=None if used is None else np.array(used), block_table=bt.numpy())
    assert_close(f64(out), o_ref, dt, "out")
    assert_lse_close(f64(lse), lse_ref, "lse")
    # the general kernel on the same call (no workspace offered): same rows to the io tolerance
    orig_ws = fi._workspace
    fi._workspace = lambda nbytes, device: None
    try:
        out2, lse2, _ = _fa().flash_attn_varlen_func(q, kp, vp, cu_q, cu_k, Tq, max(lens_k), return_attn_probs=True, **kw)
    finally:
        fi._workspace = orig_ws
    assert_close(f64(out2), o_ref, dt, "out (general kernel)")
    assert_lse_close(f64(lse2), lse_ref, "lse (general kernel)")


@pytest.mark.parametrize("Tq,qlens,Hq,Hk,dt,fp8", [
    (2, [2, 2, 1, 1], 8, 2, "bf16", False),          # the advisor's case: T 8 rows of q, B 4, max 2, cu = [0, 2, 4, 5, 6]
    (1, [1, 1, 0, 1, 1], 8, 8, "fp16", False),       # token-major kernel, a sequence without a row, one padding row
    (1, [1, 0, 1, 1], 4, 4, "bf16", True),           # fp8 head-major matrix-vector kernel
    (1, [1, 1, 1, 0], 32, 16, "bf16", True),         # fp8 token-major, G = 2
    (4, [4, 1, 3, 4], 16, 2, "fp16", False),         # MFMA decode kernel, G = 8
    (2, [2, 2, 1, 0, 2], 32, 8, "bf16", True),       # fp8-operand MFMA form, a kv-head per wave (8 kv-heads, G = 4), varlen-q rows
])
def test_varlen_decode_with_padding_rows_in_q(Tq, qlens, Hq, Hk, dt, fp8):
    """total_q == batch x max_seqlen_q does not prove that sequence b sits at row b T: q may carry padding rows behind
    cu_seqlens_q[-1] (graph-captured serving steps pad the token dimension) - the reference takes every row offset from
    cu_seqlens_q on the device (include/template.h:55-69).  The decode route keeps cu_seqlens_q (varlen-q mode); rows past
    cu_seqlens_q[-1] are never written (NaN canary)."""
    D, page = 128, 64
    lens_k = [700, 33, 300, 129, 64][:len(qlens)]
    B = len(qlens)
    total_q = B * Tq
    assert sum(qlens) < total_q and max(qlens) == Tq
    pps = [(l + page - 1) // page for l in lens_k]
    nblk = sum(pps) + 1
    g = torch.Generator().manual_seed(43)
    perm = iter(torch.randperm(nblk, generator=g).tolist())
    bt = torch.zeros((B, max(pps)), dtype=torch.int32)
    for b in range(B):
        for j in range(pps[b]):
            bt[b, j] = next(perm)
    kp = rand16((nblk, page, Hk, D), dt, 11); vp = rand16((nblk, page, Hk, D), dt, 12)
    kw = {}
    kp_ref, vp_ref = f64(kp), f64(vp)
    if fp8:
        kd, vd = 0.05, 0.04
        kp8 = (kp.float() / kd).to(torch.float8_e4m3fn); vp8 = (vp.float() / vd).to(torch.float8_e4m3fn)
        kp_ref, vp_ref = kp8.float().double().cpu().numpy() * kd, vp8.float().double().cpu().numpy() * vd
        kp, vp = kp8, vp8
        kw = dict(k_descale=kd, v_descale=vd)
    q = rand16((total_q, Hq, D), dt, 13)
    cu_q, cu_k = _cu(qlens), _cu(lens_k)
    out = torch.full((total_q, Hq, D), float("nan"), dtype=q.dtype, device="cuda")
    from flash_attn_mi355 import flash_attn_interface as fi
    o, lse = fi._varlen_forward(q, kp, vp, cu_q, cu_k, Tq, max(lens_k), 0.0, None, True, (-1, -1), 0.0, None, True,
                                bt.cuda(), out=out, **kw)[:2]
    assert o.data_ptr() == out.data_ptr()
    n = sum(qlens)
    o_ref, lse_ref = oracle.varlen_fwd(f64(q)[:n], kp_ref, vp_ref, cu_q.cpu().numpy(), cu_k.cpu().numpy(), Tq, max(lens_k),
                                       D ** -0.5, causal=True, block_table=bt.numpy())
    assert_close(f64(o)[:n], o_ref, dt, "out", mult=3.0 if fp8 else 1.0)
    assert_lse_close(f64(lse)[:, :n], lse_ref, "lse", **(dict(atol=LSE_ATOL_FP8) if fp8 else {}))
    assert torch.isnan(o[n:]).all(), "padding rows of q were written"


@pytest.mark.parametrize("qlens,Hq,Hk,D,dt,page,causal,window,softcap,alibi", [
    ([1] * 12 + [300], 32, 8, 128, "bf16", 256, True, (-1, -1), 0.0, False),              # 12 decode sequences + one prefill chunk
    ([1, 1, 700, 1, 2, 1, 4, 1, 1, 3], 16, 4, 128, "fp16", 128, True, (-1, -1), 0.0, False),   # chunk in the middle, 1-4 speculative tokens (T = 8)
    ([1, 5, 1, 1, 1, 1, 200, 1], 16, 2, 64, "bf16", 64, True, (-1, -1), 0.0, True),       # G = 8 -> T = 4: the 5-row sequence stays with the general kernel; ALiBi
    ([1] * 6 + [150, 1, 1], 8, 8, 128, "fp16", 256, False, (-1, -1), 30.0, False),        # MHA, non-causal + softcap
    ([2, 1, 1, 1, 90, 1, 1], 8, 2, 96, "bf16", 128, True, (300, 0), 0.0, False),          # narrow head dim + sliding window
])
def test_varlen_mixed_batch_splits_between_the_kernels(qlens, Hq, Hk, D, dt, page, causal, window, softcap, alibi):
    """A mixed serving batch through ONE varlen call: sequences with 1 .. T query rows (T = min(8, 32 / G)) are served by the
    decode kernels in varlen-q mode, the others by fa_fwd_kernel with those left out (fa_api.hip varlen_mixed_route; 792 ->
    ~350 us for 32 decode sequences + a 512-token chunk, tools/mixed_batch_probe.py).  Every row of the packed output must be
    written exactly once: out / lse start as NaN.  Checked against the varlen oracle, and against the general kernel alone
    (no workspace offered)."""
    from flash_attn_mi355 import flash_attn_interface as fi
    B = len(qlens)
    g = torch.Generator().manual_seed(43)
    lens_k = [int(x) for x in torch.randint(40, 900, (B,), generator=g)]
    lens_k = [max(lk, ql) for lk, ql in zip(lens_k, qlens)]
    used = [lk - (i % 3) for i, lk in enumerate(lens_k)]
    pps = [(l + page - 1) // page for l in lens_k]
    nblk = sum(pps) + 2
    perm = iter(torch.randperm(nblk, generator=g).tolist())
    bt = torch.zeros((B, max(pps)), dtype=torch.int32)
    for b in range(B):
        for j in range(pps[b]):
            bt[b, j] = next(perm)
    kp = rand16((nblk, page, Hk, D), dt, 11); vp = rand16((nblk, page, Hk, D), dt, 12)
    q = rand16((sum(qlens), Hq, D), dt, 13)
    cu_q, cu_k = _cu(qlens), _cu(lens_k)
    su = torch.tensor(used, dtype=torch.int32).cuda()
    slopes = torch.tensor([0.02 * (i + 1) for i in range(Hq)], dtype=torch.float32, device="cuda") if alibi else None
    kw = dict(causal=causal, window_size=window, softcap=softcap, alibi_slopes=slopes, block_table=bt.cuda(), seqused_k=su)
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), f64(kp), f64(vp), cu_q.cpu().numpy(), cu_k.cpu().numpy(), max(qlens), max(lens_k), D ** -0.5,
                                       causal=causal, window=window, softcap=softcap,
                                       alibi_slopes=None if slopes is None else f64(slopes),
                                       seqused_k=np.array(used), block_table=bt.numpy())
    # poison the allocator's next blocks so that an unwritten output row shows
    junk = [torch.full((sum(qlens), Hq, D), float("nan"), dtype=q.dtype, device="cuda"), torch.full((Hq, sum(qlens)), float("nan"), device="cuda")]
    del junk
    out, lse, _ = _fa().flash_attn_varlen_func(q, kp, vp, cu_q, cu_k, max(qlens), max(lens_k), return_attn_probs=True, **kw)
    assert torch.isfinite(out.float()).all(), "an output row was not written"
    assert_close(f64(out), o_ref, dt, "out")
    assert_lse_close(f64(lse), lse_ref, "lse")
    orig_ws = fi._workspace
    fi._workspace = lambda nbytes, device: None
    try:
        out2, lse2, _ = _fa().flash_attn_varlen_func(q, kp, vp, cu_q, cu_k, max(qlens), max(lens_k), return_attn_probs=True, **kw)
    finally:
        fi._workspace = orig_ws
    assert_close(f64(out2), o_ref, dt, "out (general kernel)")
    assert_lse_close(f64(lse2), lse_ref, "lse (general kernel)")


@pytest.mark.parametrize("qlens,Hq,Hk,D,page,causal", [
    ([70, 1, 300, 129], 8, 2, 128, 128, True),              # ragged prefill: fa_fwd_kernel<..., KV8> on the flat work list
    ([1] * 6, 32, 8, 128, 256, True),                       # uniform decode through the varlen op: decode kernels
    ([1] * 9 + [260, 1, 2], 16, 4, 128, 256, True),         # mixed batch: both
    ([40, 200, 1], 4, 4, 64, 64, False),                    # D = 64, non-causal
    ([70, 1, 300, 129], 8, 2, 128, 16, True),               # 16-token pages: one entry per (wave, chunk step)
    ([150, 33], 4, 4, 64, 32, True),                        # D = 64, 32-token pages
])
def test_varlen_paged_fp8_kv(qlens, Hq, Hk, D, page, causal):
    """Paged fp8-e4m3 K / V through the varlen op (this build's extension, as in flash_attn_with_kvcache: value = code x
    descale, forward only): the general kernel dequantises a tile once per 128 query rows, the decode routes serve uniform
    and mixed batches.  The oracle sees the dequantised cache; tolerance: 1.5 x the io dtype's, LSE 3e-2 (the fp8 cases')."""
    dt = "bf16"
    B = len(qlens)
    g = torch.Generator().manual_seed(47)
    lens_k = [max(int(x), ql) for x, ql in zip(torch.randint(30, 700, (B,), generator=g), qlens)]
    pps = [(l + page - 1) // page for l in lens_k]
    nblk = sum(pps) + 1
    perm = iter(torch.randperm(nblk, generator=g).tolist())
    bt = torch.zeros((B, max(pps)), dtype=torch.int32)
    for b in range(B):
        for j in range(pps[b]):
            bt[b, j] = next(perm)
    kd, vd = 0.0625, 0.03125
    kp = (rand16((nblk, page, Hk, D), dt, 11, scale=1.5).float() / kd).to(torch.float8_e4m3fn)
    vp = (rand16((nblk, page, Hk, D), dt, 12, scale=1.5).float() / vd).to(torch.float8_e4m3fn)
    q = rand16((sum(qlens), Hq, D), dt, 13)
    cu_q, cu_k = _cu(qlens), _cu(lens_k)
    su = torch.tensor(lens_k, dtype=torch.int32).cuda()
    out, lse, _ = _fa().flash_attn_varlen_func(q, kp, vp, cu_q, cu_k, max(qlens), max(lens_k), causal=causal, return_attn_probs=True,
                                               block_table=bt.cuda(), seqused_k=su, k_descale=kd, v_descale=vd)
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), kp.float().double().cpu().numpy() * kd, vp.float().double().cpu().numpy() * vd,
                                       cu_q.cpu().numpy(), cu_k.cpu().numpy(), max(qlens), max(lens_k), D ** -0.5, causal=causal,
                                       seqused_k=np.array(lens_k), block_table=bt.numpy())
    assert_close(f64(out), o_ref, dt, "out", mult=1.5)
    assert_lse_close(f64(lse), lse_ref, "lse", atol=LSE_ATOL_FP8)
    # lengths only (cu_seqlens_k = None with seqused_k, as vLLM-style wrappers call it): the same rows
    out_n = _fa().flash_attn_varlen_func(q, kp, vp, cu_q, None, max(qlens), max(lens_k), causal=causal,
                                         block_table=bt.cuda(), seqused_k=su, k_descale=kd, v_descale=vd)
    assert torch.equal(out_n, out)
    with pytest.raises(RuntimeError):                       # forward only
        qg = q.clone().requires_grad_()
        _fa().flash_attn_varlen_func(qg, kp, vp, cu_q, cu_k, max(qlens), max(lens_k), causal=causal, block_table=bt.cuda())
    with pytest.raises(RuntimeError):                       # packed (unpaged) fp8 k / v are not taken
        _fa().flash_attn_varlen_func(q, kp.reshape(-1, Hk, D)[: sum(lens_k)], vp.reshape(-1, Hk, D)[: sum(lens_k)], cu_q, cu_k,
                                     max(qlens), max(lens_k), causal=causal)


def test_config3_shape_properties():
    """BASELINE config 3: fp16 packed batch 64, seqlens in [64, 2048] (max forced to 2048), H32 D64,
    window (512, 0).  Full size via properties: window (512,0) == causal + window_left 512, and a few
    sequences equal the oracle."""
    g = torch.Generator().manual_seed(421)
    lens = torch.randint(64, 2049, (64,), generator=g).tolist()
    lens[7] = 2048
    H, D = 32, 64
    T = sum(lens)
    q = rand16((T, H, D), "fp16", 1); k = rand16((T, H, D), "fp16", 2); v = rand16((T, H, D), "fp16", 3)
    cu = _cu(lens)
    o1 = _fa().flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, window_size=(512, 0))
    o2 = _fa().flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(512, -1))
    assert torch.isfinite(o1).all() and torch.equal(o1, o2)
    cun = cu.cpu().numpy()
    for b in (0, 7, 63):
        s0, s1 = int(cun[b]), int(cun[b + 1])
        hs = slice(3, 5)
        o_ref, _, _ = oracle.attn_fwd(f64(q[s0:s1, hs]).transpose(1, 0, 2)[None],
                                      f64(k[s0:s1, hs]).transpose(1, 0, 2)[None],
                                      f64(v[s0:s1, hs]).transpose(1, 0, 2)[None], D ** -0.5, window=(512, 0))
        assert_close(f64(o1[s0:s1, hs]).transpose(1, 0, 2)[None], o_ref, "fp16", f"seq {b}")


def test_config3_backward_full_size_vs_oracle():
    """BASELINE config 3 at FULL size, backward: three sequences (first, the 2048-row one, last) x two heads of
    dQ / dK / dV against the fp64 oracle (the oracle gets its own forward output rounded to the io type)."""
    g = torch.Generator().manual_seed(421)
    lens = torch.randint(64, 2049, (64,), generator=g).tolist()
    lens[7] = 2048
    H, D, dt = 32, 64, "fp16"
    T = sum(lens)
    q = rand16((T, H, D), dt, 1).requires_grad_(True)
    k = rand16((T, H, D), dt, 2).requires_grad_(True)
    v = rand16((T, H, D), dt, 3).requires_grad_(True)
    do = rand16((T, H, D), dt, 4)
    cu = _cu(lens)
    out = _fa().flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, window_size=(512, 0))
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    for t in (dq, dk, dv):
        assert torch.isfinite(t.float()).all()
    cun = cu.cpu().numpy()
    tr = lambda x, s0, s1, hs: f64(x[s0:s1, hs]).transpose(1, 0, 2)[None]
    for b in (0, 7, 63):
        s0, s1 = int(cun[b]), int(cun[b + 1])
        hs = slice(3, 5)
        o_ref, lse_ref, _ = oracle.attn_fwd(tr(q, s0, s1, hs), tr(k, s0, s1, hs), tr(v, s0, s1, hs), D ** -0.5,
                                            window=(512, 0))
        gr = oracle.attn_bwd(tr(do, s0, s1, hs), tr(q, s0, s1, hs), tr(k, s0, s1, hs), tr(v, s0, s1, hs),
                             oracle.round_to(o_ref, dt), lse_ref.astype(np.float64), D ** -0.5, window=(512, 0))
        assert_close(tr(out, s0, s1, hs), o_ref, dt, f"out seq {b}")
        assert_close(tr(dq, s0, s1, hs), gr[0], dt, f"dq seq {b}", mult=2.0)
        assert_close(tr(dk, s0, s1, hs), gr[1], dt, f"dk seq {b}", mult=2.0)
        assert_close(tr(dv, s0, s1, hs), gr[2], dt, f"dv seq {b}", mult=2.0)


def test_varlen_with_empty_sequences():
    """cu_seqlens with zero-length entries (query-side, key-side and both): no crash, rows of the other
    sequences unaffected, empty-key rows give O = 0 / LSE = -inf, gradients finite."""
    import flash_attn
    torch.manual_seed(421)
    H, D = 4, 64
    lens_q = [50, 0, 33, 7]
    lens_k = [50, 20, 0, 130]
    cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens_q), H, D, device="cuda", dtype=torch.float16, requires_grad=True)
    k = torch.randn(sum(lens_k), H, D, device="cuda", dtype=torch.float16, requires_grad=True)
    v = torch.randn(sum(lens_k), H, D, device="cuda", dtype=torch.float16, requires_grad=True)
    out, lse, _ = flash_attn.flash_attn_varlen_func(q, k, v, cu_q, cu_k, max(lens_q), max(lens_k), causal=False,
                                                    return_attn_probs=True)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), torch.randn_like(out))
    for t in (out, dq, dk, dv):
        assert torch.isfinite(t.float()).all()
    # sequence 2 has no keys: its 33 query rows
    r0 = lens_q[0] + lens_q[1]
    assert (out[r0:r0 + 33] == 0).all() and torch.isneginf(lse[:, r0:r0 + 33]).all()
    assert (dq[r0:r0 + 33] == 0).all()
    # sequence 1 has keys but no queries: zero dk/dv there
    k0 = lens_k[0]
    assert (dk[k0:k0 + 20] == 0).all() and (dv[k0:k0 + 20] == 0).all()
    # the ordinary sequences match a per-sequence dense call
    for b in (0, 3):
        qs, qe = int(cu_q[b]), int(cu_q[b + 1]); ks, ke = int(cu_k[b]), int(cu_k[b + 1])
        ref = flash_attn.flash_attn_func(q[qs:qe][None], k[ks:ke][None], v[ks:ke][None])
        assert torch.allclose(out[qs:qe].float(), ref[0].float(), atol=2e-3, rtol=2e-3)


def test_varlen_flat_work_list_many_sequences(monkeypatch):
    """More than 64 sequences (the slot owner search runs over several 64-lane chunks), zero-length ones in between,
    GQA with nheads % 8 != 0: forward and backward against the oracle, and bit-identical to the
    batch x max_seqlen grid (FA_VARLEN_GRID=1)."""
    rng = np.random.default_rng(7)
    B, Hq, Hk, D, dt = 150, 6, 2, 64, "fp16"
    lens = rng.integers(0, 300, size=B)
    lens[rng.integers(0, B, size=12)] = 0
    lens[17] = 700
    lens = [int(x) for x in lens]
    T = sum(lens)
    q = rand16((T, Hq, D), dt, 1).requires_grad_(True)
    k = rand16((T, Hk, D), dt, 2).requires_grad_(True)
    v = rand16((T, Hk, D), dt, 3).requires_grad_(True)
    do = rand16((T, Hq, D), dt, 4)
    cu = _cu(lens)
    mx = max(lens)

    def run():
        out, lse, _ = _fa().flash_attn_varlen_func(q, k, v, cu, cu, mx, mx, causal=True, return_attn_probs=True)
        return (out, lse) + tuple(torch.autograd.grad(out, (q, k, v), do))

    flat = run()
    monkeypatch.setenv("FA_VARLEN_GRID", "1")
    grid = run()
    monkeypatch.delenv("FA_VARLEN_GRID")
    for a, b in zip(flat, grid):
        assert torch.equal(a, b)
    c = cu.cpu().numpy()
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), f64(k), f64(v), c, c, mx, mx, D ** -0.5, causal=True)
    assert_close(f64(flat[0]), o_ref, dt, "out")
    assert_lse_close(f64(flat[1]), lse_ref, "lse")
    dq_r, dk_r, dv_r, _ = oracle.varlen_bwd(f64(do), f64(q), f64(k), f64(v), o_ref, lse_ref.astype(np.float64),
                                            c, c, mx, mx, D ** -0.5, causal=True)
    assert_close(f64(flat[2]), dq_r, dt, "dq", mult=2.0)
    assert_close(f64(flat[3]), dk_r, dt, "dk", mult=2.0)
    assert_close(f64(flat[4]), dv_r, dt, "dv", mult=2.0)


# packed launches whose flat list of key blocks x kv-heads is smaller than the chip split their query tiles like the dense
# ones (fa_bwd.hip: dkv_split_factor on the average pass; partial slabs [total_k, Hk, D], the reduction skips rows past the last sequence)
VSPLIT = [
    # lens_q, lens_k, Hq, Hk, D, dtype, causal, window
    ([900, 1300, 257], None, 8, 1, 128, "bf16", True, (-1, -1)),          # hand-scheduled kernel, group of 8
    ([2048, 3, 0, 1400], None, 8, 2, 64, "fp16", True, (-1, -1)),         # D 64: empty and tiny sequences next to long ones
    ([1500, 600], [300, 2000], 4, 4, 128, "fp16", False, (-1, -1)),       # Sq != Sk, no mask
    ([800, 800, 800], None, 6, 3, 96, "bf16", True, (300, 0)),            # 96 valid columns, left window
    ([1200, 900], None, 4, 2, 256, "bf16", True, (-1, -1)),               # head dim 256: each role stores its partial
]


@pytest.mark.parametrize("case", VSPLIT, ids=lambda c: "-".join(map(str, c)))
def test_small_packed_dkdv_launches_split_their_query_rows(case, monkeypatch):
    from flash_attn_mi355 import flash_attn_interface as fi
    lens_q, lens_k, Hq, Hk, D, dt, causal, window = case
    lens_k = lens_k or lens_q
    Tq, Tk = sum(lens_q), sum(lens_k)
    pad = 37                                                # rows past the last sequence: nobody's (dk / dv keep what they hold)
    q = rand16((Tq, Hq, D), dt, 61).requires_grad_(True)
    k = rand16((Tk + pad, Hk, D), dt, 62).requires_grad_(True)
    v = rand16((Tk + pad, Hk, D), dt, 63).requires_grad_(True)
    do = rand16((Tq, Hq, D), dt, 64)
    cu_q, cu_k = _cu(lens_q), _cu(lens_k)
    mq, mk = max(lens_q), max(lens_k)
    grads, ws = {}, {}
    real = fi._workspace
    for on in (True, False):
        seen = []
        monkeypatch.setattr(fi, "_workspace", lambda n, dev: (seen.append(n), real(n, dev))[1])
        run = lambda: _fa().flash_attn_varlen_func(q, k, v, cu_q, cu_k, mq, mk, causal=causal, window_size=window,
                                                   deterministic=not on)          # the public switch (FA_FLAG_NO_DKV_SPLIT)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            grads[on] = torch.autograd.grad(run(), (q, k, v), do)
            ws[on] = max(seen)
            again = torch.autograd.grad(run(), (q, k, v), do)
        for a_, b_ in zip(grads[on], again):
            assert torch.equal(a_[:Tk] if a_.shape[0] == Tk + pad else a_, b_[:Tk] if b_.shape[0] == Tk + pad else b_)
    assert ws[True] > ws[False]                              # the split form ran: it asked for the partial slabs
    assert torch.equal(grads[True][0], grads[False][0])
    cq, ck = cu_q.cpu().numpy(), cu_k.cpu().numpy()
    kw = dict(causal=causal, window=window)
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), f64(k[:Tk]), f64(v[:Tk]), cq, ck, mq, mk, D ** -0.5, **kw)
    g = oracle.varlen_bwd(f64(do), f64(q), f64(k[:Tk]), f64(v[:Tk]), o_ref, lse_ref.astype(np.float64), cq, ck, mq, mk, D ** -0.5, **kw)
    for on in (True, False):
        assert_close(f64(grads[on][1][:Tk]), g[1], dt, f"dk split={on}", mult=2.0)
        assert_close(f64(grads[on][2][:Tk]), g[2], dt, f"dv split={on}", mult=2.0)


def test_packed_batch_past_2gib_of_q():
    """A packed batch of 303 104 tokens at H 32 / D 128 (37 x 8192: 2.48 GB of q, its last sequence starts 2.4 GB into the
    tensor) - ordinary with sequence packing; the reference offsets with size_t (include/template.h:199-217).  Until round 6
    fa_api.hip bounded the WHOLE packed tensor by 2 GiB although every kernel rebases its descriptors at the sequence's first
    row.  Checked: the packed call == the dense call on the same memory viewed as (37, 8192, H, D) (forward, LSE and all three
    gradients), and one head of the LAST sequence against the fp64 oracle."""
    nseq, S, H, D, dt = 37, 8192, 32, 128, "bf16"
    T = nseq * S
    assert T * H * D * 2 > (1 << 31)
    gen = torch.Generator(device="cuda").manual_seed(77)
    mk = lambda: torch.randn((T, H, D), generator=gen, device="cuda", dtype=torch.bfloat16)
    q, k, v, do = mk().requires_grad_(True), mk().requires_grad_(True), mk().requires_grad_(True), mk()
    cu = _cu([S] * nseq)
    out, lse, _ = _fa().flash_attn_varlen_func(q, k, v, cu, cu, S, S, causal=True, return_attn_probs=True)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    assert all(torch.isfinite(t).all() for t in (out, dq, dk, dv))

    d4 = lambda t: t.detach().view(nseq, S, H, D)
    qd, kd, vd = (d4(t).requires_grad_(True) for t in (q, k, v))
    out_d, lse_d, _ = _fa().flash_attn_func(qd, kd, vd, causal=True, return_attn_probs=True)
    gd = torch.autograd.grad(out_d, (qd, kd, vd), d4(do))
    assert torch.equal(d4(out), out_d)                       # same kernel bodies, same tiles: the same bits
    assert torch.equal(lse.view(H, nseq, S).transpose(0, 1), lse_d)
    for name, a_, b_ in zip(("dq", "dk", "dv"), (dq, dk, dv), gd):
        assert_close(f64(d4(a_)[-2:]), f64(b_[-2:]), dt, name + " packed vs dense", mult=0.5)

    s0, h = T - S, 31                                        # last sequence (byte offset 2.4 GB), last head
    t = lambda x: f64(x[s0:, h])[None, None]
    o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=True)
    assert_close(t(out), o_ref, dt, "o last sequence")
    assert_lse_close(f64(lse[h, s0:])[None, None], lse_ref, "lse last sequence")
    # (dK / dV of one head need the other heads' queries only under GQA: H_q == H_k here, so one head is self-contained)
    g = oracle.attn_bwd(t(do), t(q), t(k), t(v), o_ref, lse_ref.astype(np.float64), D ** -0.5, causal=True)
    for name, got, ref in zip(("dq", "dk", "dv"), (dq, dk, dv), g):
        assert_close(t(got), ref, dt, name + " last sequence", mult=2.0)
