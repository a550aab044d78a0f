"""GPU parity: dropout reproduces the reference's Philox-4x32-10 stream bit-exactly
(include/softmax.h:97-114) - the keep mask returned in `dmask` equals the oracle's mask for the
same (seed, offset), and outputs / gradients match the oracle run with that mask."""
import numpy as np
import pytest
import torch

import oracle
from util import assert_close, f64, rand16

pytestmark = pytest.mark.gpu


def _fa():
    import flash_attn
    return flash_attn


def _gen_state():
    g = torch.cuda.default_generators[torch.cuda.current_device()]
    return g.initial_seed(), g.get_offset()


@pytest.mark.parametrize("case", [
    (2, 4, 2, 128, 128, 128, "fp16", False, 0.17),
    (1, 2, 2, 200, 200, 64, "bf16", True, 0.1),
    (1, 2, 1, 96, 130, 128, "fp16", True, 0.5),      # Sk % 4 != 0: straddling Philox counters
])
def test_dense_dropout_mask_output_and_grads(case):
    B, Hq, Hk, Sq, Sk, D, dt, causal, pdrop = case
    torch.manual_seed(1234)
    q = rand16((B, Sq, Hq, D), dt, 1).requires_grad_(True)
    k = rand16((B, Sk, Hk, D), dt, 2).requires_grad_(True)
    v = rand16((B, Sk, Hk, D), dt, 3).requires_grad_(True)
    do = rand16((B, Sq, Hq, D), dt, 4)
    seed, offset = _gen_state()
    out, lse, dmask = _fa().flash_attn_func(q, k, v, dropout_p=pdrop, causal=causal, return_attn_probs=True)
    seed2, offset2 = _gen_state()
    assert seed2 == seed and offset2 == offset + B * Hq * 32       # fused_mha_forward.cu:382-383
    keep = oracle.dropout_keep_mask(seed, offset, pdrop, Sq, Sk)
    dm = f64(dmask)                                                # [B,H,Sq,Sk], +1 kept / -1 dropped / 0 unvisited
    vis = oracle.attention.visible_mask(Sq, Sk, causal, -1, -1)
    for b in range(B):
        for h in range(Hq):
            got = dm[b, h]
            assert np.array_equal(got[vis] > 0, keep[vis]), "dropout keep mask differs from the Philox oracle"
            assert set(np.unique(got)).issubset({-1.0, 0.0, 1.0})
    t = lambda x: f64(x).transpose(0, 2, 1, 3)
    kw = dict(causal=causal, dropout_p=pdrop, seed=seed, offset=offset)
    o_ref, lse_ref, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, **kw)
    assert_close(t(out), o_ref, dt, "out", mult=1.5)
    assert np.abs(f64(lse) - lse_ref).max() < 2e-3                 # LSE is pre-dropout
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    g = oracle.attn_bwd(t(do), t(q), t(k), t(v), o_ref, lse_ref.astype(np.float64), D ** -0.5, **kw)
    assert_close(t(dq), g[0], dt, "dq", mult=3.0)
    assert_close(t(dk), g[1], dt, "dk", mult=3.0)
    assert_close(t(dv), g[2], dt, "dv", mult=3.0)


def test_varlen_dropout_vs_oracle():
    lens = [70, 5, 130]
    Hq, Hk, D, dt, pdrop = 4, 2, 64, "fp16", 0.2
    T = sum(lens)
    q = rand16((T, Hq, D), dt, 1).requires_grad_(True)
    k = rand16((T, Hk, D), dt, 2).requires_grad_(True)
    v = rand16((T, Hk, D), dt, 3).requires_grad_(True)
    do = rand16((T, Hq, D), dt, 4)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")
    seed, offset = _gen_state()
    out = _fa().flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), dropout_p=pdrop, causal=True)
    cun = cu.cpu().numpy()
    kw = dict(causal=True, dropout_p=pdrop, seed=seed, offset=offset)
    o_ref, lse_ref = oracle.varlen_fwd(f64(q), f64(k), f64(v), cun, cun, max(lens), max(lens), D ** -0.5, **kw)
    assert_close(f64(out), o_ref, dt, "out", mult=1.5)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    g = oracle.varlen_bwd(f64(do), f64(q), f64(k), f64(v), o_ref, lse_ref.astype(np.float64), cun, cun,
                          max(lens), max(lens), D ** -0.5, **kw)
    assert_close(f64(dq), g[0], dt, "dq", mult=3.0)
    assert_close(f64(dk), g[1], dt, "dk", mult=3.0)
    assert_close(f64(dv), g[2], dt, "dv", mult=3.0)


def test_dropout_rejected_with_softcap():
    q = rand16((1, 16, 2, 64), "fp16", 1)
    with pytest.raises(RuntimeError):
        _fa().flash_attn_func(q, q, q, dropout_p=0.1, softcap=10.0)
