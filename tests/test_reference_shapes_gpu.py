"""GPU parity: EVERY shape of the reference's own test list (/root/reference/test.py:115-139, x causal in {F, T}) under the
reference's own protocol (test.py:147-159 inputs, 273-277 forward gate, 322-334 gradient gates):

    torch.manual_seed(421); q, k, v, dO = randn fp16 [B, H, M|N, D]; scale = 1 / sqrt(D)
    forward : max|o - o_ref|   <= 2 max|o_fp16torch  - o_ref|  + 1e-5      (o_ref = the fp32 evaluation of the same formula)
    backward: max|dX - dX_ref| <= 3 max|dX_fp16torch - dX_ref| + 1e-4      for X in {Q, K, V}

The 1-tile shapes up to (1, 1, 256, 256, 256) are also committed as golden vectors generated from the reference's functions
(tests/golden/make_golden.py); the long shapes are evaluated here with the same formulas on the GPU in fp32, a few heads at a
time (the reference itself only checks "finite" there when its comparator runs out of memory on a V100, test.py:296-304).
Head dim 256 is where round 5's two-wave dK/dV kernel landed: (1, 32, 4096, 4096, 256) and (1, 32, 8192, 8192, 256) run it.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

REFERENCE_SHAPES = [            # B, H, M, N, D  (test.py:116-138)
    (1, 1, 16, 16, 16), (1, 1, 32, 32, 32), (1, 1, 64, 64, 64), (1, 1, 128, 128, 128), (1, 1, 256, 256, 256),
    (1, 16, 1024, 1024, 16), (1, 32, 1024, 1024, 16),
    (1, 16, 1024, 1024, 32), (1, 32, 1024, 1024, 32),
    (1, 16, 1024, 1024, 64), (1, 32, 1024, 1024, 64),
    (1, 16, 1024, 1024, 128), (1, 32, 2048, 2048, 128), (1, 32, 4096, 4096, 128),
    (1, 16, 1024, 1024, 256), (1, 32, 2048, 2048, 256), (1, 32, 4096, 4096, 256), (1, 32, 8192, 8192, 256),
]


def _attention(q, k, v, scale, causal):
    s = torch.einsum("bhmd,bhnd->bhmn", q, k) * scale
    if causal:
        m = torch.triu(torch.ones(s.shape[-2], s.shape[-1], device=s.device, dtype=torch.bool), 1)
        s = s.masked_fill(m, float("-inf"))
    return torch.einsum("bhmn,bhnd->bhmd", torch.softmax(s, -1), v)


def _fwd_bwd(q, k, v, do, scale, causal, upcast):
    """The comparator of the protocol: fp32 evaluation (upcast) or the native fp16 one, a few heads at a time."""
    H = q.shape[1]
    step = max(1, min(H, (1 << 28) // (q.shape[2] * k.shape[2])))        # <= 1 GiB of fp32 scores per chunk
    outs, grads = [], ([], [], [])
    for h0 in range(0, H, step):
        qq, kk, vv = (t[:, h0:h0 + step].detach().clone().requires_grad_(True) for t in (q, k, v))
        a, b, c = (qq.float(), kk.float(), vv.float()) if upcast else (qq, kk, vv)
        o = _attention(a, b, c, scale, causal).to(q.dtype)
        g = torch.autograd.grad(o, (qq, kk, vv), do[:, h0:h0 + step])
        outs.append(o.detach())
        for dst, x in zip(grads, g):
            dst.append(x)
    return torch.cat(outs, 1), tuple(torch.cat(x, 1) for x in grads)


@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
@pytest.mark.parametrize("shape", REFERENCE_SHAPES, ids=lambda s: "B%dH%dM%dN%dD%d" % s)
def test_reference_shape_under_reference_protocol(shape, causal):
    import flash_attn
    B, H, M, N, D = shape
    torch.manual_seed(421)
    q = torch.randn(B, H, M, D, device="cuda", dtype=torch.float16)
    k = torch.randn(B, H, N, D, device="cuda", dtype=torch.float16)
    v = torch.randn(B, H, N, D, device="cuda", dtype=torch.float16)
    do = torch.randn(B, H, M, D, device="cuda", dtype=torch.float16)
    scale = 1.0 / math.sqrt(D)

    qq, kk, vv = (t.transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))     # the API's (B, S, H, D), as views
    out = flash_attn.flash_attn_func(qq, kk, vv, softmax_scale=scale, causal=causal)
    dq, dk, dv = torch.autograd.grad(out, (qq, kk, vv), do.transpose(1, 2))
    out, dq, dk, dv = (t.transpose(1, 2) for t in (out, dq, dk, dv))

    o_ref, g_ref = _fwd_bwd(q, k, v, do, scale, causal, upcast=True)
    o_pt, g_pt = _fwd_bwd(q, k, v, do, scale, causal, upcast=False)

    assert torch.isfinite(out).all()
    err, err_pt = (out - o_ref).abs().max().item(), (o_pt - o_ref).abs().max().item()
    assert err <= 2.0 * err_pt + 1e-5, ("o", err, err_pt)
    for name, got, ref, pt in zip(("dq", "dk", "dv"), (dq, dk, dv), g_ref, g_pt):
        assert torch.isfinite(got).all(), name
        e, e_pt = (got - ref).abs().max().item(), (pt - ref).abs().max().item()
        assert e <= 3.0 * e_pt + 1e-4, (name, e, e_pt)
