"""torch.library registration: the five ops exist with the
reference's names (kernel/fused_mha_api.cpp:308-358) and their fake implementations infer the
output shapes / dtypes, so the path is traceable without a GPU."""
import torch

import flash_attn_mi355.torch_ops  # noqa: F401  (registers the ops)


def test_ops_registered():
    for name in ("fwd", "bwd", "varlen_fwd", "varlen_bwd", "fwd_kvcache"):
        assert hasattr(torch.ops.flash_attn_mi355, name), name


def test_fake_dense_shapes():
    B, S, H, Hk, D = 2, 96, 4, 2, 64
    q = torch.empty(B, S, H, D, dtype=torch.bfloat16, device="meta")
    k = torch.empty(B, S + 32, Hk, D, dtype=torch.bfloat16, device="meta")
    v = torch.empty_like(k)
    out, lse, dmask, rng = torch.ops.flash_attn_mi355.fwd(q, k, v, None, 0.0, 0.125, True, -1, -1, 0.0, False)
    assert out.shape == q.shape and out.dtype == q.dtype
    assert lse.shape == (B, H, S) and lse.dtype == torch.float32
    assert dmask.numel() == 0 and rng.shape == (2,) and rng.dtype == torch.int64
    _, _, dmask, _ = torch.ops.flash_attn_mi355.fwd(q, k, v, None, 0.1, 0.125, False, -1, -1, 0.0, True)
    assert dmask.shape == (B, H, S, S + 32)
    dq, dk, dv, sd = torch.ops.flash_attn_mi355.bwd(out, q, k, v, out, lse, None, 0.0, 0.125, True, -1, -1,
                                                    0.0, False, None)
    assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape and sd.shape == (B, H, S)


def test_fake_varlen_and_kvcache_shapes():
    T, H, D = 300, 4, 128
    q = torch.empty(T, H, D, dtype=torch.float16, device="meta")
    cu = torch.empty(4, dtype=torch.int32, device="meta")
    out, lse, dmask, rng = torch.ops.flash_attn_mi355.varlen_fwd(q, q, q, cu, cu, None, None, 128, 128, 0.0,
                                                                 0.1, True, -1, -1, 0.0, False)
    assert out.shape == (T, H, D) and lse.shape == (H, T)
    dq, dk, dv, sd = torch.ops.flash_attn_mi355.varlen_bwd(out, q, q, q, out, lse, cu, cu, None, 128, 128, 0.0,
                                                           0.1, True, -1, -1, 0.0, False, None)
    assert dq.shape == (T, H, D) and sd.shape == (H, T)
    qd = torch.empty(3, 1, 8, 128, dtype=torch.float16, device="meta")
    kc = torch.empty(3, 1024, 2, 128, dtype=torch.float16, device="meta")
    o, l = torch.ops.flash_attn_mi355.fwd_kvcache(qd, kc, kc.clone(), None, None, None, None, None, None, None,
                                                  None, None, 0.1, True, -1, -1, 0.0, True, 0)
    assert o.shape == qd.shape and l.shape == (3, 8, 1)


def test_autograd_formula_traces_on_meta():
    q, k, v = (torch.empty(1, 64, 2, 64, dtype=torch.bfloat16, device="meta", requires_grad=True) for _ in range(3))
    out, lse, _, _ = torch.ops.flash_attn_mi355.fwd(q, k, v, None, 0.0, 0.125, True, -1, -1, 0.0, False)
    out.sum().backward()
    assert q.grad.shape == q.shape and k.grad.shape == k.shape and v.grad.shape == v.shape
