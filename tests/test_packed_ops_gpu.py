"""GPU: packed entry points (upstream API) and the torch.library ops give bit-identical results
to the unpacked functional API (same kernels, only the strides / the dispatch differ), and the
unpacked API itself is checked against the oracle elsewhere (test_fwd_gpu / test_bwd_gpu)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fa():
    import flash_attn
    return flash_attn


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [64, 128, 96])
def test_qkvpacked_matches_unpacked(dtype, D):
    torch.manual_seed(421)
    B, S, H = 2, 200, 4
    qkv = torch.randn(B, S, 3, H, D, device="cuda", dtype=dtype, requires_grad=True)
    do = torch.randn(B, S, H, D, device="cuda", dtype=dtype)
    fa = _fa()
    out_p = fa.flash_attn_qkvpacked_func(qkv, causal=True)
    (dqkv,) = torch.autograd.grad(out_p, qkv, do)
    q, k, v = (qkv[:, :, i].detach().clone().requires_grad_(True) for i in range(3))
    out_u = fa.flash_attn_func(q, k, v, causal=True)
    dq, dk, dv = torch.autograd.grad(out_u, (q, k, v), do)
    assert torch.equal(out_p, out_u)
    assert dqkv.shape == qkv.shape
    for i, g in enumerate((dq, dk, dv)):
        assert torch.equal(dqkv[:, :, i], g), i


def test_kvpacked_matches_unpacked_gqa():
    torch.manual_seed(421)
    B, Sq, Sk, H, Hk, D = 2, 130, 260, 8, 2, 128
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    kv = torch.randn(B, Sk, 2, Hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn_like(q)
    fa = _fa()
    out_p = fa.flash_attn_kvpacked_func(q, kv, causal=True, window_size=(64, 0))
    dq_p, dkv = torch.autograd.grad(out_p, (q, kv), do)
    q2 = q.detach().clone().requires_grad_(True)
    k, v = (kv[:, :, i].detach().clone().requires_grad_(True) for i in range(2))
    out_u = fa.flash_attn_func(q2, k, v, causal=True, window_size=(64, 0))
    dq, dk, dv = torch.autograd.grad(out_u, (q2, k, v), do)
    assert torch.equal(out_p, out_u) and torch.equal(dq_p, dq)
    assert torch.equal(dkv[:, :, 0], dk) and torch.equal(dkv[:, :, 1], dv)


def test_varlen_packed_matches_unpacked():
    torch.manual_seed(421)
    lens = [37, 128, 5, 200]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    T, H, D = sum(lens), 4, 64
    qkv = torch.randn(T, 3, H, D, device="cuda", dtype=torch.float16, requires_grad=True)
    do = torch.randn(T, H, D, device="cuda", dtype=torch.float16)
    fa = _fa()
    out_p = fa.flash_attn_varlen_qkvpacked_func(qkv, cu, max(lens), causal=True)
    (dqkv,) = torch.autograd.grad(out_p, qkv, do)
    q, k, v = (qkv[:, i].detach().clone().requires_grad_(True) for i in range(3))
    out_u = fa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True)
    dq, dk, dv = torch.autograd.grad(out_u, (q, k, v), do)
    assert torch.equal(out_p, out_u)
    for i, g in enumerate((dq, dk, dv)):
        assert torch.equal(dqkv[:, i], g), i
    kv = qkv[:, 1:].detach().clone()
    out_kv = fa.flash_attn_varlen_kvpacked_func(q.detach(), kv, cu, cu, max(lens), max(lens), causal=True)
    assert torch.equal(out_kv, out_u)


def test_torch_ops_match_functional_api():
    import flash_attn_mi355.torch_ops  # noqa: F401
    torch.manual_seed(421)
    B, S, H, D = 2, 256, 4, 128
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    do = torch.randn_like(q)
    scale = D ** -0.5
    out, lse, dmask, rng = torch.ops.flash_attn_mi355.fwd(q, k, v, None, 0.0, scale, True, -1, -1, 0.0, False)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    q2, k2, v2 = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    out2, lse2, _ = _fa().flash_attn_func(q2, k2, v2, causal=True, return_attn_probs=True)
    dq2, dk2, dv2 = torch.autograd.grad(out2, (q2, k2, v2), do)
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
    # dropout: the rng_state returned by fwd replays the same mask in bwd
    out_d, lse_d, dm, rng = torch.ops.flash_attn_mi355.fwd(q, k, v, None, 0.2, scale, False, -1, -1, 0.0, True)
    assert dm.shape == (B, H, S, S) and rng.shape == (2,)
    g1 = torch.ops.flash_attn_mi355.bwd(do, q, k, v, out_d, lse_d, None, 0.2, scale, False, -1, -1, 0.0, False, rng)
    g2 = torch.ops.flash_attn_mi355.bwd(do, q, k, v, out_d, lse_d, None, 0.2, scale, False, -1, -1, 0.0, False, rng)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    assert torch.isfinite(g1[0].float()).all()


def test_torch_ops_opcheck_and_compile():
    import flash_attn_mi355.torch_ops  # noqa: F401
    torch.manual_seed(421)
    q, k, v = (torch.randn(1, 128, 2, 64, device="cuda", dtype=torch.float16, requires_grad=True) for _ in range(3))
    torch.library.opcheck(torch.ops.flash_attn_mi355.fwd.default,
                          (q, k, v, None, 0.0, 0.125, True, -1, -1, 0.0, False),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))

    @torch.compile(fullgraph=True, backend="eager")
    def f(q, k, v):
        return torch.ops.flash_attn_mi355.fwd(q, k, v, None, 0.0, 0.125, True, -1, -1, 0.0, False)[0]

    ref = torch.ops.flash_attn_mi355.fwd(q, k, v, None, 0.0, 0.125, True, -1, -1, 0.0, False)[0]
    assert torch.equal(f(q, k, v), ref)


def test_torch_ops_kvcache_mutates_cache():
    import flash_attn_mi355.torch_ops  # noqa: F401
    torch.manual_seed(421)
    B, H, Hk, D, cap = 2, 8, 2, 128, 512
    q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.float16)
    kc = torch.randn(B, cap, Hk, D, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    kn, vn = torch.randn(B, 1, Hk, D, device="cuda", dtype=torch.float16), torch.randn(B, 1, Hk, D, device="cuda", dtype=torch.float16)
    lens = torch.tensor([100, 333], dtype=torch.int32, device="cuda")
    kc2, vc2 = kc.clone(), vc.clone()
    out, lse = torch.ops.flash_attn_mi355.fwd_kvcache(q, kc, vc, kn, vn, lens, None, None, None, None, None, None,
                                                      D ** -0.5, True, -1, -1, 0.0, True, 0)
    out2, lse2 = _fa().flash_attn_with_kvcache(q, kc2, vc2, k=kn, v=vn, cache_seqlens=lens, causal=True,
                                               return_softmax_lse=True)
    assert torch.equal(out, out2) and torch.equal(lse, lse2)
    assert torch.equal(kc, kc2) and torch.equal(kc[0, 100], kn[0, 0]) and torch.equal(vc[1, 333], vn[1, 0])


def test_context_parallel_merge_with_kernel_lse():
    """Keys split in two shards (what two context-parallel ranks would hold): merging the kernels' (out, lse)
    pairs reproduces attention over all keys - causal included (shard 2 is a bottom-right aligned causal
    problem; shard 1 is square-causal for the first half of the rows and unmasked for the second)."""
    from flash_attn_mi355.sharding import merge_attention_shards
    torch.manual_seed(421)
    B, S, H, D = 2, 384, 4, 128
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    fa = _fa()
    h = S // 2
    for causal in (False, True):
        full, lse_full, _ = fa.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
        if causal:
            o1a, l1a, _ = fa.flash_attn_func(q[:, :h], k[:, :h], v[:, :h], causal=True, return_attn_probs=True)
            o1b, l1b, _ = fa.flash_attn_func(q[:, h:], k[:, :h], v[:, :h], return_attn_probs=True)
            o1, l1 = torch.cat([o1a, o1b], 1), torch.cat([l1a, l1b], 2)
            o2, l2, _ = fa.flash_attn_func(q, k[:, h:], v[:, h:], causal=True, return_attn_probs=True)
        else:
            o1, l1, _ = fa.flash_attn_func(q, k[:, :h], v[:, :h], return_attn_probs=True)
            o2, l2, _ = fa.flash_attn_func(q, k[:, h:], v[:, h:], return_attn_probs=True)
        out, lse = merge_attention_shards([o1, o2], [l1, l2])
        assert torch.allclose(out.float(), full.float(), atol=2e-2, rtol=2e-2), causal
        assert torch.allclose(lse, lse_full, atol=2e-3), causal
