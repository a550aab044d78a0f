"""GPU: the ops are capturable in HIP graphs (torch.cuda.CUDAGraph) - no host synchronisation, no allocation outside
torch's pool, no host-side reads of device data on the hot path - and a replayed graph gives the eager results.
This is how a serving loop runs the decode step (`flash_attn_with_kvcache` with in-place append, lengths in a device
tensor) and how a training loop may run forward + backward.  The reference launches on the current stream without
host synchronisation as well, except for `rng_state.cpu()` in the dropout backward (kernel/fused_mha_backward.cu:660-666)."""
import pytest
import torch

from util import DT, rand16

pytestmark = pytest.mark.gpu


def _fa():
    import flash_attn
    return flash_attn


def _capture(fn, warm=3):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


@pytest.mark.parametrize("kind", ["fp16-dense", "bf16-paged", "fp8-paged-split"])
def test_decode_step_replays_in_a_graph(kind):
    """one decode step = append the new token's K / V (with RoPE) at cache_seqlens, attend, lengths += 1: captured once,
    replayed for several steps with new q / k / v contents, against the same steps run eagerly on a copy of the cache"""
    fa = _fa()
    dt = "bf16" if kind.startswith("bf16") else "fp16"
    paged = "paged" in kind
    fp8 = kind.startswith("fp8")
    B, Hq, Hk, D, Smax, page = (4, 16, 4, 128, 1024, 256) if not kind.endswith("split") else (2, 32, 8, 128, 4096, 256)
    g = torch.Generator().manual_seed(5)
    nblk = B * Smax // page
    if paged:
        kc = rand16((nblk, page, Hk, D), dt, 2)
        vc = rand16((nblk, page, Hk, D), dt, 3)
        bt = torch.randperm(nblk, generator=g).to(torch.int32).reshape(B, Smax // page).cuda()
    else:
        kc = rand16((B, Smax, Hk, D), dt, 2)
        vc = rand16((B, Smax, Hk, D), dt, 3)
        bt = None
    kw = {}
    if fp8:
        kc = (kc.float() * 0.5).to(torch.float8_e4m3fn)
        vc = (vc.float() * 0.5).to(torch.float8_e4m3fn)
        kw = dict(k_descale=2.0, v_descale=2.0)
    lens0 = torch.randint(Smax // 2, Smax - 16, (B,), generator=g, dtype=torch.int32).cuda()
    rd = 64
    pos = torch.arange(Smax + 8, dtype=torch.float32)[:, None]
    inv = 1.0 / (10000 ** (torch.arange(0, rd, 2, dtype=torch.float32) / rd))[None, :]
    cos, sin = torch.cos(pos * inv).to(DT[dt]).cuda(), torch.sin(pos * inv).to(DT[dt]).cuda()
    steps = 4
    qs = [rand16((B, 1, Hq, D), dt, 10 + i) for i in range(steps)]
    ks = [rand16((B, 1, Hk, D), dt, 20 + i) for i in range(steps)]
    vs = [rand16((B, 1, Hk, D), dt, 30 + i) for i in range(steps)]

    def make_step(kc_, vc_, lens_, q_, k_, v_):
        def step():
            o, lse = fa.flash_attn_with_kvcache(q_, kc_, vc_, k=k_, v=v_, rotary_cos=cos, rotary_sin=sin, cache_seqlens=lens_, block_table=bt,
                                                causal=True, rotary_interleaved=False, return_softmax_lse=True, **kw)
            lens_.add_(1)
            return o, lse
        return step

    # eager reference on copies
    kc_e, vc_e, lens_e = kc.clone(), vc.clone(), lens0.clone()
    ref = []
    for i in range(steps):
        o, lse = make_step(kc_e, vc_e, lens_e, qs[i], ks[i], vs[i])()
        ref.append((o.clone(), lse.clone()))
    torch.cuda.synchronize()

    # captured: static input buffers, contents replaced before each replay
    kc_g, vc_g, lens_g = kc.clone(), vc.clone(), lens0.clone()
    q_s = qs[0].clone()
    k_s, v_s = ks[0].clone(), vs[0].clone()
    step = make_step(kc_g, vc_g, lens_g, q_s, k_s, v_s)
    graph, (o_s, lse_s) = _capture(step)
    # the warm-up and the capture ran the step (appends, lengths += 1): restore the state
    kc_g.copy_(kc); vc_g.copy_(vc); lens_g.copy_(lens0)
    for i in range(steps):
        q_s.copy_(qs[i])
        k_s.copy_(ks[i]); v_s.copy_(vs[i])
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o_s, ref[i][0]), f"step {i}: out differs from the eager step"
        assert torch.equal(lse_s, ref[i][1]), f"step {i}: lse differs"
    assert torch.equal(lens_g, lens_e)
    bits = torch.int8 if fp8 else torch.int16
    assert torch.equal(kc_g.view(bits), kc_e.view(bits)) and torch.equal(vc_g.view(bits), vc_e.view(bits))


@pytest.mark.parametrize("varlen", [False, True])
def test_training_step_replays_in_a_graph(varlen):
    """forward + backward captured as one graph (no dropout: the Philox offset is host state); replay = eager, bitwise"""
    fa = _fa()
    dt = "bf16"
    H, Hk, D = 8, 2, 128
    if varlen:
        lens = [300, 1024, 77, 647]
        T = sum(lens)
        cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32).cuda()
        shape_q, shape_k = (T, H, D), (T, Hk, D)
        call = lambda q, k, v: fa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True)
    else:
        shape_q, shape_k = (2, 1024, H, D), (2, 1024, Hk, D)
        call = lambda q, k, v: fa.flash_attn_func(q, k, v, causal=True)
    q = rand16(shape_q, dt, 1).requires_grad_()
    k = rand16(shape_k, dt, 2).requires_grad_()
    v = rand16(shape_k, dt, 3).requires_grad_()
    do = rand16(shape_q, dt, 4)

    def step():
        o = call(q, k, v)
        dq, dk, dv = torch.autograd.grad(o, (q, k, v), do)
        return o, dq, dk, dv

    # detached copies: a live autograd graph from this eager step would tie q / k / v's AccumulateGrad nodes to the default
    # stream, and the captured backward would then wait on that stream (illegal inside a capture)
    ref = [t.detach().clone() for t in step()]
    graph, outs = _capture(step)
    for rep in range(2):
        with torch.no_grad():
            for t in outs:
                t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for name, a, b in zip(("out", "dq", "dk", "dv"), outs, ref):
            assert torch.equal(a, b), f"{name} differs from the eager step (replay {rep})"
