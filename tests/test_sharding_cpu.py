"""CPU: the multi-GPU decomposition (N > 1 path) on 2 gloo ranks: every rank attends its own
(batch, kv-head) shard with no collective on the data path; the gathered result equals the
unsharded computation.  The attention itself is the oracle here - the point under test is the
sharding logic that bench.py --gpus N / callers use."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from flash_attn_mi355.sharding import shard_alibi, shard_units


def test_shard_units_partition():
    for (B, Hq, Hk, W) in [(8, 16, 16, 8), (64, 32, 32, 8), (4, 32, 8, 8), (2, 8, 2, 4), (8, 16, 16, 1), (6, 12, 4, 2)]:
        seen = np.zeros((B, Hq), dtype=int)
        for r in range(W):
            bs, qs, ks = shard_units(B, Hq, Hk, W, r)
            seen[bs, qs] += 1
            g = Hq // Hk
            assert qs.start == ks.start * g and qs.stop == ks.stop * g      # GQA groups stay together
        assert (seen == 1).all()
    with pytest.raises(ValueError):
        shard_units(3, 4, 1, 2, 0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, k, v, slopes, out_file):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, S, Hq, D = q.shape
    Hk = k.shape[2]
    bs, qs, ks = shard_units(B, Hq, Hk, world, rank)
    sl = shard_alibi(slopes, qs, bs)
    t = lambda x: x.double().numpy().transpose(0, 2, 1, 3)
    o, lse, _ = oracle.attn_fwd(t(q[bs][:, :, qs]), t(k[bs][:, :, ks]), t(v[bs][:, :, ks]), D ** -0.5,
                                causal=True, alibi_slopes=sl.numpy())
    # only the timing barrier / result gathering use the process group - as in bench.py
    dist.barrier()
    mine = torch.from_numpy(np.ascontiguousarray(o.transpose(0, 2, 1, 3)))
    parts = [None] * world
    dist.all_gather_object(parts, (bs, qs, mine))
    tmax = torch.tensor([float(rank + 1)])
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        full = torch.zeros(B, S, Hq, D, dtype=torch.float64)
        for (b_, q_, m_) in parts:
            full[b_, :, q_] = m_
        torch.save({"out": full, "tmax": tmax}, out_file)
    dist.destroy_process_group()


def test_two_rank_sharded_attention_matches_unsharded(tmp_path):
    torch.manual_seed(0)
    B, S, Hq, Hk, D, world = 2, 48, 8, 4, 32, 2
    q = torch.randn(B, S, Hq, D); k = torch.randn(B, S, Hk, D); v = torch.randn(B, S, Hk, D)
    slopes = (2.0 ** (-8.0 * (torch.arange(Hq) + 1) / Hq)).float()
    out_file = str(tmp_path / "out.pt")
    mp.spawn(_worker, args=(world, _free_port(), q, k, v, slopes, out_file), nprocs=world, join=True)
    res = torch.load(out_file)
    t = lambda x: x.double().numpy().transpose(0, 2, 1, 3)
    o_ref, _, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=True, alibi_slopes=slopes.numpy())
    assert np.allclose(res["out"].numpy().transpose(0, 2, 1, 3), o_ref, atol=1e-12)
    assert float(res["tmax"]) == float(world)


def test_merge_attention_shards_matches_full_attention():
    """LSE merge of attention over disjoint key shards == attention over all keys (oracle on CPU)."""
    import numpy as np
    import torch
    import oracle
    from flash_attn_mi355.sharding import merge_attention_shards
    rng = np.random.default_rng(421)
    B, H, Sq, Sk, D = 2, 3, 17, 40, 16
    q = rng.standard_normal((B, H, Sq, D)); k = rng.standard_normal((B, H, Sk, D)); v = rng.standard_normal((B, H, Sk, D))
    o_ref, lse_ref, _ = oracle.attn_fwd(q, k, v, D ** -0.5)
    cuts = [0, 13, 13, 29, 40]                                     # includes an EMPTY shard
    outs, lses = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        o, l, _ = oracle.attn_fwd(q, k[:, :, a:b], v[:, :, a:b], D ** -0.5)
        outs.append(torch.from_numpy(o).permute(0, 2, 1, 3))           # (B, Sq, H, D)
        lses.append(torch.from_numpy(l))
    out, lse = merge_attention_shards(outs, lses)
    assert np.allclose(out.permute(0, 2, 1, 3).numpy(), o_ref, atol=1e-5)
    assert np.allclose(lse.numpy(), lse_ref, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------
# context parallelism: keys / values sharded along the sequence, (out, lse) all-gathered and merged
# ---------------------------------------------------------------------------------------------------------
def _torch_attn(q, k, v, window, scale):
    """test-local stand-in for the GPU kernel (the product has no CPU path): fp64 attention with a right window,
    bottom-right aligned like the kernels; returns (out, lse)."""
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    scale = D ** -0.5 if scale is None else scale
    qf = q.double().permute(0, 2, 1, 3)
    kf = k.double().permute(0, 2, 1, 3).repeat_interleave(H // Hk, 1)
    vf = v.double().permute(0, 2, 1, 3).repeat_interleave(H // Hk, 1)
    s = qf @ kf.transpose(-1, -2) * scale
    if window[1] >= 0:
        i = torch.arange(Sq)[:, None]; j = torch.arange(Sk)[None, :]
        s = s.masked_fill(j > i + (Sk - Sq) + window[1], float("-inf"))
    lse = torch.logsumexp(s, -1)
    p = torch.nan_to_num(torch.exp(s - lse[..., None]), nan=0.0)
    return (p @ vf).permute(0, 2, 1, 3).contiguous(), lse.float()


def _cp_worker(rank, world, port, q, k, v, causal, out_file):
    from flash_attn_mi355.sharding import context_parallel_attention
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    skl = k.shape[1] // world
    ks, vs = k[:, rank * skl:(rank + 1) * skl], v[:, rank * skl:(rank + 1) * skl]
    out, lse = context_parallel_attention(q, ks, vs, causal=causal, attn_fn=_torch_attn)
    if rank == 1:                                   # any rank holds the full result
        torch.save({"out": out, "lse": lse}, out_file)
    dist.destroy_process_group()


@pytest.mark.parametrize("causal", [False, True])
def test_context_parallel_attention_two_ranks(tmp_path, causal):
    torch.manual_seed(1)
    B, Sq, Sk, H, Hk, D, world = 2, 40, 96, 4, 2, 16, 2
    q = torch.randn(B, Sq, H, D); k = torch.randn(B, Sk, Hk, D); v = torch.randn(B, Sk, Hk, D)
    out_file = str(tmp_path / "cp.pt")
    mp.spawn(_cp_worker, args=(world, _free_port(), q, k, v, causal, out_file), nprocs=world, join=True)
    res = torch.load(out_file)
    o_ref, lse_ref = _torch_attn(q, k, v, (-1, 0) if causal else (-1, -1), None)
    assert (res["out"].double() - o_ref).abs().max() < 2e-6          # the merge runs in fp32
    assert (res["lse"].double() - lse_ref.double()).abs().max() < 1e-5
    # and against the oracle (independent of the stand-in)
    t = lambda x: x.double().numpy().transpose(0, 2, 1, 3)
    o2, lse2, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=causal)
    assert np.abs(res["out"].double().numpy().transpose(0, 2, 1, 3) - o2).max() < 2e-6


def _torch_attn_bwd(dout, q, k, v, out, lse, window, scale):
    """test-local stand-in for the backward kernels, written the way they work: P from the GIVEN lse, D from the GIVEN out
    (fp64) - with the merged lse / out of a context-parallel forward these are the global probabilities."""
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    G = H // Hk
    scale = D ** -0.5 if scale is None else scale
    qf, dof, of = (t.double().permute(0, 2, 1, 3) for t in (q, dout, out))
    kf = k.double().permute(0, 2, 1, 3).repeat_interleave(G, 1)
    vf = v.double().permute(0, 2, 1, 3).repeat_interleave(G, 1)
    s = qf @ kf.transpose(-1, -2) * scale
    if window[1] >= 0:
        i = torch.arange(Sq)[:, None]; j = torch.arange(Sk)[None, :]
        s = s.masked_fill(j > i + (Sk - Sq) + window[1], float("-inf"))
    p = torch.nan_to_num(torch.exp(s - lse.double()[..., None]), nan=0.0)
    dp = dof @ vf.transpose(-1, -2)
    dd = (dof * of).sum(-1, keepdim=True)
    ds = p * (dp - dd) * scale
    dq = (ds @ kf).permute(0, 2, 1, 3)
    dk = (ds.transpose(-1, -2) @ qf).reshape(B, Hk, G, Sk, D).sum(2).permute(0, 2, 1, 3)
    dv = (p.transpose(-1, -2) @ dof).reshape(B, Hk, G, Sk, D).sum(2).permute(0, 2, 1, 3)
    return dq.to(q.dtype).contiguous(), dk.to(k.dtype).contiguous(), dv.to(v.dtype).contiguous()


def _cp_bwd_worker(rank, world, port, q, k, v, dout, causal, out_dir):
    from flash_attn_mi355.sharding import context_parallel_attention
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    skl = k.shape[1] // world
    q = q.clone().requires_grad_(True)
    ks = k[:, rank * skl:(rank + 1) * skl].clone().requires_grad_(True)
    vs = v[:, rank * skl:(rank + 1) * skl].clone().requires_grad_(True)
    out, lse = context_parallel_attention(q, ks, vs, causal=causal, attn_fn=_torch_attn, bwd_fn=_torch_attn_bwd)
    out.backward(dout)
    torch.save({"dq": q.grad, "dk": ks.grad, "dv": vs.grad, "out": out.detach()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("causal", [False, True])
def test_context_parallel_backward_two_ranks(tmp_path, causal):
    """autograd through context_parallel_attention on two gloo ranks: dk / dv of each shard come out of the local
    backward with the merged out / lse, dq after one all-reduce - all equal the gradients of the unsharded problem
    (torch autograd in fp64 and the oracle)."""
    torch.manual_seed(2)
    B, Sq, Sk, H, Hk, D, world = 2, 40, 96, 4, 2, 16, 2
    q = torch.randn(B, Sq, H, D, dtype=torch.float64); k = torch.randn(B, Sk, Hk, D, dtype=torch.float64)
    v = torch.randn(B, Sk, Hk, D, dtype=torch.float64); dout = torch.randn(B, Sq, H, D, dtype=torch.float64)
    mp.spawn(_cp_bwd_worker, args=(world, _free_port(), q, k, v, dout, causal, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(str(tmp_path / f"r{r}.pt")) for r in range(world)]
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref, _ = _torch_attn(qr, kr, vr, (-1, 0) if causal else (-1, -1), None)
    o_ref.backward(dout)
    skl = Sk // world
    for r in range(world):
        assert (res[r]["out"] - o_ref.detach()).abs().max() < 2e-6       # (the merge runs in fp32)
        assert (res[r]["dq"] - qr.grad).abs().max() < 5e-6, r
        assert (res[r]["dk"] - kr.grad[:, r * skl:(r + 1) * skl]).abs().max() < 5e-6, r
        assert (res[r]["dv"] - vr.grad[:, r * skl:(r + 1) * skl]).abs().max() < 5e-6, r
    t = lambda x: x.double().numpy().transpose(0, 2, 1, 3)
    o2, lse2, _ = oracle.attn_fwd(t(q), t(k), t(v), D ** -0.5, causal=causal)
    g = oracle.attn_bwd(t(dout), t(q), t(k), t(v), o2, lse2, D ** -0.5, causal=causal)
    assert np.abs(t(res[0]["dq"]) - g[0]).max() < 5e-6
    assert np.abs(np.concatenate([t(res[0]["dk"]), t(res[1]["dk"])], 2) - g[1]).max() < 5e-6
    assert np.abs(np.concatenate([t(res[0]["dv"]), t(res[1]["dv"])], 2) - g[2]).max() < 5e-6
