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
