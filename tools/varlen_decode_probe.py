"""Decode issued through the varlen op (one query token per sequence, paged K / V with block_table + seqused_k, as vLLM-style
callers do) against the same step through flash_attn_with_kvcache.   python tools/varlen_decode_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_us(f, n=20):
    for _ in range(4): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
for (B, Tq, Hq, Hk, ctx) in ((1, 1, 32, 8, 8192), (8, 1, 32, 8, 8192), (64, 1, 32, 8, 8192), (16, 1, 32, 32, 4096), (8, 4, 32, 8, 8192), (8, 512, 32, 8, 8192)):
    D, page = 128, 256
    dt = torch.bfloat16
    nblk = B * ctx // page
    kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=dt); vc = torch.randn_like(kc)
    bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
    lens = torch.full((B,), ctx - 64, dtype=torch.int32, device="cuda")
    q = torch.randn(B, Tq, Hq, D, device="cuda", dtype=dt)
    qv = q.reshape(B * Tq, Hq, D)
    cu_q = (torch.arange(B + 1, dtype=torch.int32, device="cuda") * Tq)
    cu_k = torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda"), lens.cumsum(0).to(torch.int32)])
    o1 = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=True)
    o2 = fa.flash_attn_varlen_func(qv, kc, vc, cu_q, cu_k, Tq, ctx, causal=True, block_table=bt, seqused_k=lens)
    err = (o1.reshape(B * Tq, Hq, D).float() - o2.float()).abs().max().item()
    a = t_us(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=True))
    b = t_us(lambda: fa.flash_attn_varlen_func(qv, kc, vc, cu_q, cu_k, Tq, ctx, causal=True, block_table=bt, seqused_k=lens))
    print(f"B{B:3d} Tq{Tq:4d} Hq{Hq} Hk{Hk:2d} ctx{ctx}: kvcache op {a:8.1f} us   varlen op {b:8.1f} us   max|diff| {err:.3e}", flush=True)
