"""Decode step time and KV rate across head dims (64 / 96 / 128 / 256), paged bf16 cache.  python tools/decode_head_dims.py"""
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
for D in (128, 256, 96, 64):
    for (B, Hq, Hk, ctx) in ((1, 16, 8, 8192), (8, 16, 8, 8192), (64, 16, 8, 8192), (16, 8, 4, 4096)):
        page = 256
        nblk = B * ctx // page
        kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.bfloat16); vc = torch.randn_like(kc)
        bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
        lens = torch.full((B,), ctx - 64, dtype=torch.int32, device="cuda")
        q = torch.randn(B, 1, Hq, D, device="cuda", dtype=torch.bfloat16)
        us = t_us(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=True))
        gb = 2.0 * (ctx - 64) * Hk * D * 2 * B / 1e9
        print(f"D{D:3d} B{B:3d} Hq{Hq} Hk{Hk} ctx{ctx}: {us:8.1f} us  {gb / us * 1e3:5.2f} TB/s", flush=True)
