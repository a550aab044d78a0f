"""Decode step with few kv-heads per GPU (tensor-parallel shards: H 8/1, 16/2, 4/1) against H 32/8 at equal cache bytes.
  python tools/decode_fewheads_probe.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa

def b2b(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

ctx, page, D = 8192, 256, 128
for dt_kv in (torch.float16, torch.float8_e4m3fn):
    for (H, Hk) in ((32, 8), (8, 1), (16, 2), (4, 1), (8, 8)):
        for B in (1, 8, 64, 256):
            if B * Hk * ctx * D * 2 * (2 if dt_kv == torch.float16 else 1) > 20e9: continue
            nblk = B * ctx // page
            kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.float16)
            vc = torch.randn_like(kc)
            if dt_kv != torch.float16:
                kc, vc = kc.to(dt_kv), vc.to(dt_kv)
            bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
            lens = torch.full((B,), ctx - 3, dtype=torch.int32, device="cuda")
            q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.float16)
            t = b2b(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=True))
            byts = 2.0 * B * (ctx - 3) * Hk * D * kc.element_size()
            print(f"{str(dt_kv).split('.')[-1]:14s} H{H}/{Hk} B{B:4d}: {t:8.1f} us  {byts / t / 1e6:7.2f} TB/s", flush=True)
