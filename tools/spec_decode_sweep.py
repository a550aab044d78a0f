"""Speculative / multi-token decode: T_q query tokens per sequence over a paged cache (causal), time and KV rate.
The packed rows of a kv-head (T_q x G) decide the kernel: <= 32 rows the MFMA decode kernel (split-KV), more: row blocks
(fp8 caches) or fa_fwd_kernel on the cache.   python tools/spec_decode_sweep.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_us(f, n=20):
    for _ in range(4): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
TQ = tuple(int(x) for x in os.environ.get("TQ", "1,2,4,8,9,16,32,64").split(","))
print("us per step (KV TB/s)".ljust(34) + "".join(f"{'Tq=' + str(t):>16s}" for t in TQ))
for kv in ("fp16", "fp8"):
    for (B, Hq, Hk, ctx) in ((1, 32, 8, 8192), (8, 32, 8, 8192), (32, 32, 8, 8192), (8, 64, 8, 8192), (8, 32, 32, 8192), (64, 64, 8, 4096)):
        D, page = 128, 256
        nblk = B * ctx // page
        kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.float16); vc = torch.randn_like(kc)
        kw = {}
        if kv == "fp8":
            kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn); kw = dict(k_descale=1.0, v_descale=1.0)
        bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
        row = f"B{B:3d} Hq{Hq} Hk{Hk:2d} ctx{ctx} {kv:4s}:".ljust(34)
        for T in TQ:
            lens = torch.full((B,), ctx - 64 - T, dtype=torch.int32, device="cuda")
            q = torch.randn(B, T, Hq, D, device="cuda", dtype=torch.float16)
            kn = torch.randn(B, T, Hk, D, device="cuda", dtype=torch.float16); vn = torch.randn_like(kn)
            us = t_us(lambda: fa.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, cache_seqlens=lens, block_table=bt, causal=True, **kw))
            gb = 2.0 * (ctx - 64) * Hk * D * kc.element_size() * B / 1e9
            row += f"{us:9.1f} ({gb / us * 1e3:4.2f})"
        print(row, flush=True)
        del kc, vc
